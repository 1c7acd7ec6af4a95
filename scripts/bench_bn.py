"""Micro-benchmark of the BatchNorm(+activation) passes per layer shape, replayed from a HIP graph (dependent
launches, as in the real iteration).  Usage: python scripts/bench_bn.py [B]   -> us and GB/s vs algorithmic bytes
(train fwd = 3·n·4 B, bwd = 5·n·4 B)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from video_filler_amd.backend import get_backend

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
hb = get_backend()
SHAPES = [(64, 32), (128, 16), (256, 8), (512, 4), (4000, 1), (64, 64), (192, 32)]
tot = {"fwd": 0.0, "bwd": 0.0}
for C, H in SHAPES:
    x = hb.empty_act(B, C, H, H).normal_()
    y = torch.empty_like(x)
    gy = torch.empty_like(x).normal_()
    gx = torch.empty_like(x)
    gamma, beta = hb.zeros(C) + 1, hb.zeros(C)
    rm, rv, sm, si = hb.zeros(C), hb.zeros(C) + 1, hb.zeros(C), hb.zeros(C)
    gg, gb = hb.zeros(C), hb.zeros(C)
    sums = torch.zeros(2 * C, dtype=torch.float64, device=x.device)
    fns = {
        "fwd": lambda: hb.bn_train_fwd(x, y, gamma, beta, rm, rv, sm, si, sums, 0.1, 1e-5, "lrelu", 0.2),
        "bwd": lambda: hb.bn_bwd(x, y, gy, gx, gg, gb, gamma, sm, si, sums, "lrelu", 0.2, 0.0),
    }
    n = x.numel()
    for name, fn in fns.items():
        fn(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            hb.use_current_stream()
            for _ in range(20):
                fn()
        hb.use_current_stream()
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            g.replay()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 200 * 1e3
        alg = (3 if name == "fwd" else 5) * n * 4
        print("C=%4d H=%3d n=%6.2f MB  %s %7.2f us  %7.1f GB/s (alg)" % (C, H, n * 4 / 1e6, name, us, alg / us / 1e3))
