"""Would the fused bottleneck update (HBM-bound, 0.29 ms) hide beside the encoder's data-gradient passes (matrix-core-bound)?
Two streams of one process: the planes GEMMs of E5..E2's data-gradients (x REPS per round) on the main stream, the update of the two
bottleneck tensors on a side stream, one round = one fork/join.   python scripts/probe/adam_beside_gemms.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from video_filler_amd.backend import get_backend

hb = get_backend()
side = hb.fork()
B = 64
REPS = int(os.environ.get("REPS", "3"))
passes = []
for name, Cin, H, Cout in (("E5", 256, 8, 512), ("E4", 128, 16, 256), ("E3", 64, 32, 128), ("E2", 64, 64, 64)):
    x = hb.empty_act(B, Cin, H, H).normal_()
    y = hb.empty_act(B, Cout, H // 2, H // 2).normal_()
    gx = hb.empty_act(B, Cin, H, H)
    w = (hb.empty(Cout, 4, 4, Cin).normal_() * 0.02).permute(0, 3, 1, 2)
    yp = hb.planes_split(y)
    wn, wt = hb.weight_planes(w)
    passes.append((yp, wt, gx, H // 2, Cout, Cin, x))
K, Nu, Ncols = 64, 4000, 8192
n = Nu * Ncols
pairs = []
for _ in range(2):
    U = torch.randn(K, Nu, device=hb.device)
    V = torch.randn(K, Ncols, device=hb.device)
    x_, m_, v_ = (torch.randn(n, device=hb.device) for _ in range(3))
    v_.abs_()
    pairs.append((U, V, x_, m_, v_))
t_dev = hb.zeros(2, dtype=torch.int32)
hb.adam_prep(2e-4, 0.5, 0.999, t_dev)


def chain():
    for _ in range(REPS):
        for yp, wt, gx, h, Cout, Cin, x in passes:
            hb.pconv_scatter(yp, wt, None, gx, B, h, h, Cout, Cin, dmask=x, dact="lrelu", dslope=0.2)


def update(b):
    for U, V, x_, m_, v_ in pairs:
        b.wgrad_adam_outer(U, V, x_, m_, v_, None, 0.5, 0.999, 1e-8, t_dev)


def both():
    with side.on():
        update(side)
    chain()
    side.join()


def timeit(fn, nb=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(nb):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / nb * 1e3


tc = timeit(chain)
tu = timeit(lambda: update(hb))
ts = timeit(lambda: (chain(), update(hb)))
tb = timeit(both)
print("data-gradient chain x%d: %.1f us   update pair: %.1f us   one after the other: %.1f us   side by side: %.1f us   (ideal %.1f)"
      % (REPS, tc, tu, ts, tb, max(tc, tu)))
