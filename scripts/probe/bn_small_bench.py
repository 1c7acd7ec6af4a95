"""One-launch BatchNorm for small tensors (k_bn_small_*) beside the three-launch form, forward and backward, back-to-back launches.
python scripts/probe/bn_small_bench.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from video_filler_amd.backend import get_backend

hb = get_backend()      # (needs scripts/probe/bn_small_one_launch.patch applied: bn_set_small_rows)
SHAPES = [(64, 4000, 1, 1, 1), (64, 512, 4, 4, 1), (128, 512, 4, 4, 2), (64, 512, 4, 4, 1), (16, 4000, 1, 1, 1), (16, 512, 4, 4, 1),
          (16, 256, 8, 8, 1), (4, 6400, 1, 1, 1), (4, 768, 4, 4, 1)]


def timeit(fn, nb=200):
    for _ in range(10):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(nb):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / nb * 1e3


for B, C, H, W, G in SHAPES:
    x, gy = hb.empty_act(B, C, H, W).normal_(), hb.empty_act(B, C, H, W).normal_()
    y, gx = hb.empty_act(B, C, H, W), hb.empty_act(B, C, H, W)
    gamma, beta = hb.zeros(C) + 1, hb.zeros(C)
    rm, rv, sm, si = hb.zeros(C), hb.zeros(C) + 1, hb.zeros(G * C), hb.zeros(G * C)
    sums = hb.zeros(G * 2 * C, dtype=torch.float64)
    gg, gb = hb.zeros(C), hb.zeros(C)
    row = []
    for limit in (0, 1024):
        hb.bn_set_small_rows(limit)
        row.append(timeit(lambda: hb.bn_train_fwd_groups(x, y, gamma, beta, rm, rv, sm, si, sums, G, 0.1, 1e-5, "lrelu", 0.2)))
        row.append(timeit(lambda: hb.bn_bwd_groups(x, y, gy, gx, gg, gb, gamma, sm, si, sums, G, "lrelu", 0.2, 0.0)))
    print("B %3d C %4d %dx%d groups %d (%4d px/group, %5.2f MB): forward three %6.1f us one %6.1f | backward three %6.1f one %6.1f" % (
        B, C, H, W, G, B // G * H * W, B * C * H * W * 4 / 1e6, row[0], row[2], row[1], row[3]))
