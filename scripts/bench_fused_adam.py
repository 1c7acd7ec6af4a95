"""The fused bottleneck update alone (vf_wgrad_adam_outer): K = batch rows, [Nu][Ncols] weights; 24 B per weight.
   python scripts/bench_fused_adam.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from video_filler_amd.backend import get_backend

hb = get_backend()


def timeit(fn, nb=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(nb):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / nb * 1e3


for K, Nu, Ncols in ((64, 4000, 8192), (16, 4000, 8192), (4, 6400, 24576), (512, 4000, 8192)):
    n = Nu * Ncols
    U = torch.randn(K, Nu, device=hb.device)
    V = torch.randn(K, Ncols, device=hb.device)
    x, m, v = (torch.randn(n, device=hb.device) for _ in range(3))
    v.abs_()
    t_dev = hb.zeros(2, dtype=torch.int32)
    hb.adam_prep(2e-4, 0.5, 0.999, t_dev)
    t = timeit(lambda: hb.wgrad_adam_outer(U, V, x, m, v, None, 0.5, 0.999, 1e-8, t_dev))
    print("K=%3d %5d x %5d: %7.1f us  %.2f TB/s" % (K, Nu, Ncols, t, 24.0 * n / t / 1e6))

# data parallel, 8 ranks x batchSize 64: operands gathered in 8 segments
K, Nu, Ncols, world = 64, 4000, 8192, 8
n = Nu * Ncols
seg = K * Nu + K * Ncols
buf = torch.randn(world * seg, device=hb.device)
x, m, v = (torch.randn(n, device=hb.device) for _ in range(3))
v.abs_()
t_dev = hb.zeros(2, dtype=torch.int32)
hb.adam_prep(2e-4, 0.5, 0.999, t_dev)
t = timeit(lambda: hb.wgrad_adam_outer_gathered(buf, 0, K * Nu, world, K, seg, Nu, Ncols, x, m, v, None, 0.5, 0.999, 1e-8, t_dev))
print("gathered %d x %d rows: %7.1f us  %.1f TFLOP/s" % (world, K, t, 2.0 * world * K * n / t / 1e6))
