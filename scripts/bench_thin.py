"""timing of the image-side conv forward (3 input channels): direct kernel vs the implicit-GEMM scalar path (VF_NO_THIN=1)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from video_filler_amd.backend import get_backend
hb = get_backend()
def timeit(fn, nb=30):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(nb): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / nb * 1e3
for Bn, H in ((64, 128), (128, 64)):
    x = hb.empty_act(Bn, 3, H, H).normal_()
    y = hb.empty_act(Bn, 64, H // 2, H // 2)
    w = (hb.empty(64, 4, 4, 3).normal_() * 0.02).permute(0, 3, 1, 2)
    b = hb.zeros(64)
    yp = torch.empty((3, y.numel()), dtype=torch.bfloat16, device=hb.device)
    print("B=%d H=%d: conv2d_fwd %.1f us   conv2d_fwd_planes %.1f us   planes_split alone %.1f us" % (
        Bn, H, timeit(lambda: hb.conv2d_fwd(x, w, b, y, 4, 2, 1, "lrelu", 0.2)),
        timeit(lambda: hb.conv2d_fwd_planes(x, w, b, y, yp, 4, 2, 1, "lrelu", 0.2)), timeit(lambda: hb.planes_split(y, yp))))
