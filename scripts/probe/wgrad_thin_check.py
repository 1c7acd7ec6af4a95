"""k_wgrad_thin (VF_NO_THIN_WGRAD=0, default) against the implicit-GEMM weight gradient (=1 in another process): values against numpy in
double and timing, on the three image-side layers of configs[1].   python scripts/probe/wgrad_thin_check.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from video_filler_amd.backend import get_backend

hb = get_backend()


def timeit(fn, nb=50):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(nb):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / nb * 1e3


# conv 3 -> 64 (E1 at B = 64, 128 x 128; netD's first conv at 2B, 64 x 64) and the full-conv 64 -> 3 (32 -> 64)
for name, B, H in (("E1", 64, 128), ("C1@2B", 128, 64), ("small", 2, 32)):
    g = torch.Generator().manual_seed(3)
    x = torch.randn((B, 3, H, H), generator=g).to(hb.device).contiguous(memory_format=torch.channels_last)
    gy = torch.randn((B, 64, H // 2, H // 2), generator=g).to(hb.device).contiguous(memory_format=torch.channels_last)
    gw = hb.zeros(64, 4, 4, 3).permute(0, 3, 1, 2)
    hb.conv2d_bwd_weight(x, gy, gw, None, 4, 2, 1, 0.0)
    if B <= 2:
        import torch.nn.functional as F
        want = torch.nn.grad.conv2d_weight(x.double().cpu().contiguous(), (64, 3, 4, 4), gy.double().cpu().contiguous(), stride=2, padding=1)
        err = float((gw.cpu().double() - want).abs().max() / want.abs().max())
        print("%-6s conv dW against torch double: %.2e" % (name, err))
    gw2 = gw.clone()
    hb.conv2d_bwd_weight(x, gy, gw2, None, 4, 2, 1, 1.0)          # beta = 1: accumulates
    print("%-6s beta=1 doubles it: %.2e" % (name, float((gw2 - 2 * gw).abs().max() / gw.abs().max())))
    t = timeit(lambda: hb.conv2d_bwd_weight(x, gy, gw, None, 4, 2, 1, 0.0))
    print("%-6s conv dW  B=%3d %dx%d: %7.1f us" % (name, B, H, H, t))
for name, B, H in (("D5", 64, 32), ("small", 2, 16)):
    g = torch.Generator().manual_seed(4)
    x = torch.randn((B, 64, H, H), generator=g).to(hb.device).contiguous(memory_format=torch.channels_last)
    gy = torch.randn((B, 3, 2 * H, 2 * H), generator=g).to(hb.device).contiguous(memory_format=torch.channels_last)
    gw = hb.zeros(64, 4, 4, 3).permute(0, 3, 1, 2)                 # full-conv weight [Cin][Cout][kh][kw] logical, [Cin][kh][kw][Cout] physical
    hb.deconv2d_bwd_weight(x, gy, gw, None, 4, 2, 1, 0.0)
    if B <= 2:
        want = torch.nn.grad.conv2d_weight(gy.double().cpu().contiguous(), (64, 3, 4, 4), x.double().cpu().contiguous(), stride=2, padding=1)
        err = float((gw.cpu().double() - want).abs().max() / want.abs().max())
        print("%-6s full-conv dW against torch double: %.2e" % (name, err))
    t = timeit(lambda: hb.deconv2d_bwd_weight(x, gy, gw, None, 4, 2, 1, 0.0))
    print("%-6s full-conv dW B=%3d %dx%d: %7.1f us" % (name, B, H, H, t))
