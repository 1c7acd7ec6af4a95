"""Weight gradients of the bottleneck pair alone (conv C8 -> nB on a 4x4 map, full-conv nB -> C8 onto a 4x4 map): K = batch, the
131 MB / 629 MB output write is the job.  Compares against a memset of the same bytes.
   python scripts/bench_bottleneck_wgrad.py [B [nB C8]]        VF_NO_WGRAD_SMALLK=1 keeps the tiled kernel of vf_conv.hip"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from video_filler_amd.backend import get_backend

hb = get_backend()


def timeit(fn, nb=30):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(nb):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / nb * 1e3


B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
nB, C8 = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (4000, 512)
x = hb.empty_act(B, C8, 4, 4).normal_()
gy = hb.empty_act(B, nB, 1, 1).normal_()
gw = hb.zeros(nB, 4, 4, C8).permute(0, 3, 1, 2)
t = timeit(lambda: hb.conv2d_bwd_weight(x, gy, gw, None, 4, 1, 0, 0.0))
print("B=%d  conv %d -> %d  dW: %.1f us, %.2f TB/s written" % (B, C8, nB, t, gw.numel() * 4 / t / 1e6))
x2 = hb.empty_act(B, nB, 1, 1).normal_()
gy2 = hb.empty_act(B, C8, 4, 4).normal_()
gw2 = hb.zeros(nB, 4, 4, C8).permute(0, 3, 1, 2)
t = timeit(lambda: hb.deconv2d_bwd_weight(x2, gy2, gw2, None, 4, 1, 0, 0.0))
print("B=%d  full-conv %d -> %d  dW: %.1f us, %.2f TB/s written" % (B, nB, C8, t, gw2.numel() * 4 / t / 1e6))
z = torch.empty(gw.numel(), device=hb.device)
t = timeit(lambda: z.zero_())
print("memset of the same size: %.1f us, %.2f TB/s" % (t, z.numel() * 4 / t / 1e6))
