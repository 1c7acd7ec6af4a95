import os, sys
sys.path.insert(0, "/root/repo")
import torch
from video_filler_amd.backend import get_backend
hb = get_backend()
def timeit(fn, nb=30):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(nb): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / nb * 1e3
B=int(sys.argv[1]) if len(sys.argv) > 1 else 64
NB_, C8 = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (4000, 512)
# bottleneck conv 512 -> 4000, 4x4 s1 p0 on a 4x4 map
x = hb.empty_act(B, C8, 4, 4).normal_(); gy = hb.empty_act(B, NB_, 1, 1).normal_()
gw = hb.zeros(NB_, 4, 4, C8).permute(0, 3, 1, 2)
t = timeit(lambda: hb.conv2d_bwd_weight(x, gy, gw, None, 4, 1, 0, 0.0))
print("B=%d %d<->%d  " % (B, C8, NB_) + "E6 dW (K=batch): %.1f us, %.2f TB/s written" % (t, gw.numel()*4/t/1e6))
# full-conv 4000 -> 512, 1x1 -> 4x4
x2 = hb.empty_act(B, NB_, 1, 1).normal_(); gy2 = hb.empty_act(B, C8, 4, 4).normal_()
gw2 = hb.zeros(NB_, 4, 4, C8).permute(0, 3, 1, 2)
t = timeit(lambda: hb.deconv2d_bwd_weight(x2, gy2, gw2, None, 4, 1, 0, 0.0))
print("D1 dW (4000->512): %.1f us, %.2f TB/s written" % (t, gw2.numel()*4/t/1e6))
z = torch.empty(gw.numel(), device=hb.device)
t = timeit(lambda: z.zero_())
print("memset of the same size: %.1f us, %.2f TB/s" % (t, z.numel()*4/t/1e6))
