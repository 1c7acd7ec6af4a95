"""Weight gradient from planes (k_pwgrad_group, vf_pgemm.hip) beside the fp32-operand kernel (k_wgrad, vf_conv.hip) on the layers of
BASELINE.json configs[1]: back-to-back launches, one event pair around NB of them (split-K slab reduce included on both sides).
python scripts/bench_pwgrad.py [B]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from video_filler_amd.backend import get_backend

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
hb = get_backend()
LAYERS = [("E3", 64, 32, 128), ("E4", 128, 16, 256), ("E5", 256, 8, 512), ("C1@2B", 64, 32, 128), ("C2@2B", 128, 16, 256),
          ("C3@2B", 256, 8, 512)]
only = os.environ.get("ONLY", "")
LAYERS = [l for l in LAYERS if not only or l[0] in only.split(",")]


def timeit(fn, nb=int(os.environ.get("NB", "50"))):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(nb):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / nb * 1e3


for name, Cin, H, Cout in LAYERS:
    Bn = 2 * B if "@2B" in name else B
    x = hb.empty_act(Bn, Cin, H, H).normal_()
    gy = hb.empty_act(Bn, Cout, H // 2, H // 2).normal_()
    gw = hb.zeros(Cout, 4, 4, Cin).permute(0, 3, 1, 2)
    xp, gp = hb.planes_split(x), hb.planes_split(gy)
    fl = 2.0 * Bn * (H // 2) ** 2 * Cout * Cin * 16
    t_old = timeit(lambda: hb.conv2d_bwd_weight(x, gy, gw, None, 4, 2, 1, 0.0))
    t_new = timeit(lambda: hb.conv2d_bwd_weight(x, gy, gw, None, 4, 2, 1, 0.0, xp, gp))
    print("%-6s dW  B=%3d  fp32 operands %7.1f us %6.1f TF   planes %7.1f us %6.1f TF   x%.2f" % (name, Bn, t_old, fl / t_old / 1e6, t_new, fl / t_new / 1e6, t_old / t_new))
hb.lib.vf_prof_begin(hb.ctx)
hb.conv2d_bwd_weight(x, gy, gw, None, 4, 2, 1, 0.0, xp, gp)
hb.lib.vf_prof_end(hb.ctx)
import ctypes as C
nm = C.create_string_buffer(128); a = C.c_int64(); ms = C.c_double(); f = C.c_double(); by = C.c_double()
print("kernels of the last planes call:", end=" ")
for i in range(hb.lib.vf_prof_count()):
    hb.lib.vf_prof_get(i, nm, 128, C.byref(a), C.byref(ms), C.byref(f), C.byref(by)); print(nm.value.decode(), end=" ")
print()
