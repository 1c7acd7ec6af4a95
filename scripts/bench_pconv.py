"""Single-pass timing of the planes GEMM (vf_pgemm.hip) beside the in-kernel-split kernels (vf_conv.hip) on the layers of
BASELINE.json configs[1]: back-to-back launches, one event pair around 50 of them.   python scripts/bench_pconv.py [B]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from video_filler_amd.backend import get_backend

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
hb = get_backend()
# (name, gathered channels, gather grid H, output channels)   conv-like: x [B][H][H][Cin] -> [B][H/2][H/2][Cout]
LAYERS = [("E2", 64, 64, 64), ("E3", 64, 32, 128), ("E4", 128, 16, 256), ("E5", 256, 8, 512), ("C1@2B", 64, 32, 128),
          ("C2@2B", 128, 16, 256), ("C3@2B", 256, 8, 512)]


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
    y = hb.empty_act(Bn, Cout, H // 2, H // 2).normal_()
    gx = hb.empty_act(Bn, Cin, H, H)
    w = (hb.empty(Cout, 4, 4, Cin).normal_() * 0.02).permute(0, 3, 1, 2)
    xp, yp = hb.planes_split(x), hb.planes_split(y)
    wn, wt = hb.weight_planes(w)
    fl = 2.0 * Bn * (H // 2) ** 2 * Cout * Cin * 16
    t_old = timeit(lambda: hb.conv2d_fwd(x, w, None, y, 4, 2, 1))
    t_new = timeit(lambda: hb.pconv_gather(xp, wn, None, y, Bn, H, H, Cin, Cout))
    hb.pconv_set_routing(gather_patch=0)
    t_dma = timeit(lambda: hb.pconv_gather(xp, wn, None, y, Bn, H, H, Cin, Cout))
    hb.pconv_set_routing(gather_patch=1)
    print("%-6s gather  B=%3d  old %7.1f us %6.1f TF   planes %7.1f us %6.1f TF   x%.2f   (tap-staged k_pconv_dma %7.1f us %6.1f TF)" % (name, Bn, t_old, fl / t_old / 1e6, t_new, fl / t_new / 1e6, t_old / t_new, t_dma, fl / t_dma / 1e6))
    # the gather pass as the iteration runs it: forward with bias + LeakyReLU + BatchNorm forward sums; data-gradient with mask + backward sums
    smg = hb.zeros(Cout)
    partg = hb.zeros((max(Bn * (H // 2) ** 2 // 64, 512) + 8) * 2 * Cout, dtype=torch.float64)
    bias_g = hb.zeros(Cout)
    def g_fwd():
        hb.bn_fuse_next_fwd(smg, partg, 1)
        hb.pconv_gather(xp, wn, bias_g, y, Bn, H, H, Cin, Cout)
        hb.bn_fuse_result()
    def g_bwd():
        hb.bn_fuse_next_bwd(y, y, "relu", 0.0, smg, partg, 1)
        hb.pconv_gather(xp, wn, None, y, Bn, H, H, Cin, Cout)
        hb.bn_fuse_result()
    res = []
    for route in (1, 0):
        hb.pconv_set_routing(gather_patch=route)
        res += [timeit(g_fwd), timeit(g_bwd)]
    hb.pconv_set_routing(gather_patch=1)
    print("%-6s gather + forward sums %7.1f us (dma %7.1f)   + mask + backward sums %7.1f us (dma %7.1f)" % (name, res[0], res[2], res[1], res[3]))
    t_old = timeit(lambda: hb.conv2d_bwd_data(y, w, gx, 4, 2, 1))
    t_new = timeit(lambda: hb.pconv_scatter(yp, wt, None, gx, Bn, H // 2, H // 2, Cout, Cin))
    print("%-6s scatter B=%3d  old %7.1f us %6.1f TF   planes %7.1f us %6.1f TF   x%.2f" % (name, Bn, t_old, fl / t_old / 1e6, t_new, fl / t_new / 1e6, t_old / t_new))
    # the data-gradient as the iteration runs it: with the derivative mask of the activation below (conv -> LeakyReLU -> conv), and
    # with the BatchNorm-backward sums + mask (conv -> BatchNorm -> LeakyReLU -> conv)
    t_m = timeit(lambda: hb.pconv_scatter(yp, wt, None, gx, Bn, H // 2, H // 2, Cout, Cin, dmask=x, dact="lrelu", dslope=0.2))
    sm = hb.zeros(Cin)
    part = hb.zeros((max(Bn * H * H // 64, 512) + 8) * 2 * Cin, dtype=torch.float64)
    def with_stats():
        hb.bn_fuse_next_bwd(x, x, "lrelu", 0.2, sm, part, 1)
        hb.pconv_scatter(yp, wt, None, gx, Bn, H // 2, H // 2, Cout, Cin)
        hb.bn_fuse_result()
    t_s = timeit(with_stats)
    print("%-6s scatter + mask %7.1f us %6.1f TF   + BatchNorm-backward sums %7.1f us %6.1f TF" % (name, t_m, fl / t_m / 1e6, t_s, fl / t_s / 1e6))
t = timeit(lambda: hb.planes_split(x, xp))
print("planes_split of %d elements: %.1f us (%.0f GB/s)" % (x.numel(), t, 10.0 * x.numel() / t / 1e3))
