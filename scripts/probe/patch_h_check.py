"""k_pconv_patch_h (half chunks, two blocks per CU; routing value 2) beside k_pconv_patch_g (1) and k_pconv_dma (0) on the gather passes of
configs[1] whose output maps tile by 8 x 16: values against routing 1, then back-to-back timings.   python scripts/probe/patch_h_check.py [B]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from video_filler_amd.backend import get_backend

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
hb = get_backend()
LAYERS = [("E2", 64, 64, 64), ("E3", 64, 32, 128), ("C1@2B", 64, 32, 128), ("C2@2B", 128, 16, 256), ("D4dX", 128, 32, 64)]
only = os.environ.get("ONLY", "")
LAYERS = [l for l in LAYERS if not only or l[0] in only.split(",")]


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


for name, Cin, H, Cout in LAYERS:
    Bn = 2 * B if "@2B" in name else B
    torch.manual_seed(1)
    x = hb.empty_act(Bn, Cin, H, H).normal_()
    w = (hb.empty(Cout, 4, 4, Cin).normal_() * 0.02).permute(0, 3, 1, 2)
    xp = hb.planes_split(x)
    wn, wt = hb.weight_planes(w)
    bias = hb.empty(Cout).normal_()
    msk = hb.empty_act(Bn, Cout, H // 2, H // 2).normal_()
    sm = hb.zeros(Cout).normal_()
    fl = 2.0 * Bn * (H // 2) ** 2 * Cout * Cin * 16
    out, parts = {}, {}
    for route in (1, 2):
        hb.pconv_set_routing(gather_patch=route)
        y = hb.empty_act(Bn, Cout, H // 2, H // 2).zero_()
        hb.pconv_gather(xp, wn, None, y, Bn, H, H, Cin, Cout)
        y1 = hb.empty_act(Bn, Cout, H // 2, H // 2).zero_()
        part1 = hb.zeros((max(Bn * (H // 2) ** 2 // 64, 512) + 8) * 2 * Cout, dtype=torch.float64)
        hb.bn_fuse_next_fwd(sm, part1, 1)
        hb.pconv_gather(xp, wn, bias, y1, Bn, H, H, Cin, Cout, act="lrelu", slope=0.2)
        hb.bn_fuse_result()
        y2 = hb.empty_act(Bn, Cout, H // 2, H // 2).zero_()
        part2 = hb.zeros((max(Bn * (H // 2) ** 2 // 64, 512) + 8) * 2 * Cout, dtype=torch.float64)
        hb.bn_fuse_next_bwd(msk, msk, "relu", 0.0, sm, part2, 1)
        hb.pconv_gather(xp, wn, None, y2, Bn, H, H, Cin, Cout)
        hb.bn_fuse_result()
        torch.cuda.synchronize()
        out[route] = (y.clone(), y1.clone(), y2.clone())
        parts[route] = (part1.clone(), part2.clone())
    for i, what in enumerate(("plain", "bias + lrelu + forward sums", "mask + backward sums")):
        a, b = out[1][i], out[2][i]
        print("%-6s %-30s max |d| %.3e of max %.3e   equal %s" % (name, what, (a - b).abs().max().item(), a.abs().max().item(), torch.equal(a, b)))
    for i in range(2):
        a, b = parts[1][i], parts[2][i]
        print("%-6s partial rows %d: max |d| %.3e of max %.3e" % (name, i, (a - b).abs().max().item(), a.abs().max().item()))
    y = hb.empty_act(Bn, Cout, H // 2, H // 2)
    part = hb.zeros((max(Bn * (H // 2) ** 2 // 64, 512) + 8) * 2 * Cout, dtype=torch.float64)
    def g_fwd():
        hb.bn_fuse_next_fwd(sm, part, 1)
        hb.pconv_gather(xp, wn, bias, y, Bn, H, H, Cin, Cout)
        hb.bn_fuse_result()
    def g_bwd():
        hb.bn_fuse_next_bwd(msk, msk, "relu", 0.0, sm, part, 1)
        hb.pconv_gather(xp, wn, None, y, Bn, H, H, Cin, Cout)
        hb.bn_fuse_result()
    for rep in range(2):
        row = []
        for route in (0, 1, 2):
            hb.pconv_set_routing(gather_patch=route)
            row += [timeit(lambda: hb.pconv_gather(xp, wn, None, y, Bn, H, H, Cin, Cout)), timeit(g_fwd), timeit(g_bwd)]
        print("%-6s us  dma %6.1f %6.1f %6.1f | patch_g %6.1f %6.1f %6.1f | patch_h %6.1f %6.1f %6.1f   (plain, + forward sums, + mask + backward sums)  %.0f TF" % (
            (name,) + tuple(row) + (fl / row[6] / 1e6,)))
    hb.pconv_set_routing(gather_patch=1)
