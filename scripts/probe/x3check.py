import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
from oracle import oracle as O
import video_filler_amd
from video_filler_amd.backend import get_backend
from helpers import to_dev, to_np, rel_err
hb = get_backend()
for mode in ("f32", "f32_3xbf16", "bf16"):
    hb.set_mfma_mode(mode)
    errs = []
    for (full, B, Cin, H, Cout, s, p) in [(False, 4, 64, 16, 64, 2, 1), (False, 2, 128, 4, 100, 1, 0), (True, 2, 128, 4, 64, 2, 1), (False, 8, 256, 8, 512, 2, 1), (True, 4, 512, 4, 256, 2, 1)]:
        rng = np.random.default_rng(B * 7 + Cin)
        r = lambda *sh: rng.standard_normal(sh).astype(np.float32)
        m = (O.SpatialFullConvolution if full else O.SpatialConvolution)(Cin, Cout, 4, 4, s, s, p, p)
        m.weight[...] = r(*m.weight.shape) * 0.05; m.bias[...] = r(Cout)
        x = r(B, Cin, H, H); y = np.array(m.forward(x), copy=True); gy = r(*y.shape)
        gx = np.array(m.updateGradInput(x, gy), copy=True)
        fwd, bwd = (hb.deconv2d_fwd, hb.deconv2d_bwd_data) if full else (hb.conv2d_fwd, hb.conv2d_bwd_data)
        dy = hb.empty_act(*y.shape); fwd(to_dev(x, hb), to_dev(m.weight, hb), to_dev(m.bias, hb), dy, 4, s, p)
        dgx = hb.empty_act(*x.shape); bwd(to_dev(gy, hb), to_dev(m.weight, hb), dgx, 4, s, p)
        errs.append((rel_err(to_np(dy), y), rel_err(to_np(dgx), gx)))
    print(mode, " ".join("%.1e/%.1e" % e for e in errs))
hb.set_mfma_mode("f32")
