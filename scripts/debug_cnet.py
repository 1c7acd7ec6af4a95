"""cabi host (vf_net_*) against the mirror: same seed, same batch, N iterations -> parameters must be BITWISE equal
(same kernels in the same order).  Usage: python scripts/debug_cnet.py [center|vid] [gate]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import video_filler_amd  # noqa: E402,F401
from video_filler_amd import nn  # noqa: E402
from video_filler_amd.trainers import CenterTrainer, VidTrainer  # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else "center"
if len(sys.argv) > 2:
    nn._PCONV_MIN_GFLOP = float(sys.argv[2])
from video_filler_amd.backend import tensor_from_ptr  # noqa: E402
t = torch.arange(12, dtype=torch.float32, device="cuda")
v = tensor_from_ptr(t.data_ptr(), (3, 4), t.device)
v[1, 1] = -5
assert float(t[5]) == -5.0, "tensor_from_ptr is not a view"
print("tensor_from_ptr ok")
gen = torch.Generator().manual_seed(3)
if kind == "center":
    opt = dict(nBottleneck=64, wtl2=0.999, overlapPred=4)
    mk = lambda host: CenterTrainer(opt, seed=5, host=host, overlap=False)
    batch = (torch.rand((4, 3, 128, 128), generator=gen) * 2 - 1,)
else:
    opt = dict(nBottleneck=64, predLen=2)
    mk = lambda host: VidTrainer(opt, seed=5, host=host, overlap=False)
    full = torch.rand((4, 6, 128, 128), generator=gen) * 2 - 1
    mask = torch.zeros((4, 6, 128, 128), dtype=torch.uint8)
    mask[:, :, 32:96, 32:96] = 1
    ctx = full.clone()
    ctx[mask != 0] = -0.1373
    batch = (ctx, full, mask)
for bd in (True, False):
    a, b = mk("mirror"), mk("cabi")
    assert type(b.netG).__name__ == "CNet" and type(b.netD).__name__ == "CNet", (type(b.netG), type(b.netD))
    for t_ in (a, b):
        t_.set_batch_d(bd)
        t_.set_batch(*batch)
    for it in range(3):
        a.step()
        b.step()
        torch.cuda.synchronize()
        eg = bool(torch.equal(a.gradParametersG, b.gradParametersG))
        ed = bool(torch.equal(a.gradParametersD, b.gradParametersD))
        pg = bool(torch.equal(a.parametersG, b.parametersG))
        pd = bool(torch.equal(a.parametersD, b.parametersD))
        print(kind, "batch_d", bd, "it", it, "gradG", eg, "gradD", ed, "paramG", pg, "paramD", pd, a.losses(), b.losses())
        if not (eg and ed and pg and pd):
            dg = (a.gradParametersG - b.gradParametersG).abs().max().item()
            dd = (a.gradParametersD - b.gradParametersD).abs().max().item()
            print("   max |dgG| %.3e  max |dgD| %.3e" % (dg, dd))
            for net_a, net_b, nm in ((a.netD, b.netD, "D"), (a.netG, b.netG, "G")):
                for (m, name, gname, o, n), (m2, *_r) in zip(net_a._flat[2], net_b._flat[2]):
                    ga, gb = net_a._flat[1][o:o + n], net_b._flat[1][o:o + n]
                    if not torch.equal(ga, gb):
                        print("   ", nm, m.type_name(), name, "differs: %.3e of %.3e" % ((ga - gb).abs().max().item(), ga.abs().max().item()))
    # graph capture through the cabi host
    b.capture(warmup=2)
    for _ in range(2):
        a.step()
    a.step()
    a.step()
    b.replay()
    b.replay()
    torch.cuda.synchronize()
    print(kind, "batch_d", bd, "after capture+2 replays: paramG", bool(torch.equal(a.parametersG, b.parametersG)), "paramD", bool(torch.equal(a.parametersD, b.parametersD)))
print("done")
