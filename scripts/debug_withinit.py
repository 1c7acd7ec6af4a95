"""Per-tensor gradient errors of the withInit variant (real nets) against the oracle, with the ReLU kinks pinned."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import oracle as O
from helpers import KinkSync, rel_err, to_np
import video_filler_amd
from video_filler_amd.trainers import VidTrainer, build_netG

for delta in (1e-5, 1e-3):
    opt = dict(nBottleneck=64, predLen=1, smooth=False)
    ref = O.VidTrainer(opt, np.random.default_rng(2))
    tr = VidTrainer(opt)
    tr.set_batch_d(False)
    dev = tr.parametersG.device
    tr.netG.load_reference_flat(torch.from_numpy(ref.parametersG.copy()).to(dev))
    tr.netD.load_reference_flat(torch.from_numpy(ref.parametersD.copy()).to(dev))
    rI = O.build_netG(3, 3, 16, 16, 32, True, False)
    O.weights_init(rI, np.random.default_rng(8))
    pI, _ = rI.getParameters()
    hI = build_netG(3, 3, 16, 16, 32, True, smooth=False)
    hI.getParameters()
    hI.load_reference_flat(torch.from_numpy(pI.copy()).to(dev))
    ref.netI = rI
    tr.set_initializer(hI)
    ks = KinkSync(O, [(ref.netG, tr.netG), (ref.netD, tr.netD), (rI, hI)], delta=delta)
    ctx, full, mask = O.synth_vid_batch(4, np.random.default_rng(20), 3, 3)
    ref.set_batch(ctx, full, mask)
    tr.set_batch(torch.from_numpy(ctx), torch.from_numpy(full), torch.from_numpy(mask))
    ks.oracle_step(ref.step)
    ks.hip_step(tr.step)
    print("delta", delta, "touched", ks.touched, "of", ks.checked)
    print(" netI out err", rel_err(to_np(hI.output), rI.output), " filled ctx err", rel_err(to_np(tr._ctx_filled), ref.input_ctx))
    print(" fake err", rel_err(to_np(tr.netG.output), ref.netG.output), "df_dg err", rel_err(to_np(tr.netG.modules[-1].gradInput if False else tr.netD.gradInput) if tr.netD.gradInput is not None else 0, ref.df_dg) if False else "")
    g = to_np(tr.netG.reference_flat(grads=True)); gr = ref.gradParametersG
    print(" gradG total", rel_err(g, gr), " gradD", rel_err(to_np(tr.netD.reference_flat(grads=True)), ref.gradParametersD))
    off = 0
    for m, name, gname, o, n in tr.netG._flat[2]:
        a, b = g[off:off + n], gr[off:off + n]
        print("   %-34s %-6s n=%8d  |ref|max %.3e  err/max %.3e  err/globalmax %.3e" % (m.type_name(), name, n, np.abs(b).max(), np.abs(a - b).max() / (np.abs(b).max() + 1e-30), np.abs(a - b).max() / np.abs(gr).max()))
        off += n
