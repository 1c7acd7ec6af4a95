"""debug: several identical trainers in one process, stepped in turn without host synchronisation; every one must walk the same
trajectory bit for bit.  Run several copies at once to share the GPU (scripts/probe/mt_loop.sh).  On the first mismatch: which
tensors differ."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from video_filler_amd.trainers import CenterTrainer, VidTrainer
# VF_PROBE_NET: small (default: train.lua's nets at nef 32, batch 4) | center (BASELINE configs[1], full width, batch 64) |
#               vid16 (configs[2], batch 16) | wholeim (configs[4], batch 4)
NET = os.environ.get("VF_PROBE_NET", "small")
gen = torch.Generator().manual_seed(7)
if NET in ("small", "center"):
    b = 4 if NET == "small" else 64
    opt = (dict(nBottleneck=128, wtl2=0.999, overlapPred=4, nef=32, ngf=32, ndf=32, batchSize=b, smooth=True) if NET == "small"
           else dict(nBottleneck=4000, wtl2=0.999, overlapPred=4, batchSize=b))
    batch = ((torch.rand((b, 3, 128, 128), generator=gen) * 2 - 1).contiguous(),)
    cls = CenterTrainer
else:
    b, nci, nco = (16, 48, 48) if NET == "vid16" else (4, 27, 12)
    opt = (dict(batchSize=b, nBottleneck=4000, predLen=16) if NET == "vid16"
           else dict(batchSize=b, nc_in=27, nc_out=12, nef=192, ngf=192, ndf=128, nBottleneck=6400, weight_nomask=1, wtgdl=0.5))
    full = torch.rand((b, nco, 128, 128), generator=gen) * 2 - 1
    mask = torch.zeros((b, nco, 128, 128), dtype=torch.uint8)
    mask[:, :, 32:96, 32:96] = 1
    ctx = (torch.rand((b, nci, 128, 128), generator=gen) * 2 - 1) if nci != nco else full.clone()
    if nci == nco:
        ctx[mask != 0] = 2 * (110.0 / 255.0) - 1
    batch = (ctx.contiguous(), full.contiguous(), mask)
    cls = VidTrainer
def make():
    tr = cls(opt, seed=11, host=os.environ.get("VF_PROBE_HOST", "cabi"))
    tr.fuse_adam = os.environ.get("VF_PROBE_FUSE", "off")
    tr.set_batch(*batch)
    return tr
trs = [make() for _ in range(int(os.environ.get("VF_PROBE_TRAINERS", "4")))]
bad = 0
# taps: what fGx feeds into netG:backward, and what netD:updateGradInput returned before the reconstruction gradient was mixed in
taps = {id(t): {} for t in trs}
for t in trs:
    def wrap(t=t):
        bg, gi = t._backward_G, t._netD_grad_input
        def _backward_G(df_dg):
            taps[id(t)]["df_dg"] = df_dg.clone()
            return bg(df_dg)
        def _netD_grad_input(x, df_do):
            taps[id(t)]["df_do"] = df_do.clone()
            taps[id(t)]["stale_out"] = t._netD_stale_output().clone()
            r = gi(x, df_do)
            taps[id(t)]["adv"] = (r[1] if isinstance(r, (list, tuple)) else r).clone()
            return r
        t._backward_G, t._netD_grad_input = _backward_G, _netD_grad_input
    wrap()
def segs(net, flat_a, flat_b):
    out = []
    for i, (m, name, gname, o, n) in enumerate(net._flat[2]):
        d = float((flat_a[o:o + n] - flat_b[o:o + n]).abs().max())
        if d > 0:
            out.append("%d:%s.%s %.2e" % (i, type(m).__name__[:8], name, d))
    return out
for it in range(int(os.environ.get("VF_PROBE_ITERS", "6"))):
    snaps = []
    for t in trs:
        t.step()
        snaps.append((t.gradParametersD.clone(), t.gradParametersG.clone(), t.netG.output.clone()))
    torch.cuda.synchronize()
    same = [torch.equal(trs[0].parametersG, t.parametersG) and torch.equal(trs[0].parametersD, t.parametersD) for t in trs]
    if not all(same):
        bad += 1
        k = same.index(False)
        ref = 0 if k != 0 else 1
        if k == 1 and not same[2]:      # everybody differs from trainer 0: trainer 0 is the odd one
            k, ref = 0, 1
        print("it %d MISMATCH same=%s  odd trainer %d" % (it, same, k), flush=True)
        print("   fake equal: %s" % torch.equal(snaps[k][2], snaps[ref][2]))
        if os.environ.get("VF_PROBE_HOST") == "mirror":
            mk, mr = trs[k].netG.leaves(), trs[ref].netG.leaves()
            for i, (a, bq) in enumerate(zip(mk, mr)):
                gi_a, gi_b = getattr(a, "gradInput", None), getattr(bq, "gradInput", None)
                if torch.is_tensor(gi_a) and torch.is_tensor(gi_b) and gi_a.shape == gi_b.shape:
                    d = float((gi_a - gi_b).abs().max())
                    print("   module %2d %-28s gradInput maxdiff %.3e (max %.3e) %s" % (i, a.type_name()[:28], d, float(gi_b.abs().max()), tuple(gi_b.shape)))
        for nm in ("stale_out", "df_do", "adv", "df_dg"):
            a, bq = taps[id(trs[k])][nm], taps[id(trs[ref])][nm]
            print("   %s equal: %s  maxdiff %.3e  max %.3e" % (nm, torch.equal(a, bq), float((a - bq).abs().max()), float(bq.abs().max())))
        print("   gradD segments: %s" % segs(trs[k].netD, snaps[k][0], snaps[ref][0])[:12])
        sg = segs(trs[k].netG, snaps[k][1], snaps[ref][1]); print("   gradG segments (%d of %d differ), the last ones: %s" % (len(sg), len(trs[k].netG._flat[2]), sg[-14:]))
        break
print("pid %d done, %d mismatching iterations" % (os.getpid(), bad), flush=True)
try:                                    # a CHECK build of scripts/probe/rowdot_variants.hip: what its per-launch checker logged
    from video_filler_amd import _lib
    _lib.load().vf_probe_rowdot_report()
except AttributeError:
    pass
