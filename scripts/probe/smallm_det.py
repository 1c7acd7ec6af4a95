"""probe: the row-dot small-M pass repeated in place on the same operands must give the same bits every time, also with other
processes on the GPU.  Compares the split-K SLABS the row-dot kernel wrote (the head of the context's workspace), the combined
output y, and — with a DIAG build of scripts/probe/rowdot_variants.hip — the per-lane partial sums in front of the cross-lane
reduction, so that a mismatch says which stage produced it:
    partials differ            -> the loads / FMAs (operands arrived wrong or late)
    partials equal, slab not   -> the cross-lane reduction or the store
    slab equal, y not          -> the combine kernel
python scripts/probe/smallm_det.py           (VF_HIP_LIB=<variant library>; scripts/probe/rowdot_variants.sh runs the set)"""
import ctypes
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from video_filler_amd.backend import get_backend
hb = get_backend()
g = torch.Generator().manual_seed(3)
Bn, nB, C8 = 4, 128, 256
KS = 8                                   # vf_internal_smallm_plan for M = 4, N = 128, K = 4096
x = torch.randn(Bn, 4, 4, C8, generator=g).to(hb.device).permute(0, 3, 1, 2)
w = (torch.randn(nB, 4, 4, C8, generator=g) * 0.05).to(hb.device).permute(0, 3, 1, 2)
y = hb.empty_act(Bn, nB, 1, 1)
slab = hb.workspace[:KS * Bn * nB * 4].view(torch.float32)
dbg = None
try:
    fn = hb.lib.vf_probe_rowdot_dbg
    fn.restype = ctypes.c_void_p
    var = hb.lib.vf_probe_rowdot_variant()
    if var % 10 == 1:
        import numpy as np
        ptr = fn()
        n = 128 * 32 * 64
        # a torch view of the library's static device buffer
        class _A:       # __cuda_array_interface__ holder
            pass
        a = _A()
        a.__cuda_array_interface__ = dict(shape=(n,), typestr="<f4", data=(ptr, False), version=2)
        dbg = torch.as_tensor(a, device=hb.device)
    print("variant", var, "diag" if dbg is not None else "", flush=True)
except AttributeError:
    print("shipped library", flush=True)
slab.zero_()
hb.conv2d_fwd(x, w, None, y, 4, 1, 0)
torch.cuda.synchronize()
ref_y, ref_slab = y.clone(), slab.clone()
ref_dbg = dbg.clone() if dbg is not None else None
assert float(ref_slab.abs().max()) > 0, "the pass did not go through the row-dot kernel (slabs untouched)"
bad = 0
filler = torch.randn(1 << 22, device=hb.device)
N = int(os.environ.get("N", "20000"))
for i in range(N):
    if i % 3 == 0:
        filler.mul_(1.0001)            # (something else in flight)
    y.zero_()
    slab.zero_()
    hb.conv2d_fwd(x, w, None, y, 4, 1, 0)
    es, ey = torch.equal(slab, ref_slab), torch.equal(y, ref_y)
    if not (es and ey):
        bad += 1
        if bad <= 8:
            ds = torch.nonzero(slab != ref_slab).reshape(-1).tolist()
            where = [(j // (Bn * nB), (j // nB) % Bn, j % nB) for j in ds[:8]]        # (ks, b, n)
            vals = [(float(ref_slab[j]), float(slab[j])) for j in ds[:8]]
            msg = "run %d: slab equal %s (%d of %d differ) y equal %s; (ks,b,n) %s; (ref, got) %s" % (
                i, es, len(ds), slab.numel(), ey, where, ["%.6g/%.6g" % v for v in vals])
            if dbg is not None:
                dd = torch.nonzero(dbg != ref_dbg).reshape(-1)
                msg += "; per-lane partials equal %s (%d differ)" % (dd.numel() == 0, dd.numel())
                for j in ds[:4]:
                    ks, b, n = j // (Bn * nB), (j // nB) % Bn, j % nB
                    wv = ks * (nB // 8) + n // 8
                    v = (n % 8) * 4 + b
                    lanes = ref_dbg[(wv * 32 + v) * 64:(wv * 32 + v) * 64 + 64].double()
                    diff = float(slab[j]) - float(ref_slab[j])
                    got_l = dbg[(wv * 32 + v) * 64:(wv * 32 + v) * 64 + 64]
                    msg += " | value (wave %d, v %d): diff %.6g, lane sum %.6g, lanes(ref) %s lanes(got) %s" % (
                        wv, v, diff, float(lanes.sum()), [round(float(q), 5) for q in lanes], [round(float(q), 5) for q in got_l])
            print(msg, flush=True)
print("pid %d: %d mismatching runs of %d" % (os.getpid(), bad, N), flush=True)
