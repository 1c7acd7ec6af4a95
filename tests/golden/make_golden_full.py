"""Full-width golden vectors: ONE training iteration of each reference driver at the net sizes BASELINE.json names,
computed by the CPU oracle in the build container (minutes of CPU time, so they are committed, not recomputed on the GPU
box):

    python tests/golden/make_golden_full.py [center8 vid16 wholeim]

  center8  configs[0]: train.lua, README recipe (nBottleneck=4000 wtl2=0.999 overlapPred=4), fineSize 128, batchSize 8
  vid16    configs[2]: train_vid_weighted.lua, predLen 16 (48 channels), nBottleneck 4000 (batchSize 4 of its 16 here)
  wholeim  configs[4]: train_wholeim_input.lua defaults (27 -> 12 channels, nef = ngf = 192, ndf = 128, nBottleneck 6400),
           wtgdl = 0.5 so that the GDL value path runs (batchSize 4)

Stored: the four loss scalars, every STRIDE-th entry (+ double-precision sum and sum of squares) of both gradient
vectors, both parameter vectors and Adam's first and second moments (m, v) after the Adam steps, and of the generator's output; plus a strided sample of the
INITIAL parameters so that a reader can check it rebuilt the same weights.  Inputs are regenerated from seeds
(`tests/helpers.py`: FastRng / fast_init_flat; oracle.synth_*_batch), so the files hold results only.
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(HERE))
from helpers import FastRng  # noqa: E402
from oracle import oracle as O  # noqa: E402

STRIDE = 4999      # prime: samples walk every tensor of the flat vectors at changing positions
CONFIGS = {
    "center8": dict(kind="center", B=8, wseed=11, bseed=12, opt=dict(nBottleneck=4000, wtl2=0.999, overlapPred=4)),
    "vid16": dict(kind="vid", B=4, wseed=21, bseed=22, nc_in=48, nc_out=48, opt=dict(nBottleneck=4000, predLen=16)),
    "wholeim": dict(kind="vid", B=4, wseed=31, bseed=32, nc_in=27, nc_out=12,
                    opt=dict(nBottleneck=6400, nc_in=27, nc_out=12, nef=192, ngf=192, ndf=128, weight_nomask=1, wtgdl=0.5)),
}


def batch_of(cfg):
    rng = np.random.default_rng(cfg["bseed"])
    if cfg["kind"] == "center":
        return (O.synth_center_batch(cfg["B"], rng),)
    return O.synth_vid_batch(cfg["B"], rng, cfg["nc_in"], cfg["nc_out"])


def summarize(v):
    v = np.asarray(v).reshape(-1)
    d = v.astype(np.float64)
    return v[::STRIDE].copy(), np.array([d.sum(), (d * d).sum()])


def make(name):
    cfg = CONFIGS[name]
    t0 = time.time()
    cls = O.CenterTrainer if cfg["kind"] == "center" else O.VidTrainer
    tr = cls(cfg["opt"], FastRng(cfg["wseed"]))
    out = {"pG_init_sample": tr.parametersG[::STRIDE].copy(), "pD_init_sample": tr.parametersD[::STRIDE].copy()}
    tr.set_batch(*batch_of(cfg))
    r = tr.step()
    out["losses"] = np.array([r["errD"], r["errG"], r["errG_l2"], r.get("errG_gdl") or 0.0], np.float64)
    for nm, vec in (("gG", tr.gradParametersG), ("gD", tr.gradParametersD), ("pG", tr.parametersG), ("pD", tr.parametersD),
                    ("fake", tr.netG.output), ("mG", tr.optimStateG["m"]), ("vG", tr.optimStateG["v"]), ("mD", tr.optimStateD["m"]),
                    ("vD", tr.optimStateD["v"])):
        out[nm + "_sample"], out[nm + "_sums"] = summarize(vec)
    out["n_params"] = np.array([tr.parametersG.size, tr.parametersD.size], np.int64)
    path = os.path.join(HERE, "full_%s.npz" % name)
    np.savez_compressed(path, **out)
    print("%s: %.0f s, losses %s, %d bytes" % (name, time.time() - t0, out["losses"], os.path.getsize(path)), flush=True)


if __name__ == "__main__":
    O.set_num_threads(int(os.environ.get("VF_ORACLE_THREADS", "8")))
    for n in (sys.argv[1:] or list(CONFIGS)):
        make(n)
