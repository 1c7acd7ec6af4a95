import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from oracle import oracle as O
from video_filler_amd.trainers import CenterTrainer
from helpers import to_np, rel_err
from test_gpu_trainers import _leaves, _load
opt = dict(nBottleneck=64, wtl2=0.999, overlapPred=4, smooth=True)
ref = O.CenterTrainer(opt, np.random.default_rng(1))
tr = CenterTrainer(opt)
_load(tr, ref)
batch = O.synth_center_batch(3, np.random.default_rng(10))
ref.set_batch(batch); tr.set_batch(torch.from_numpy(batch))
ref.step(); tr.step()
for nm, net, rnet in (("D", tr.netD, ref.netD), ("G", tr.netG, ref.netG)):
    rb = [m for m in _leaves(rnet) if hasattr(m, "running_mean")]
    hb = [m for m in net.leaves() if hasattr(m, "running_mean")]
    for a, b in zip(rb, hb):
        print(nm, a.nOutputPlane, "rm err %.2e  rv err %.2e  |rm|max %.3e %.3e  save_mean err %.2e" % (
            rel_err(to_np(b.running_mean), a.running_mean), rel_err(to_np(b.running_var), a.running_var),
            np.abs(a.running_mean).max(), np.abs(to_np(b.running_mean)).max(), rel_err(to_np(b.save_mean), a.save_mean)))
