"""debug: which of the two data-parallel paths differs from run to run (see scripts/dp_rehearsal.py)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import torch.distributed as dist
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo", rank=rank, world_size=world)
from video_filler_amd.trainers import CenterTrainer
b = 4
opt = dict(nBottleneck=128, wtl2=0.999, overlapPred=4, nef=32, ngf=32, ndf=32, batchSize=b, smooth=True)
gen = torch.Generator().manual_seed(7)
full = torch.rand((world * b, 3, 128, 128), generator=gen) * 2 - 1
shard = full[rank * b:(rank + 1) * b].contiguous()
def make(mode):
    tr = CenterTrainer(opt, seed=11, world=world, rank=rank, group=None, host="cabi")
    tr.fuse_adam = mode
    tr.set_batch(shard)
    return tr
trs = {k: make(k[:-1]) for k in ("on1", "on2", "off1", "off2")}
hist = {k: [] for k in trs}
for it in range(3):
    for k, t in trs.items():
        t.step_phased()
        if os.environ.get("VF_PROBE_SYNC", "1") == "1":
            torch.cuda.synchronize()
        hist[k].append((t.parametersG.clone(), t.parametersD.clone(), float(t.errD), float(t.errG)))
def same(a, b, it):
    return (torch.equal(hist[a][it][0], hist[b][it][0]), torch.equal(hist[a][it][1], hist[b][it][1]))
for it in range(3):
    print("rank %d it %d  on1==on2 %s  off1==off2 %s  | maxdiff G on1-off1 %.3e  D %.3e | errD %s" % (
        rank, it, same("on1", "on2", it), same("off1", "off2", it),
        float((hist["on1"][it][0] - hist["off1"][it][0]).abs().max()), float((hist["on1"][it][1] - hist["off1"][it][1]).abs().max()),
        [round(hist[k][it][2], 6) for k in trs]), flush=True)
dist.barrier()
dist.destroy_process_group()
