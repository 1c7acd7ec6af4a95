"""Several data-parallel ranks on ONE GPU, exchanging over gloo (RCCL refuses two ranks on a device): what a one-GPU box can say
about the N > 1 iteration.  Every rank runs the train.lua nets through the C-ABI host on cuda:0 on its own shard of a batch:

  * `on`  : the phased step with the bottleneck pair's OPERANDS all-gathered and the global-batch gradient formed inside the fused
            Adam kernel on every rank (trainers._phase_b / _phase_c, vf_net_fused_adam_pack / vf_net_adam_fused_gathered);
  * `off` : the same step with that pair's gradients all-reduced like the rest;
  * `rows`: `on` with the fused update SHARDED BY WEIGHT ROWS (trainer.dp_fused = "rows", the default: rank r forms and applies its
            row block of the pair — equal blocks at 2 ranks, 68 / 68 / 64 of the 200 rows at 3 — and the exchange of the updated
            blocks opens the NEXT iteration, or flush()) — must hold the SAME BITS as `on` in every parameter after flush(), in
            Adam's moments on this rank's rows before gather_adam_state() and everywhere after it.

Checked on every rank: the two walk the same trajectory to fp32 rounding (gradients of everything that is still exchanged,
parameters where the gradient is significant), the fused slices were really left out of the exchange, and the replicas hold the
SAME BITS on every rank after three iterations (parameters, Adam moments).

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 scripts/dp_rehearsal.py
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import torch.distributed as dist

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo", rank=rank, world_size=world)

from video_filler_amd.backend import get_backend
from video_filler_amd.trainers import CenterTrainer

B = get_backend()
b = int(os.environ.get("VF_REHEARSAL_BATCH", "4"))
# (smooth nets — every (Leaky)ReLU replaced by LeakyReLU(1.0), same graph and kernels: two trajectories that differ by fp32
#  rounding must not be told apart by an activation that sits at its kink, tests/test_gpu_trainers.py)
opt = dict(nBottleneck=200, wtl2=0.999, overlapPred=4, nef=32, ngf=32, ndf=32, batchSize=b, smooth=True)
gen = torch.Generator().manual_seed(7)
full = torch.rand((world * b, 3, 128, 128), generator=gen) * 2 - 1          # the same draw on every rank; rank r takes shard r
shard = full[rank * b:(rank + 1) * b].contiguous()


def make(mode):
    tr = CenterTrainer(opt, seed=11, world=world, rank=rank, group=None, host="cabi")
    tr.fuse_adam = mode
    tr.set_batch(shard)
    return tr


on, off, rows = make("on"), make("off"), make("on")
on.dp_fused = "gathered"
rows.dp_fused = "rows"
assert torch.equal(on.parametersG, off.parametersG)
for _ in range(3):
    rows.step_phased()
    on.step_phased()
    if os.environ.get("VF_REHEARSAL_SYNC") == "1":
        torch.cuda.synchronize()
    off.step_phased()
    if os.environ.get("VF_REHEARSAL_SYNC") == "1":
        torch.cuda.synchronize()
torch.cuda.synchronize()
ranges = on.fused_adam_ranges()
assert len(ranges) == 2 and off.fused_adam_ranges() == [], (ranges, off.fused_adam_ranges())
assert on._opbuf is not None and on._opbuf.numel() == world * on._opbuf.numel() // world
n = on.parametersG.numel()
mask = torch.zeros(n, dtype=torch.bool, device=on.parametersG.device)
for lo, hi in ranges:
    mask[lo:hi] = True
    assert float(on.gradParametersG[lo:hi].abs().max()) == 0.0, "the fused slices must not have been written / exchanged"
g_on, g_off = on.gradParametersG, off.gradParametersG
err_g = float((g_on - g_off)[~mask].abs().max() / g_off.abs().max())
lr = on.optimStateG["learningRate"]
sel = g_off.abs() > 1e-3 * g_off.abs().max()
err_p = float((on.parametersG - off.parametersG).abs()[sel].max() / lr)
err_m = float((on.optimStateG["m"] - off.optimStateG["m"]).abs().max() / off.optimStateG["m"].abs().max())
err_d = float((on.parametersD - off.parametersD).abs().max() / on.optimStateD["learningRate"])


def bits(t):
    return int(t.view(torch.int32).to(torch.int64).sum().item())


# the row-sharded form against the gathered one: before flush() the other ranks' rows of the last update are still owed ...
rows_active = rows._rows_stale and rows._row_ranges is not None
own = torch.zeros(n, dtype=torch.bool, device=on.parametersG.device)          # this rank's row blocks of the pair
for per_rank in (rows._row_ranges or []):
    own[per_rank[rank][0]:per_rank[rank][1]] = True
stale_before = bool((rows.parametersG != on.parametersG)[mask & ~own].any()) if rows_active else False
rows.flush()                                                                     # ... and after it every parameter holds the same bits
rows_ok = torch.equal(rows.parametersG, on.parametersG) and torch.equal(rows.parametersD, on.parametersD) and not rows._rows_stale
rows_ok = rows_ok and (stale_before or not rows_active or world == 1)
blocks = [[hi - lo for lo, hi in per_rank] for per_rank in (rows._row_ranges or [])]
for key in ("m", "v"):                                                           # Adam's moments: this rank's rows, and everything outside the pair
    a, c = rows.optimStateG[key], on.optimStateG[key]
    rows_ok = rows_ok and torch.equal(a[~mask], c[~mask]) and torch.equal(a[own], c[own])
assert (rows.optimStateG.get("row_shard") == (rank, world)) == rows_active
rows.gather_adam_state()                                                         # whole again on every rank
for key in ("m", "v"):
    rows_ok = rows_ok and torch.equal(rows.optimStateG[key], on.optimStateG[key])
rows_ok = rows_ok and "row_shard" not in rows.optimStateG


sig = torch.tensor([bits(on.parametersG), bits(on.optimStateG["m"]), bits(on.optimStateG["v"]), bits(on.parametersD)], dtype=torch.int64)
every = [torch.zeros_like(sig) for _ in range(world)]
dist.all_gather(every, sig)
same = all(torch.equal(e, every[0]) for e in every)
line = ("rank %d/%d  batch %d/rank  fused slices %s  gather buffer %d floats  |  on vs off: grad(exchanged part) %.2e  param %.3f lr  "
        "adam m %.2e  netD %.3f lr  |  replicas bit-identical: %s  |  row-sharded update (%s; row blocks %s) == gathered, bit for bit: %s"
        % (rank, world, b, ranges, on._opbuf.numel(), err_g, err_p, err_m, err_d, same, "active" if rows_active else "rows do not split: gathered",
           blocks, rows_ok))
print(line, flush=True)
ok = err_g < 1e-4 and err_p < 0.05 and err_m < 1e-3 and same and rows_ok
if not ok:
    sys.stderr.write("dp_rehearsal FAILED on " + line + "\n")
dist.barrier()
dist.destroy_process_group()
sys.exit(0 if ok else 1)
