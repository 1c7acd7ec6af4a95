"""Two and three data-parallel ranks on the one GPU of the box, exchanging over gloo (scripts/dp_rehearsal.py): the N > 1 form of the
phased step with the bottleneck pair's operands gathered (vf_net_fused_adam_pack -> all-gather -> vf_net_adam_fused_gathered) against
the same step with that pair's gradients all-reduced; replicas must hold the same bits; and the same step with the fused update
sharded by weight rows (dp_fused = "rows": each rank updates its rows, the updated rows are all-gathered) — the same bits again.  (RCCL itself refuses two ranks on one
device; its entries are exercised on a one-rank communicator in tests/test_gpu_comm.py.)"""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world", [2, 3])
def test_gathered_operand_exchange_with_real_ranks(world):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "scripts", "dp_rehearsal.py")]
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    if r.returncode != 0:
        with open(os.path.join(ROOT, "gpurun_out", "dp_rehearsal_%d_failed.log" % world), "a") as fh:
            fh.write("==== stdout\n" + r.stdout + "\n==== stderr\n" + r.stderr + "\n")
    # (every rank exits non-zero on a failed check and torchrun passes that on; the ranks' lines may interleave on the shared stdout)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-6000:])
    assert r.stdout.count("replicas bit-identical: True") == world, r.stdout[-2000:]
    # the update sharded by weight rows: active at both world sizes — 200 rows are 100 / 100 at 2 ranks and a ragged 68 / 68 / 64 at 3
    assert r.stdout.count("== gathered, bit for bit: True") == world, r.stdout[-2000:]
    assert r.stdout.count("row-sharded update (active") == world, r.stdout[-2000:]
    # (blocks in elements of the flat vector: 4096 columns per row at ngf = 32)
    assert ("[409600, 409600]" in r.stdout) == (world == 2) and ("[278528, 278528, 262144]" in r.stdout) == (world == 3), r.stdout[-2000:]
    lines = [r.stdout]
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "dp_rehearsal_%d.log" % world), "w") as fh:
        fh.write("\n".join(lines) + "\n")
