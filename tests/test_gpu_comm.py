"""vf_comm_* — the data-parallel exchange of the C-ABI (include/vf_hip.h, csrc/vf_comm.hip) — on the one GPU a test box has:
a communicator of ONE rank (RCCL refuses two ranks on one device).  What can be checked here: the library binds RCCL at run
time, every entry does what it says when averaging over one rank is the identity, the stream ordering of the asynchronous
form (a producer kernel before, a consumer kernel after), and that the inline form is recorded by a graph capture together
with the kernels around it (SyncBN's sums).  The arithmetic of N > 1 ranks is covered on the CPU over gloo
(tests/test_host_logic.py: world 2 and 4, synchronised and local BatchNorm)."""
import ctypes as C

import numpy as np
import pytest
import torch

from helpers import attach_world1_comm, rel_err

pytestmark = pytest.mark.gpu


def test_one_rank_communicator_entries(hipb):
    B = attach_world1_comm(hipb)
    lib, comm, ctx = B.lib, B.comm, B.ctx
    assert lib.vf_comm_world(comm) == 1 and lib.vf_comm_rank(comm) == 0
    n = 1 << 22
    x = torch.randn(n, device=B.device)
    want = x.clone()
    t = C.c_int32(-1)
    # asynchronous average, ordered after a producer on the context's stream and before a consumer
    B.scale_shift(x, 2.0, 1.0)
    assert lib.vf_comm_allreduce_avg_async(comm, ctx, C.c_void_p(x.data_ptr()), n, C.byref(t)) == 0
    assert 0 <= t.value < 64
    assert lib.vf_comm_wait(comm, ctx, t.value) == 0
    B.scale_shift(x, 0.5, -0.5)
    torch.cuda.synchronize()
    assert torch.equal(x, (want * 2.0 + 1.0) * 0.5 - 0.5)
    # f64 sums inline (SyncBN), max / min, broadcast, barrier
    s = torch.arange(64, dtype=torch.float64, device=B.device)
    for op in ("sum", "max", "min"):
        B.all_reduce(s, op=op)
    B.comm_broadcast(s, 0)
    B.comm_barrier()
    assert torch.equal(s.cpu(), torch.arange(64, dtype=torch.float64))
    # many collectives in flight: tickets wrap round a ring of 64
    seen = set()
    for _ in range(70):
        assert lib.vf_comm_allreduce_async(comm, ctx, C.c_void_p(x.data_ptr()), 1024, 0, 0, C.byref(t)) == 0
        seen.add(t.value)
    assert lib.vf_comm_wait(comm, ctx, t.value) == 0
    torch.cuda.synchronize()
    assert len(seen) == 64
    # the two halves of an all-reduce (sharded optimiser): one rank -> its shard is the whole vector, both are the identity
    v = torch.randn(1 << 16, device=B.device)
    v0 = v.clone()
    sh = B.reduce_scatter_avg(v, 1, 0)
    B.all_gather_shards(v, 1, 0)
    torch.cuda.synchronize()
    assert sh.data_ptr() == v.data_ptr() and torch.equal(v, v0)
    # argument errors come back as codes with a message, not as crashes
    assert lib.vf_comm_allreduce_async(comm, ctx, C.c_void_p(x.data_ptr()), 16, 7, 0, C.byref(t)) != 0
    assert b"dtype" in lib.vf_last_error()
    assert lib.vf_comm_wait(comm, ctx, 64) != 0
    assert lib.vf_comm_broadcast(comm, ctx, C.c_void_p(x.data_ptr()), 16, 0, 3) != 0


def test_gradient_bucket_overlaps_the_kernels_issued_after_it(hipb):
    """vf_comm_allreduce_avg_async returns with the collective on the communicator's stream: kernels given to the context's
    stream afterwards run without waiting for it, and vf_comm_wait is the only join."""
    B = attach_world1_comm(hipb)
    big = torch.ones(64 << 20, device=B.device)                 # 256 MB bucket: netG's tail bucket
    other = torch.zeros(1 << 20, device=B.device)
    h = B.all_reduce_avg(big, 1, async_op=True)
    B.scale_shift(other, 1.0, 3.0)                              # independent work behind the launch
    h.wait()
    B.scale_shift(big, 2.0, 0.0)                                # consumer: after the collective
    torch.cuda.synchronize()
    assert float(other[0]) == 3.0 and float(big[0]) == 2.0 and float(big[-1]) == 2.0


@pytest.mark.parametrize("kind", ["center", "vid"])
def test_syncbn_collectives_are_captured_with_the_phases(kind, oracle, hipb):
    """SyncBN puts one collective per BatchNorm layer and pass INSIDE the phases (a layer's statistics are needed before the
    layer can be applied: the dependency is per layer, only the two moments travel together).  They are issued on the
    context's stream (vf_comm_allreduce_inline), so capture_phased records them in the same four graphs.  One rank: the
    sums come back unchanged, so the replayed trajectory must agree with a trainer that never leaves the device-local path
    (different kernels for the statistics: fp32 agreement, not bitwise), and replaying must equal running eagerly (bitwise)."""
    from video_filler_amd import nn
    from video_filler_amd.trainers import CenterTrainer, VidTrainer
    attach_world1_comm(hipb)
    if kind == "center":
        opt = dict(nBottleneck=128, wtl2=0.999, overlapPred=4, smooth=True)
        batch = (torch.from_numpy(oracle.synth_center_batch(4, np.random.default_rng(5))),)
        mk = lambda **kw: CenterTrainer(opt, seed=3, **kw)
    else:
        opt = dict(nBottleneck=128, predLen=2, smooth=True)
        batch = tuple(torch.from_numpy(a) for a in oracle.synth_vid_batch(4, np.random.default_rng(5), 6))
        mk = lambda **kw: VidTrainer(opt, seed=3, **kw)
    # smooth nets (LeakyReLU(1.0) everywhere: same graph, same kernels): the first-iteration comparison below is between two
    # BatchNorm statistics paths that round differently, and on the real nets one pre-activation within rounding distance of
    # its kink moves a batch-of-4 gradient by ~1e-3 of its norm — this test is about the collectives

    def synced():
        t = mk(overlap=False)                  # what the trainers do for world > 1 with sync_bn (one stream)
        t.fuse_adam = "keep"                   # (its gradient vector is read below)
        t.set_batch_d(False)                   # (SyncBN all-reduces one group's sums per layer and pass: no 2B netD batching)
        t.set_batch(*batch)
        t.force_comm = True
        n = 0
        for net in (t.netG, t.netD):
            for m in net.leaves():
                if isinstance(m, nn.SpatialBatchNormalization):
                    m.sync_force = True
                    n += 1
        assert n >= 8
        return t

    plain = mk()
    plain.fuse_adam = "keep"       # (its gradient vector is read below)
    plain.set_batch_d(False)
    plain.set_batch(*batch)
    eager, graph = synced(), synced()
    plain.step()
    eager.step_phased()
    torch.cuda.synchronize()
    # first iteration: same weights, same batch — the two BatchNorm paths agree to fp32 rounding
    assert rel_err(eager.gradParametersG.cpu().numpy(), plain.gradParametersG.cpu().numpy()) < 1e-4
    l0, l1 = plain.losses(), eager.losses()
    for k in ("errD", "errG", "errG_l2"):
        assert abs(l0[k] - l1[k]) <= 2e-5 * max(1.0, abs(l0[k])), (k, l0[k], l1[k])
    for _ in range(4):
        plain.step()
        eager.step_phased()
    graph.capture_phased(warmup=3)
    graph.step_phased()
    graph.step_phased()
    torch.cuda.synchronize()
    assert torch.equal(graph.parametersG, eager.parametersG) and torch.equal(graph.parametersD, eager.parametersD)
    # seven Adam steps later the trajectories are still together (Adam amplifies rounding on near-zero gradients: DESIGN.md)
    assert rel_err(graph.parametersG.cpu().numpy(), plain.parametersG.cpu().numpy()) < 5e-2


def test_torch_distributed_fallback_collectives_stay_on_the_device(hipb):
    """bench.py --comm torch / the --comm auto fallback: no C-ABI communicator, an NCCL (= RCCL) process group.  Every collective
    the data-parallel step uses must run on the DEVICE tensor (reduce_scatter_avg / all_gather_shards went through a
    host copy, which an NCCL-only group refuses) — a world of one rank, where every collective is the identity."""
    import os
    import torch.distributed as dist
    B = hipb
    saved, B.comm = B.comm, None
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29581")
    own = not dist.is_initialized()
    if own:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=B.device)
    try:
        assert dist.get_backend() == "nccl"
        v = torch.randn(1 << 16, device=B.device)
        v0 = v.clone()
        sh = B.reduce_scatter_avg(v, 1, 0)
        B.all_gather_shards(v, 1, 0)
        h = B.all_gather_shards(v, 1, 0, async_op=True)
        h.wait()
        B.all_reduce_avg(v, 1)
        s = torch.arange(8, dtype=torch.float64, device=B.device)
        B.all_reduce(s, op="sum")
        torch.cuda.synchronize()
        assert sh.data_ptr() == v.data_ptr() and torch.equal(v, v0) and torch.equal(s.cpu(), torch.arange(8, dtype=torch.float64))
    finally:
        B.comm = saved
        if own:
            dist.destroy_process_group()



def test_row_exchange_waits_in_front_of_the_bottleneck_conv(hipb):
    """the rows another rank updated arrive by a collective on the exchange stream, and the generator's next forward
    waits for it in front of its bottleneck conv only (vf_net_forward_wait_fused; train.lua:89-104: E1 ... E5 read none of those rows).
    One rank: an all-gather of one block and a broadcast (the ragged form) of the two fused slices are the identity — the forward with
    the tickets armed must equal the plain one, consume the tickets, and be capturable in one graph with the collectives it waits for."""
    from video_filler_amd.cnet import adopt_if_chain
    from video_filler_amd.trainers import build_netG, weights_init
    B = attach_world1_comm(hipb)
    net = build_netG(3, 3, 16, 16, 96, False)
    weights_init(net, torch.Generator().manual_seed(5))
    net = adopt_if_chain(net)
    assert type(net).__name__ == "CNet"
    flat, _ = net.getParameters()
    net.evaluate()       # (training-mode forwards shift their BatchNorm sums by the running mean, which moves: not bit-repeatable)
    x = torch.rand((4, 3, 128, 128), generator=torch.Generator().manual_seed(6)).to(B.device).contiguous(memory_format=torch.channels_last) * 2 - 1
    want = net.forward(x).clone()
    slices = net.set_fused_adam(True)
    assert len(slices) == 2 and net.fused_adam_rows_ok(1)
    ranges = net.fused_adam_row_ranges(1)
    assert [r[0] for r in ranges] == slices                      # one rank owns every row
    net.set_fused_adam(False)

    def exchange_and_forward():
        hs = []
        hs += B.all_gather_ranges(flat, [ranges[0][0]], 0, async_op=True)                       # equal blocks: an all-gather
        t = C.c_int32(-1)
        lo, hi = ranges[1][0]                                                                   # ragged blocks: one broadcast per rank
        assert B.lib.vf_comm_broadcast_async(B.comm, B.ctx, C.c_void_p(flat[lo:hi].data_ptr()), hi - lo, 0, C.byref(t)) == 0
        tickets = [h.ticket for h in hs] + [t.value]
        net.forward_wait_fused(B.comm, tickets)
        return net.forward(x)
    got = exchange_and_forward().clone()
    torch.cuda.synchronize()
    assert torch.equal(got, want)
    assert torch.equal(net.forward(x), want)                     # (the tickets were one-shot)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        B.use_current_stream()
        out = exchange_and_forward()
    B.use_current_stream()
    out.zero_()
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, want)
    assert B.lib.vf_net_forward_wait_fused(net._net, None, 0) != 0 and b"bad argument" in B.lib.vf_last_error()
