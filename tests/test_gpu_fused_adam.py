"""optim.adam inside the bottleneck pair's weight-gradient kernel (vf_wgrad_adam_outer, vf_net_adam_fused,
optim.adam_update_fused): train.lua:104 (conv nef*8 -> nBottleneck on a 4x4 map) and :134 (full-conv nBottleneck -> ngf*8),
THNN accGradParameters with K = batch, followed by optim/adam.lua's element update.

What is held to what:
  * the gradient the kernel forms (stored on request) against numpy in double: 2e-6 of its max-norm (K < 64: fp32 products on the
    fp32 matrix pipe, fp32 accumulation in batch order; K >= 64 in the three-plane mode: operands pre-split into bf16 planes in
    fragment order by k_fused_planes_prep, six product terms on the bf16 pipe — the launch list says which);
  * x, m, v after the fused update against vf_adam_apply fed with that same gradient: BIT FOR BIT (one definition of the
    element update, vf_common.h vf_adam_upd);
  * not storing the gradient changes nothing else: bit for bit;
  * the trainers: fuse_adam "on" == "keep" bit for bit in every persistent tensor, eager and captured; "keep" == "off"
    (accGradParameters + the plain update) bit for bit where the plain path forms that gradient with the same kernel
    (K <= 32), and at 1e-6 of the gradient's max-norm where it uses the three-plane tiles (K = 64).
"""
import numpy as np
import pytest
import torch

from helpers import rel_err, to_np

pytestmark = pytest.mark.gpu


def _state(hipb, n, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, generator=g).to(hipb.device)
    m = (0.1 * torch.randn(n, generator=g)).to(hipb.device)
    v = (0.01 * torch.rand(n, generator=g)).to(hipb.device)
    t_dev = hipb.zeros(2, dtype=torch.int32)
    return x, m, v, t_dev


@pytest.mark.parametrize("K,Nu,Ncols", [(4, 64, 128), (3, 66, 256), (16, 200, 1024), (64, 192, 2048), (8, 4000, 8192), (33, 70, 384),
                                        (80, 70, 384), (128, 200, 1024), (72, 64, 128)])
def test_fused_kernel_is_accgrad_then_adam(K, Nu, Ncols, hipb):
    assert hipb.lib.vf_wgrad_adam_outer_supported(K, Nu, Ncols) == 1
    # which pipe forms the gradient: the operand-planes pre-pass runs for K >= 64, K % 16 == 0 (five k-groups at K = 80: an odd count;
    # a ragged last row tile at Nu = 70 / 200), never below and never for a K that is no multiple of 16
    x0, m0, v0, t0 = _state(hipb, Nu * Ncols, 7)
    U0, V0 = hipb.zeros(K, Nu), hipb.zeros(K, Ncols)
    hipb.adam_prep(2e-4, 0.5, 0.999, t0)
    hipb.prof_begin()
    hipb.wgrad_adam_outer(U0, V0, x0, m0, v0, None, 0.5, 0.999, 1e-8, t0)
    names = hipb.prof_end()
    assert ("adam_fused_operand_planes" in names) == (K >= 64 and K % 16 == 0), names.keys()
    gen = torch.Generator().manual_seed(K * 1000 + Nu)
    U = torch.randn(K, Nu, generator=gen).to(hipb.device)
    V = torch.randn(K, Ncols, generator=gen).to(hipb.device)
    n = Nu * Ncols
    lr, b1, b2, eps = 2e-4, 0.5, 0.999, 1e-8
    # fused, gradient stored
    x1, m1, v1, t1 = _state(hipb, n, 7)
    g1 = torch.full((n,), 123.0, device=hipb.device)
    for _ in range(2):                         # two updates: the second runs on carried m, v and t = 2
        hipb.adam_prep(lr, b1, b2, t1)
        hipb.wgrad_adam_outer(U, V, x1, m1, v1, g1, b1, b2, eps, t1)
    want = U.double().cpu().numpy().T @ V.double().cpu().numpy()
    assert rel_err(to_np(g1).reshape(Nu, Ncols), want) < 2e-6
    # the plain update fed with that gradient
    x2, m2, v2, t2 = _state(hipb, n, 7)
    for _ in range(2):
        hipb.adam_prep(lr, b1, b2, t2)
        hipb.adam_apply(x2, g1, m2, v2, b1, b2, eps, t2)
    assert torch.equal(x1, x2) and torch.equal(m1, m2) and torch.equal(v1, v2)
    assert int(t1[0].item()) == 2
    # gradient not stored: nothing else moves
    x3, m3, v3, t3 = _state(hipb, n, 7)
    for _ in range(2):
        hipb.adam_prep(lr, b1, b2, t3)
        hipb.wgrad_adam_outer(U, V, x3, m3, v3, None, b1, b2, eps, t3)
    assert torch.equal(x1, x3) and torch.equal(m1, m3) and torch.equal(v1, v3)
    assert float((x1 - _state(hipb, n, 7)[0]).abs().max()) > 0


@pytest.mark.parametrize("world,K,Nu,Ncols", [(2, 4, 64, 128), (4, 3, 66, 256), (8, 64, 128, 1024), (1, 5, 64, 128)])
def test_gathered_operands_give_the_mean_gradient_of_the_ranks(world, K, Nu, Ncols, hipb):
    """data parallel without exchanging the gradient: every rank's operands side by side (one segment per rank, as
    vf_comm_allgather_async leaves them) -> the mean over ranks of U_r^T V_r, formed and consumed in the fused kernel"""
    pad4 = lambda n: (n + 3) & ~3
    u_off, v_off = 0, pad4(K * Nu)
    seg = v_off + pad4(K * Ncols) + 8                      # (some slack: segments need not be dense)
    gen = torch.Generator().manual_seed(world * 100 + K)
    buf = torch.randn(world * seg, generator=gen).to(hipb.device)
    n = Nu * Ncols
    lr, b1, b2, eps = 2e-4, 0.5, 0.999, 1e-8
    x1, m1, v1, t1 = _state(hipb, n, 9)
    g1 = torch.zeros(n, device=hipb.device)
    hipb.adam_prep(lr, b1, b2, t1)
    hipb.wgrad_adam_outer_gathered(buf, u_off, v_off, world, K, seg, Nu, Ncols, x1, m1, v1, g1, b1, b2, eps, t1)
    h = buf.double().cpu().numpy()
    want = np.zeros((Nu, Ncols))
    for r in range(world):
        U = h[r * seg + u_off:r * seg + u_off + K * Nu].reshape(K, Nu)
        V = h[r * seg + v_off:r * seg + v_off + K * Ncols].reshape(K, Ncols)
        want += U.T @ V
    want /= world
    assert rel_err(to_np(g1).reshape(Nu, Ncols), want) < 2e-6
    x2, m2, v2, t2 = _state(hipb, n, 9)
    hipb.adam_prep(lr, b1, b2, t2)
    hipb.adam_apply(x2, g1, m2, v2, b1, b2, eps, t2)
    assert torch.equal(x1, x2) and torch.equal(m1, m2) and torch.equal(v1, v2)


@pytest.mark.parametrize("world,K,Nu,Ncols", [(2, 4, 256, 128), (4, 16, 512, 256), (8, 8, 4000, 1024),
                                              (3, 4, 200, 128), (3, 8, 4000, 256), (6, 4, 4000, 128), (7, 4, 1000, 128)])      # ragged row blocks
def test_update_sharded_by_weight_rows_equals_the_whole_tensor_update(world, K, Nu, Ncols, hipb):
    """data parallel without redoing the work N times: rank r forms the global-batch gradient of rows
    [r Nu / N, (r + 1) Nu / N) only and updates them; the row blocks of all ranks together are, BIT FOR BIT, the whole-tensor
    gathered update (x, m, v and the stored gradient) — so replicas that all-gather the updated rows stay identical."""
    pad4 = lambda n: (n + 3) & ~3
    u_off, v_off = 0, pad4(K * Nu)
    seg = v_off + pad4(K * Ncols)
    gen = torch.Generator().manual_seed(world * 10 + K)
    buf = torch.randn(world * seg, generator=gen).to(hipb.device)
    n = Nu * Ncols
    lr, b1, b2, eps = 2e-4, 0.5, 0.999, 1e-8
    x1, m1, v1, t1 = _state(hipb, n, 9)
    g1 = torch.zeros(n, device=hipb.device)
    hipb.adam_prep(lr, b1, b2, t1)
    hipb.wgrad_adam_outer_gathered(buf, u_off, v_off, world, K, seg, Nu, Ncols, x1, m1, v1, g1, b1, b2, eps, t1)
    x2, m2, v2, t2 = _state(hipb, n, 9)
    g2 = torch.zeros(n, device=hipb.device)
    hipb.adam_prep(lr, b1, b2, t2)
    # the library's row blocks (vf_net_fused_adam_row_range): even blocks of 2 * ceil(Nu / (2 world)) rows, a shorter last one where
    # the rows do not split (200 rows over 3 ranks: 68, 68, 64; 4000 over 3: 1334, 1334, 1332)
    bs = 2 * ((Nu + 2 * world - 1) // (2 * world))
    x0 = x2.clone()
    covered = 0
    for r in range(world):       # every virtual rank updates its row block of the shared tensors
        r0 = min(Nu, r * bs)
        rows = min(Nu, r0 + bs) - r0
        assert rows >= 64 and r0 == covered
        hipb.wgrad_adam_outer_rows(buf, u_off, v_off, world, K, seg, Nu, Ncols, r0, rows, x2, m2, v2, g2, b1, b2, eps, t2)
        covered = r0 + rows
        assert torch.equal(x2[covered * Ncols:], x0[covered * Ncols:]), "rank %d wrote past its rows" % r
    assert covered == Nu
    assert torch.equal(x1, x2) and torch.equal(m1, m2) and torch.equal(v1, v2) and torch.equal(g1, g2)
    with pytest.raises(Exception, match="rows"):
        hipb.wgrad_adam_outer_rows(buf, u_off, v_off, world, K, seg, Nu, Ncols, 1, bs, x2, m2, v2, None, b1, b2, eps, t2)


def test_adam_over_several_ranges_in_one_launch(hipb):
    """vf_adam_apply_ranges (the generator's flat vector around the two fused slices: three ranges, one launch) == vf_adam_apply range
    by range, bit for bit; elements outside the ranges are untouched; ranges must be whole float4s."""
    n = 1 << 16
    lr, b1, b2, eps = 2e-4, 0.5, 0.999, 1e-8
    ranges = [(0, 1024), (4096, 4096 + 20000), (40000, n)]
    g = torch.randn(n, generator=torch.Generator().manual_seed(4)).to(hipb.device)
    x1, m1, v1, t1 = _state(hipb, n, 9)
    x2, m2, v2, t2 = _state(hipb, n, 9)
    x0 = x1.clone()
    hipb.adam_prep(lr, b1, b2, t1)
    hipb.adam_apply_ranges(x1, g, m1, v1, ranges + [(512, 512)], b1, b2, eps, t1)      # (an empty range is skipped)
    hipb.adam_prep(lr, b1, b2, t2)
    for lo, hi in ranges:
        hipb.adam_apply(x2[lo:hi], g[lo:hi], m2[lo:hi], v2[lo:hi], b1, b2, eps, t2)
    assert torch.equal(x1, x2) and torch.equal(m1, m2) and torch.equal(v1, v2)
    assert torch.equal(x1[1024:4096], x0[1024:4096]) and float((x1[:1024] - x0[:1024]).abs().max()) > 0
    with pytest.raises(Exception, match="whole float4s"):
        hipb.adam_apply_ranges(x1, g, m1, v1, [(2, 1026)], b1, b2, eps, t1)


def test_unsupported_shapes_are_refused(hipb):
    for K, Nu, Ncols in ((4, 62, 128), (4, 65, 128), (4, 64, 192), (0, 64, 128)):
        assert hipb.lib.vf_wgrad_adam_outer_supported(K, Nu, Ncols) == 0
    U = torch.randn(4, 62).to(hipb.device)
    V = torch.randn(4, 128).to(hipb.device)
    x, m, v, t = _state(hipb, 62 * 128, 1)
    with pytest.raises(Exception, match="not this kernel's shape"):
        hipb.wgrad_adam_outer(U, V, x, m, v, None, 0.5, 0.999, 1e-8, t)


def _trainer(kind, opt, mode, batch, host="cabi"):
    from video_filler_amd.trainers import CenterTrainer, VidTrainer
    tr = (CenterTrainer if kind == "center" else VidTrainer)(opt, seed=3, host=host)
    tr.fuse_adam = mode
    tr.set_batch(*batch)
    return tr


def _persistent(tr):
    out = [tr.parametersG, tr.parametersD, tr.optimStateG["m"], tr.optimStateG["v"], tr.optimStateD["m"], tr.optimStateD["v"]]
    for net in (tr.netG, tr.netD):
        for mod in net.leaves():
            if hasattr(mod, "running_mean"):
                out += [mod.running_mean, mod.running_var]
    return out


@pytest.mark.parametrize("kind,B", [("center", 4), ("center", 64), ("vid", 4)])
def test_trainer_modes_walk_the_same_trajectory(kind, B, oracle, hipb, planes_gate):
    """fuse_adam on / keep / off from the same weights and batch, three iterations each (eager), and "on" captured."""
    if kind == "center":
        opt = dict(nBottleneck=128, wtl2=0.999, overlapPred=4, nef=16, ngf=16, ndf=16) if B == 64 else dict(nBottleneck=64, wtl2=0.999, overlapPred=4)
        batch = (torch.from_numpy(oracle.synth_center_batch(B, np.random.default_rng(5))),)
    else:
        opt = dict(nBottleneck=96, nc_in=27, nc_out=12, nef=32, ngf=32, ndf=32, weight_nomask=1, wtgdl=0.5)      # train_wholeim_input.lua's shape
        batch = tuple(torch.from_numpy(a) for a in oracle.synth_vid_batch(B, np.random.default_rng(6), 27, 12))
    on, keep, off, cap = [_trainer(kind, opt, m, batch) for m in ("on", "keep", "off", "on")]
    for t in (on, keep, off):
        for _ in range(3):
            t.step()
    cap.capture(warmup=2)
    cap.replay()
    torch.cuda.synchronize()
    assert len(on.fused_adam_ranges()) == 2 and not off.fused_adam_ranges() and not keep.fused_adam_ranges()
    assert keep.fuse_adam_slices() == on.fuse_adam_slices() == on.fused_adam_ranges() and not off.fuse_adam_slices()
    ranges = on.fused_adam_ranges()
    assert all(hi - lo == 16 * opt["nBottleneck"] * 8 * opt.get("nef", 64) for lo, hi in ranges)      # E6 and D1 (nef == ngf here)
    for a, b, c in zip(_persistent(on), _persistent(keep), _persistent(cap)):
        assert torch.equal(a, b) and torch.equal(a, c)
    # "keep" writes the whole gradient vector; "on" everything but the two slices (which keep what was there: zeros)
    gk, go = keep.gradParametersG, on.gradParametersG
    mask = torch.zeros_like(gk, dtype=torch.bool)
    for lo, hi in ranges:
        mask[lo:hi] = True
        assert float(gk[lo:hi].abs().max()) > 0 and float(go[lo:hi].abs().max()) == 0
    assert torch.equal(gk[~mask], go[~mask])
    # against accGradParameters + the plain update
    if B <= 32:
        for a, b in zip(_persistent(keep), _persistent(off)):
            assert torch.equal(a, b)
        assert torch.equal(keep.gradParametersG, off.gradParametersG)
    else:
        # K = 64: the plain path's three-plane tiles against the fused kernel's fp32 matrix-core products — one iteration from
        # the same weights (after that the two trajectories hold different weights)
        keep1, off1 = _trainer(kind, opt, "keep", batch), _trainer(kind, opt, "off", batch)
        keep1.step()
        off1.step()
        torch.cuda.synchronize()
        assert rel_err(to_np(keep1.gradParametersG), to_np(off1.gradParametersG)) < 1e-6
        lr = keep1.optimStateG["learningRate"]
        sel = off1.gradParametersG.abs() > 1e-3 * off1.gradParametersG.abs().max()
        assert float((keep1.parametersG - off1.parametersG).abs()[sel].max()) <= 0.02 * lr
    for k in ("errD", "errG", "errG_l2"):       # (the criteria sum in double with atomics: equal to the last few bits, not bit for bit)
        assert abs(on.losses()[k] - keep.losses()[k]) <= 1e-12 * abs(keep.losses()[k])
        assert abs(on.losses()[k] - cap.losses()[k]) <= 1e-12 * abs(keep.losses()[k])


def test_data_parallel_step_gathers_operands_instead_of_reducing_the_pair(oracle, hipb):
    """the phased data-parallel step on one rank: the bottleneck pair's slices are not exchanged (their operands are packed,
    all-gathered — the identity on one rank — and consumed by the fused kernel), everything else is all-reduced; the trajectory is
    the single-device one bit for bit.  The module-by-module host keeps accGradParameters + the plain update."""
    from helpers import attach_world1_comm
    opt = dict(nBottleneck=64, wtl2=0.999, overlapPred=4)
    batch = (torch.from_numpy(oracle.synth_center_batch(4, np.random.default_rng(5))),)
    mirror = _trainer("center", opt, "on", batch, host="mirror")
    mirror.step()
    assert mirror.fused_adam_ranges() == []
    attach_world1_comm(hipb)
    dp, dpk, plain = _trainer("center", opt, "on", batch), _trainer("center", opt, "keep", batch), _trainer("center", opt, "on", batch)
    dp.force_comm = dpk.force_comm = True
    for _ in range(3):
        dp.step_phased()
        dpk.step_phased()
        plain.step()
    torch.cuda.synchronize()
    ranges = dp.fused_adam_ranges()
    assert len(ranges) == 2 and ranges == plain.fused_adam_ranges() and dpk.fused_adam_ranges() == []
    n = dp.parametersG.numel()
    assert dp._exchange_ranges(0, n) == [(0, ranges[0][0]), (ranges[0][1], ranges[1][0]), (ranges[1][1], n)]
    assert dp._opbuf.numel() == 4 * (64 + 8192) * 2            # batch x (Nu + Ncols) floats per layer: what travels instead of 2 x 524288 gradients
    for a, b, c in zip(_persistent(dp), _persistent(plain), _persistent(dpk)):
        assert torch.equal(a, b) and torch.equal(a, c)
    for lo, hi in ranges:
        assert float(dp.gradParametersG[lo:hi].abs().max()) == 0 and float(dpk.gradParametersG[lo:hi].abs().max()) > 0
    # pipelined step and sharded Adam: gradients travel whole
    pp = _trainer("center", opt, "on", batch)
    pp.force_comm = True
    pp._pipelined = True
    pp.set_batch_d(False)
    pp.step_pipelined()
    pp.flush()
    assert pp.fused_adam_ranges() == [] and float(pp.gradParametersG[ranges[0][0]:ranges[0][1]].abs().max()) > 0


def test_second_backward_without_zeroing_is_refused(oracle, hipb):
    """the fused form needs a fresh gradient: accumulating onto a gradient that was never written must fail loudly"""
    from video_filler_amd.trainers import CenterTrainer
    opt = dict(nBottleneck=64, wtl2=0.999, overlapPred=4)
    tr = CenterTrainer(opt, seed=3, host="cabi")
    tr.set_batch(torch.from_numpy(oracle.synth_center_batch(4, np.random.default_rng(5))))
    tr.step()
    net = tr.netG
    x = tr._g_in()
    y = net.forward(x)
    gy = torch.ones_like(y)
    assert len(net.set_fused_adam(True)) == 2
    try:
        net.zeroGradParameters()
        net.backward(x, gy)
        with pytest.raises(Exception, match="one backward per zeroGradParameters"):
            net.backward(x, gy)
    finally:
        net.set_fused_adam(False)
