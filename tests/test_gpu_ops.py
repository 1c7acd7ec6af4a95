"""GPU parity: every C-ABI op of the HIP path against the CPU oracle on the same seeded inputs.

fp32 tolerance (stated): 2e-5 of the tensor's max-norm for single ops whose contraction is <= 8192 long
(the f32 MFMA is an fmaf chain; the oracle sums in a different order), 1e-6 for pointwise ops.
"""
import numpy as np
import pytest
import torch

from helpers import assert_close, rel_err, to_dev, to_np

pytestmark = pytest.mark.gpu

TOL = 2e-5

# (B, Cin, H, Cout, stride, pad)
CONV_CASES = [
    (2, 3, 16, 64, 2, 1),      # first layer of train.lua nets: Cin = 3 (scalar gather path)
    (2, 12, 16, 32, 2, 1),     # first netD layer of the video nets, predLen = 4
    (3, 16, 8, 32, 2, 1),      # smallest vectorised path
    (2, 64, 16, 64, 2, 1),
    (2, 32, 8, 128, 2, 1),
    (5, 64, 4, 128, 2, 1),     # ragged M (B = 5), 4x4 -> 2x2
    (2, 128, 4, 100, 1, 0),    # bottleneck conv 4x4 -> 1x1, nBottleneck = 100 (train.lua default)
    (3, 64, 4, 260, 1, 0),     # bottleneck with N not a multiple of the tile
    (4, 512, 4, 1, 1, 0),      # netD's last conv + sigmoid
    (1, 48, 8, 64, 2, 1),      # B = 1, 16-frame clips (nc = 48)
]


def _rand(rng, *shape):
    return rng.standard_normal(shape).astype(np.float32)


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_fwd_bwd(case, oracle, hipb):
    B, Cin, H, Cout, s, p = case
    rng = np.random.default_rng(abs(hash(case)) % (2 ** 31))
    ref = oracle.SpatialConvolution(Cin, Cout, 4, 4, s, s, p, p)
    ref.weight[...] = _rand(rng, *ref.weight.shape) * 0.05
    ref.bias[...] = _rand(rng, Cout)
    x = _rand(rng, B, Cin, H, H)
    y = ref.forward(x)
    gy = _rand(rng, *y.shape)
    ref.gradWeight[...] = _rand(rng, *ref.weight.shape)     # accumulate onto a non-zero buffer
    ref.gradBias[...] = _rand(rng, Cout)
    gw0, gb0 = ref.gradWeight.copy(), ref.gradBias.copy()
    ref.backward(x, gy)

    dx, dw, db = to_dev(x, hipb), to_dev(ref.weight, hipb), to_dev(ref.bias, hipb)
    dy = hipb.empty_act(*y.shape)
    hipb.conv2d_fwd(dx, dw, db, dy, 4, s, p)
    assert_close(to_np(dy), y, TOL, "conv fwd %s" % (case,))
    dgy = to_dev(gy, hipb)
    dgx = hipb.empty_act(*x.shape)
    hipb.conv2d_bwd_data(dgy, dw, dgx, 4, s, p)
    assert_close(to_np(dgx), ref.gradInput, TOL, "conv bwd_data %s" % (case,))
    dgw, dgb = to_dev(gw0, hipb), to_dev(gb0, hipb)
    hipb.conv2d_bwd_weight(dx, dgy, dgw, dgb, 4, s, p, 1.0)
    assert_close(to_np(dgw), ref.gradWeight, TOL, "conv bwd_weight(beta=1) %s" % (case,))
    assert_close(to_np(dgb), ref.gradBias, TOL, "conv bias grad %s" % (case,))
    hipb.conv2d_bwd_weight(dx, dgy, dgw, dgb, 4, s, p, 0.0)
    assert_close(to_np(dgw), ref.gradWeight - gw0, TOL * 2, "conv bwd_weight(beta=0) %s" % (case,))


def test_conv_fused_activation(oracle, hipb):
    rng = np.random.default_rng(7)
    ref = oracle.SpatialConvolution(16, 32, 4, 4, 2, 2, 1, 1)
    ref.weight[...] = _rand(rng, *ref.weight.shape) * 0.1
    x = _rand(rng, 2, 16, 8, 8)
    y = ref.forward(x).copy()
    dx, dw, db = to_dev(x, hipb), to_dev(ref.weight, hipb), to_dev(ref.bias, hipb)
    for act, fn in (("lrelu", lambda v: np.where(v > 0, v, 0.2 * v)), ("relu", lambda v: np.maximum(v, 0)),
                    ("tanh", np.tanh), ("sigmoid", lambda v: 1 / (1 + np.exp(-v)))):
        dy = hipb.empty_act(*y.shape)
        hipb.conv2d_fwd(dx, dw, db, dy, 4, 2, 1, act, 0.2)
        assert_close(to_np(dy), fn(y), TOL, "fused " + act)


# (B, Cin, H, Cout, stride, pad)
DECONV_CASES = [
    (2, 100, 1, 128, 1, 0),    # decoder's first full-conv: 1x1 -> 4x4 (plain GEMM)
    (3, 260, 1, 64, 1, 0),
    (2, 128, 4, 64, 2, 1),
    (2, 64, 8, 3, 2, 1),       # last layer of train.lua's netG: Cout = 3
    (2, 64, 8, 12, 2, 1),      # last layer of the video netG, predLen = 4
    (5, 16, 2, 16, 2, 1),      # ragged M
    (1, 32, 16, 48, 2, 1),
]


@pytest.mark.parametrize("case", DECONV_CASES)
def test_deconv_fwd_bwd(case, oracle, hipb):
    B, Cin, H, Cout, s, p = case
    rng = np.random.default_rng(abs(hash(case)) % (2 ** 31))
    ref = oracle.SpatialFullConvolution(Cin, Cout, 4, 4, s, s, p, p)
    ref.weight[...] = _rand(rng, *ref.weight.shape) * 0.05
    ref.bias[...] = _rand(rng, Cout)
    x = _rand(rng, B, Cin, H, H)
    y = ref.forward(x)
    gy = _rand(rng, *y.shape)
    ref.gradWeight[...] = _rand(rng, *ref.weight.shape)
    ref.gradBias[...] = _rand(rng, Cout)
    gw0, gb0 = ref.gradWeight.copy(), ref.gradBias.copy()
    ref.backward(x, gy)

    dx, dw, db = to_dev(x, hipb), to_dev(ref.weight, hipb), to_dev(ref.bias, hipb)
    dy = hipb.empty_act(*y.shape)
    hipb.deconv2d_fwd(dx, dw, db, dy, 4, s, p)
    assert_close(to_np(dy), y, TOL, "deconv fwd %s" % (case,))
    hipb.deconv2d_fwd(dx, dw, db, dy, 4, s, p, "relu")
    assert_close(to_np(dy), np.maximum(y, 0), TOL, "deconv fwd+relu %s" % (case,))
    dgy = to_dev(gy, hipb)
    dgx = hipb.empty_act(*x.shape)
    hipb.deconv2d_bwd_data(dgy, dw, dgx, 4, s, p)
    assert_close(to_np(dgx), ref.gradInput, TOL, "deconv bwd_data %s" % (case,))
    dgw, dgb = to_dev(gw0, hipb), to_dev(gb0, hipb)
    hipb.deconv2d_bwd_weight(dx, dgy, dgw, dgb, 4, s, p, 1.0)
    assert_close(to_np(dgw), ref.gradWeight, TOL, "deconv bwd_weight %s" % (case,))
    assert_close(to_np(dgb), ref.gradBias, TOL, "deconv bias grad %s" % (case,))


@pytest.mark.parametrize("shape", [(4, 64, 8, 8), (2, 128, 16, 16), (8, 100, 1, 1), (3, 260, 4, 4), (16, 4000, 1, 1)])
@pytest.mark.parametrize("act", ["none", "lrelu", "relu"])
def test_batchnorm(shape, act, oracle, hipb):
    B, C, H, W = shape
    rng = np.random.default_rng(C * 7 + H)
    ref = oracle.SpatialBatchNormalization(C)
    ref.weight[...] = 1 + 0.1 * _rand(rng, C)
    ref.bias[...] = 0.1 * _rand(rng, C)
    ref.running_mean[...] = 0.3 * _rand(rng, C)
    ref.running_var[...] = 1 + 0.2 * np.abs(_rand(rng, C))
    rm0, rv0 = ref.running_mean.copy(), ref.running_var.copy()
    x = (_rand(rng, *shape) * 1.7 + 0.8).astype(np.float32)
    y = ref.forward(x).copy()
    slope = 0.2
    if act == "lrelu":
        ya = np.where(y > 0, y, slope * y).astype(np.float32)
    elif act == "relu":
        ya = np.maximum(y, 0)
    else:
        ya = y
    gy = _rand(rng, *shape)
    g_eff = gy.copy()
    if act == "lrelu":
        g_eff = np.where(ya > 0, gy, slope * gy).astype(np.float32)
    elif act == "relu":
        g_eff = np.where(ya > 0, gy, 0).astype(np.float32)
    ref.gradWeight[...] = _rand(rng, C)
    ref.gradBias[...] = _rand(rng, C)
    gw0, gb0 = ref.gradWeight.copy(), ref.gradBias.copy()
    ref.backward(x, g_eff)

    dx = to_dev(x, hipb)
    dy = hipb.empty_act(*shape)
    gamma, beta = to_dev(ref.weight, hipb), to_dev(ref.bias, hipb)
    rm, rv = to_dev(rm0, hipb), to_dev(rv0, hipb)
    sm, si = hipb.zeros(C), hipb.zeros(C)
    sums = hipb.zeros(2 * C, dtype=torch.float64)
    hipb.bn_stats(dx, rm, sums)
    hipb.bn_finalize(sums, rm, rv, sm, si, B * H * W, 0.1, 1e-5)
    hipb.bn_apply(dx, dy, gamma, beta, sm, si, act, slope)
    assert_close(to_np(dy), ya, 1e-5, "bn fwd")
    assert_close(to_np(sm), ref.save_mean, 1e-5, "save_mean")
    assert_close(to_np(si), ref.save_std, 1e-5, "save_invstd")
    assert_close(to_np(rm), ref.running_mean, 1e-5, "running_mean")
    if B * H * W > 1:
        assert_close(to_np(rv), ref.running_var, 1e-5, "running_var")
    dgy = to_dev(gy, hipb)
    dgx = hipb.empty_act(*shape)
    ggam, gbet = to_dev(gw0, hipb), to_dev(gb0, hipb)
    hipb.bn_bwd_stats(dx, dy if act != "none" else None, dgy, sm, sums, act, slope)
    hipb.bn_bwd_apply(dx, dy if act != "none" else None, dgy, dgx, ggam, gbet, gamma, sm, si, sums, B * H * W, act, slope, 1.0)
    assert_close(to_np(dgx), ref.gradInput, 5e-5, "bn gradInput")
    assert_close(to_np(ggam), ref.gradWeight, 2e-5, "bn gradWeight")
    assert_close(to_np(gbet), ref.gradBias, 2e-5, "bn gradBias")
    # evaluate mode
    ref.train = False
    ye = ref.forward(x)
    hipb.bn_eval_fwd(dx, dy, gamma, beta, rm, rv, 1e-5)
    assert_close(to_np(dy), ye, 1e-5, "bn eval")


def test_pointwise_and_layout(oracle, hipb):
    rng = np.random.default_rng(3)
    x = _rand(rng, 3, 5, 8, 8)
    d = torch.from_numpy(x).to(hipb.device)             # NCHW contiguous on device
    out = hipb.empty(3, 8, 8, 5)
    hipb._c("vf_nchw_to_nhwc", d.data_ptr(), out.data_ptr(), 3, 5, 8, 8)
    np.testing.assert_array_equal(to_np(out), x.transpose(0, 2, 3, 1))
    back = hipb.empty(3, 5, 8, 8)
    hipb._c("vf_nhwc_to_nchw", out.data_ptr(), back.data_ptr(), 3, 5, 8, 8)
    np.testing.assert_array_equal(to_np(back), x)
    a, b = to_dev(x, hipb), to_dev(_rand(rng, 3, 5, 8, 8), hipb)
    an, bn = to_np(a), to_np(b)
    y = b.clone()
    hipb.axpby(0.25, a, -1.5, y)
    assert_close(to_np(y), 0.25 * an - 1.5 * bn, 1e-6, "axpby")
    y = b.clone()
    hipb.cmul(a, y)
    assert_close(to_np(y), an * bn, 1e-6, "cmul")
    y = b.clone()
    hipb.scale_shift(y, 0.95, 0.05)
    assert_close(to_np(y), bn * np.float32(0.95) + np.float32(0.05), 1e-6, "scale_shift")
    mask = (rng.random((3, 5, 8, 8)) > 0.5).astype(np.float32)
    o = hipb.empty_act(3, 5, 8, 8)
    hipb.masked_compose(o, a, b, to_dev(mask, hipb))
    np.testing.assert_array_equal(to_np(o), np.where(mask != 0, bn, an))
    for act in ("lrelu", "relu", "tanh", "sigmoid"):
        yv = torch.empty_like(a)
        hipb.act_fwd(a, yv, act, 0.2)
        g = torch.empty_like(a)
        hipb.act_bwd(yv, b, g, act, 0.2)
        ynp = to_np(yv)
        if act == "lrelu":
            ey, eg = np.where(an > 0, an, 0.2 * an), np.where(ynp > 0, bn, 0.2 * bn)
        elif act == "relu":
            ey, eg = np.maximum(an, 0), np.where(ynp > 0, bn, 0)
        elif act == "tanh":
            ey, eg = np.tanh(an), bn * (1 - ynp * ynp)
        else:
            ey = 1 / (1 + np.exp(-an))
            eg = bn * (1 - ynp) * ynp
        assert_close(ynp, ey, 2e-6, act + " fwd")
        assert_close(to_np(g), eg, 2e-6, act + " bwd")
    # zero_segments
    base = hipb.zeros(100) + 1
    offs = torch.tensor([3, 50], dtype=torch.int64, device=hipb.device)
    lens = torch.tensor([5, 10], dtype=torch.int64, device=hipb.device)
    hipb.zero_segments(base, offs, lens)
    e = np.ones(100, np.float32)
    e[3:8] = 0
    e[50:60] = 0
    np.testing.assert_array_equal(to_np(base), e)


def test_criteria(oracle, hipb):
    rng = np.random.default_rng(5)
    loss = hipb.zeros(1, dtype=torch.float64)
    # BCE incl. saturated inputs (eps = 1e-12 semantics differ from torch's clamp)
    p = np.concatenate([rng.random(30), [0.0, 1.0, 1e-9, 1 - 1e-7]]).astype(np.float32)
    for label in (0.0, 1.0):
        t = np.full(p.shape, label, np.float32)
        hipb.bce_fwd(to_dev(p, hipb), label, loss)
        assert abs(loss.item() - oracle.lib().vfo_bce_fwd(oracle._p(p), oracle._p(t), p.size)) <= 1e-9 * max(1, abs(loss.item()))
        g = hipb.zeros(p.size)
        hipb.bce_bwd(to_dev(p, hipb), label, g)
        assert_close(to_np(g), oracle.BCECriterion().backward(p, t), 1e-6, "bce bwd")
    x, t = _rand(rng, 2, 6, 16, 16), _rand(rng, 2, 6, 16, 16)
    dx, dt = to_dev(x, hipb), to_dev(t, hipb)
    hipb.mse_fwd(dx, dt, loss)
    assert abs(loss.item() - oracle.MSECriterion().forward(x, t)) < 1e-6
    g = torch.empty_like(dx)
    hipb.mse_bwd(dx, dt, g)
    assert_close(to_np(g), oracle.MSECriterion().backward(x, t), 1e-6, "mse bwd")
    hipb.gdl_fwd(dx, dt, loss)
    assert abs(loss.item() - oracle.GDLCriterion(1).forward(x, t)) < 1e-6
    m = (rng.random(x.shape) > 0.6).astype(np.uint8)
    crit = oracle.MaskedMSECriterion(0.05)
    crit.setMask(m)
    dm = to_dev(m, hipb)
    hipb.masked_mse_fwd(dx, dt, dm, 0.05, loss)
    assert abs(loss.item() - crit.forward(x, t)) < 1e-6
    hipb.masked_mse_bwd(dx, dt, dm, 0.05, g)
    assert_close(to_np(g), crit.backward(x, t), 1e-6, "masked mse bwd")
    # fused recon gradient: mask form and band form against the reference op sequence
    dfdg0 = _rand(rng, *x.shape)
    wt, lam, wtgdl = 0.999, 0.05, 0.5
    gl2 = oracle.MSECriterion().backward(x, t)
    w = m.astype(np.float32) * np.float32(1 - lam) + np.float32(lam)
    exp = dfdg0 * np.float32(1 - wt) + np.float32(wt) * (gl2 * w) + np.float32(wtgdl) * gl2
    d = to_dev(dfdg0, hipb)
    hipb.recon_grad_mix(d, dx, dt, to_dev(m.astype(np.float32), hipb), 1 - wt, wt * lam + wtgdl, wt * (1 - lam), 0, loss)
    assert_close(to_np(d), exp, 2e-6, "recon_grad_mix(mask)")
    assert abs(loss.item() - oracle.MSECriterion().forward(x, t)) < 1e-6
    ov = 3
    W = np.full(x.shape, np.float32(10 * wt), np.float32)
    W[:, :, ov:16 - ov, ov:16 - ov] = np.float32(wt)
    exp = dfdg0 * np.float32(1 - wt) + W * gl2
    d = to_dev(dfdg0, hipb)
    hipb.recon_grad_mix(d, dx, dt, None, 1 - wt, wt, 9 * wt, ov, loss)
    assert_close(to_np(d), exp, 2e-6, "recon_grad_mix(band)")


def test_adam(oracle, hipb):
    rng = np.random.default_rng(11)
    n = 10007
    x, g = _rand(rng, n), _rand(rng, n) * 1e-3
    g[:50] = 0.0
    g[50:100] *= 1e-6
    xr = x.copy()
    state = {"learningRate": 0.002, "beta1": 0.5}
    npad = (n + 3) // 4 * 4
    dx, dg = hipb.zeros(npad), hipb.zeros(npad)
    dx[:n] = torch.from_numpy(x).to(hipb.device)
    dg[:n] = torch.from_numpy(g).to(hipb.device)
    m, v = hipb.zeros(npad), hipb.zeros(npad)
    t_dev = hipb.zeros(2, dtype=torch.int32)
    for it in range(3):
        oracle.adam(lambda _x: (0.0, g), xr, state)
        hipb.adam_step(dx[:n], dg[:n], m[:n], v[:n], 0.002, 0.5, 0.999, 1e-8, t_dev)
        np.testing.assert_allclose(to_np(dx[:n]), xr, rtol=0, atol=2e-7 * (it + 1))
    assert int(t_dev[0].item()) == 3
    np.testing.assert_allclose(to_np(m[:n]), state["m"], rtol=1e-6, atol=1e-12)
    np.testing.assert_allclose(to_np(v[:n]), state["v"], rtol=1e-6, atol=1e-15)


@pytest.mark.parametrize("shape", [(4, 64, 8, 8), (16, 4000, 1, 1), (8, 512, 4, 4), (2, 64, 128, 64), (4, 128, 64, 64)])
@pytest.mark.parametrize("act", ["none", "lrelu"])
def test_batchnorm_fused_entry_points(shape, act, oracle, hipb):
    """vf_bn_train_fwd / vf_bn_bwd: the single-launch channel-sliced kernels (npix <= 8192) and the three-launch
    path (larger tensors) behind the same entry points."""
    B, C, H, W = shape
    rng = np.random.default_rng(C + H)
    ref = oracle.SpatialBatchNormalization(C)
    ref.weight[...] = 1 + 0.1 * _rand(rng, C)
    ref.bias[...] = 0.1 * _rand(rng, C)
    ref.running_mean[...] = 0.2 * _rand(rng, C)
    rm0, rv0 = ref.running_mean.copy(), ref.running_var.copy()
    x = (_rand(rng, *shape) * 1.3 + 0.4).astype(np.float32)
    y = ref.forward(x).copy()
    slope = 0.2
    ya = np.where(y > 0, y, slope * y).astype(np.float32) if act == "lrelu" else y
    gy = _rand(rng, *shape)
    g_eff = np.where(ya > 0, gy, slope * gy).astype(np.float32) if act == "lrelu" else gy
    ref.backward(x, g_eff)
    dx, dy = to_dev(x, hipb), hipb.empty_act(*shape)
    gamma, beta = to_dev(ref.weight, hipb), to_dev(ref.bias, hipb)
    rm, rv, sm, si = to_dev(rm0, hipb), to_dev(rv0, hipb), hipb.zeros(C), hipb.zeros(C)
    sums = hipb.zeros(2 * C, dtype=torch.float64)
    hipb.bn_train_fwd(dx, dy, gamma, beta, rm, rv, sm, si, sums, 0.1, 1e-5, act, slope)
    assert_close(to_np(dy), ya, 1e-5, "bn fused fwd")
    assert_close(to_np(rm), ref.running_mean, 1e-5, "running_mean")
    assert_close(to_np(rv), ref.running_var, 1e-5, "running_var")
    assert_close(to_np(si), ref.save_std, 1e-5, "save_invstd")
    dgx, gg, gb = hipb.empty_act(*shape), hipb.zeros(C), hipb.zeros(C)
    hipb.bn_bwd(dx, dy if act != "none" else None, to_dev(gy, hipb), dgx, gg, gb, gamma, sm, si, sums, act, slope, 0.0)
    assert_close(to_np(dgx), ref.gradInput, 5e-5, "bn fused gradInput")
    assert_close(to_np(gg), ref.gradWeight, 2e-5, "bn fused gradWeight")
    assert_close(to_np(gb), ref.gradBias, 2e-5, "bn fused gradBias")


@pytest.mark.parametrize("shape", [(6, 64, 8, 8), (4, 128, 32, 32), (16, 512, 4, 4), (6, 100, 1, 1)])
@pytest.mark.parametrize("act", ["none", "lrelu"])
def test_batchnorm_batch_groups_equal_separate_calls(shape, act, hipb):
    """vf_bn_train_fwd_groups / vf_bn_bwd_groups over [real; fake] against two vf_bn_train_fwd / vf_bn_bwd calls on the
    halves: same outputs, saved statistics, running averages (updated real first, then fake) and accumulated
    gamma/beta gradients.  The grouped form takes both halves' sums about the SAME shift (the running mean before the
    first update) where the second separate call uses the once-updated one: identical in exact arithmetic, a few fp32
    ulps apart in practice — hence 2e-6, not bitwise."""
    B, C, H, W = shape
    G, h = 2, B // 2
    gen = torch.Generator().manual_seed(B * C + H)
    r = lambda *sh: torch.randn(sh, generator=gen)
    x = to_dev((r(B, C, H, W) * 1.5 + 0.3).numpy(), hipb)
    gy = to_dev(r(B, C, H, W).numpy(), hipb)
    gamma, beta = hipb.from_host(1 + 0.1 * r(C)), hipb.from_host(0.1 * r(C))
    rm0, rv0 = hipb.from_host(0.2 * r(C)), hipb.from_host(1 + 0.1 * r(C).abs())
    slope = 0.2
    f64 = torch.float64
    # separate calls
    rm, rv = rm0.clone(), rv0.clone()
    y = hipb.empty_act(B, C, H, W)
    gx = hipb.empty_act(B, C, H, W)
    sm, si, su = [hipb.zeros(C) for _ in range(G)], [hipb.zeros(C) for _ in range(G)], [hipb.zeros(2 * C, dtype=f64) for _ in range(G)]
    gg, gb = hipb.from_host(r(C)), hipb.from_host(r(C))
    gg0, gb0 = gg.clone(), gb.clone()
    for g in range(G):
        sl = slice(g * h, (g + 1) * h)
        hipb.bn_train_fwd(x[sl], y[sl], gamma, beta, rm, rv, sm[g], si[g], su[g], 0.1, 1e-5, act, slope)
    for g in range(G):
        sl = slice(g * h, (g + 1) * h)
        hipb.bn_bwd(x[sl], y[sl] if act != "none" else None, gy[sl], gx[sl], gg, gb, gamma, sm[g], si[g], su[g], act, slope,
                    0.5 if g == 0 else 1.0)
    # grouped
    rm2, rv2 = rm0.clone(), rv0.clone()
    y2, gx2 = hipb.empty_act(B, C, H, W), hipb.empty_act(B, C, H, W)
    gm, gs, gsum = hipb.zeros(G * C), hipb.zeros(G * C), hipb.zeros(G * 2 * C, dtype=f64)
    gg2, gb2 = gg0.clone(), gb0.clone()
    hipb.bn_train_fwd_groups(x, y2, gamma, beta, rm2, rv2, gm, gs, gsum, G, 0.1, 1e-5, act, slope)
    hipb.bn_bwd_groups(x, y2 if act != "none" else None, gy, gx2, gg2, gb2, gamma, gm, gs, gsum, G, act, slope, 0.5)
    tol = 2e-6 if h * H * W >= 64 else 5e-5      # three samples per channel: x-hat itself is ill-conditioned
    assert rel_err(to_np(y2), to_np(y)) < tol
    assert rel_err(to_np(gx2), to_np(gx)) < 5 * tol
    assert rel_err(to_np(gm), to_np(torch.cat(sm))) < tol and rel_err(to_np(gs), to_np(torch.cat(si))) < tol
    assert rel_err(to_np(rm2), to_np(rm)) < tol and rel_err(to_np(rv2), to_np(rv)) < tol
    assert rel_err(to_np(gg2), to_np(gg)) < 5 * tol and rel_err(to_np(gb2), to_np(gb)) < 5 * tol
    # a pass over ONE group with that group's row of the saved state (fGx's third netD pass)
    gx3 = hipb.empty_act(h, C, H, W)
    hipb.bn_bwd(x[h:], y2[h:] if act != "none" else None, gy[h:], gx3, None, None, gamma, gm[C:], gs[C:], gsum[2 * C:], act, slope, 1.0)
    assert rel_err(to_np(gx3), to_np(gx[h:])) < 5 * tol


@pytest.mark.parametrize("case", [(2, 64, 16, 64, "lrelu"), (3, 16, 8, 32, "relu"), (64, 64, 64, 64, "lrelu"), (5, 64, 4, 128, "lrelu"),
                                  (2, 12, 16, 32, "lrelu")])
def test_conv_bwd_data_with_input_activation_backward(case, oracle, hipb):
    """vf_conv2d_bwd_data_act: gx = (W^T gy) .* act'(x) in the epilogue (direct, split-K and thin-output paths) equals the
    two passes of the reference (SpatialConvolution:updateGradInput, then LeakyReLU:updateGradInput)."""
    B, Cin, H, Cout, act = case
    rng = np.random.default_rng(B * 131 + Cin)
    ref = oracle.SpatialConvolution(Cin, Cout, 4, 4, 2, 2, 1, 1)
    ref.weight[...] = _rand(rng, *ref.weight.shape) * 0.05
    x = _rand(rng, B, Cin, H, H)
    x = np.where(x > 0, x, 0.2 * x if act == "lrelu" else 0.0).astype(np.float32)      # an activated tensor
    gy = _rand(rng, B, Cout, H // 2, H // 2)
    ref.forward(x)
    gx = ref.updateGradInput(x, gy)
    want = np.where(x > 0, gx, gx * np.float32(0.2) if act == "lrelu" else np.float32(0))
    dgx = hipb.empty_act(*x.shape)
    hipb.conv2d_bwd_data_act(to_dev(gy, hipb), to_dev(ref.weight, hipb), dgx, to_dev(x, hipb), act, 0.2, 4, 2, 1)
    assert_close(to_np(dgx), want, TOL, "bwd_data_act %s" % (case,))


def test_weight_gradient_group_equals_individual_launches(oracle, hipb):
    """vf_wgrad_group_begin/_end: recorded weight gradients (several layers, split-K and not, conv and full-conv,
    accumulate and overwrite) equal the per-layer launches bit for bit; an abandoned group is dropped by the next begin."""
    import torch
    rng = np.random.default_rng(17)
    layers = [(False, 4, 64, 16, 128, 2, 1, 0.0), (False, 4, 128, 8, 256, 2, 1, 1.0), (True, 4, 256, 4, 128, 2, 1, 0.0),
              (False, 4, 512, 4, 100, 1, 0, 0.0), (True, 4, 100, 1, 512, 1, 0, 1.0)]
    args = []
    for full, B, Cin, H, Cout, s, p, beta in layers:
        Ho = ((H - 1) * s - 2 * p + 4) if full else ((H + 2 * p - 4) // s + 1)
        x = to_dev(_rand(rng, B, Cin, H, H), hipb)
        gy = to_dev(_rand(rng, B, Cout, Ho, Ho), hipb)
        wshape = (Cin, Cout, 4, 4) if full else (Cout, Cin, 4, 4)
        gw0 = _rand(rng, *wshape)
        args.append((full, x, gy, gw0, s, p, beta))

    def run(grouped):
        outs = []
        if grouped:
            hipb.wgrad_group_begin()
        for full, x, gy, gw0, s, p, beta in args:
            gw = to_dev(gw0, hipb)
            (hipb.deconv2d_bwd_weight if full else hipb.conv2d_bwd_weight)(x, gy, gw, None, 4, s, p, beta)
            outs.append(gw)
        if grouped:
            hipb.wgrad_group_end()
        torch.cuda.synchronize()
        return [to_np(g) for g in outs]

    a, b = run(False), run(True)
    for u, v in zip(a, b):
        np.testing.assert_array_equal(u, v)
    # abandoned group: begin, record one layer, never end; the next begin starts clean
    hipb.wgrad_group_begin()
    full, x, gy, gw0, s, p, beta = args[0]
    hipb.conv2d_bwd_weight(x, gy, to_dev(gw0, hipb), None, 4, s, p, beta)
    c = run(True)
    for u, v in zip(a, c):
        np.testing.assert_array_equal(u, v)


# ------------------------------------------------------------------------------------------------ BatchNorm statistics from the neighbouring GEMMs
def _fused_bn_net(kind, widths, hipb, seed):
    from video_filler_amd import nn
    from video_filler_amd.trainers import weights_init
    net = nn.Sequential()
    mk = (lambda a, b: nn.SpatialFullConvolution(a, b, 4, 4, 2, 2, 1, 1)) if kind == "full" else (
        lambda a, b: nn.SpatialConvolution(a, b, 4, 4, 2, 2, 1, 1))
    for i in range(len(widths) - 1):
        net.add(mk(widths[i], widths[i + 1])).add(nn.SpatialBatchNormalization(widths[i + 1]))
        net.add(nn.ReLU(True) if kind == "full" else nn.LeakyReLU(0.2, True))
    weights_init(net, torch.Generator().manual_seed(seed))
    for m in net.leaves():          # non-trivial conv biases and BatchNorm shifts
        if m.parameters() and m.bias is not None:
            m.bias.copy_(0.1 * torch.randn(m.bias.shape, generator=torch.Generator().manual_seed(seed + 1)).to(hipb.device))
    net.getParameters()
    return net


@pytest.mark.parametrize("kind,widths,Bn,H,groups", [
    ("conv", (64, 64, 128, 256), 16, 64, 1),       # epilogue statistics (one launch per layer)
    ("conv", (64, 128, 256, 512), 16, 32, 2),      # two batch groups (netD's real + fake passes as one batch); deeper layers split K
    ("full", (512, 256, 128, 64), 16, 4, 1),       # transposed passes: four output-parity classes per row tile; split-K combine
    ("conv", (32, 48, 96), 6, 32, 1),              # channel counts that are not powers of two (C % 16 == 0 still)
    ("conv", (16, 20, 40), 3, 16, 1),              # C % 16 != 0: the scalar-gather tiles carry the same epilogue
    ("full", (128, 64, 64), 32, 16, 1),            # transposed passes into 64 channels: the patch kernel's epilogue (forward sums)
    ("conv", (64, 64, 64, 128), 32, 64, 2),        # ... and its backward sums + derivative mask (data-gradients into 64 channels)
    ("conv", (128, 128, 128), 32, 64, 1),          # ... into 128 channels (two column slices per patch tile)
])
def test_batchnorm_statistics_from_the_neighbouring_gemms(kind, widths, Bn, H, groups, hipb):
    """nn.Sequential lets the convolution in front of a BatchNorm sum that BatchNorm's forward statistics in its own
    epilogue (or split-K combine), and the data-gradient pass behind it sum its backward statistics and store the
    gradient already masked by the activation's derivative (vf_bn_fuse_next_fwd / _bwd, vf_bn_*_pre).  Same arithmetic
    per element, another summation order: everything agrees with the separate statistics passes to fp32 rounding, and
    the separate passes are really gone from the launch list."""
    from video_filler_amd import nn
    g = torch.Generator().manual_seed(5)
    x = torch.randn((Bn, widths[0], H, H), generator=g).to(hipb.device).contiguous(memory_format=torch.channels_last)
    res = {}
    for fused in (False, True):
        nn._NO_BN_FUSE = not fused
        try:
            net = _fused_bn_net(kind, widths, hipb, 3)
            net.setBatchGroups(groups)
            net.zeroGradParameters()
            hipb.prof_begin()
            y = net.forward(x).clone()
            gy = torch.randn(y.shape, generator=torch.Generator().manual_seed(6)).to(hipb.device).contiguous(memory_format=torch.channels_last)
            gx = net.backward(x, gy).clone()
            names = hipb.prof_end()
        finally:
            nn._NO_BN_FUSE = False
        bns = [m for m in net.leaves() if isinstance(m, nn.SpatialBatchNormalization)]
        res[fused] = dict(y=y, gx=gx, g=net.reference_flat(grads=True).clone(), rm=torch.cat([m.running_mean for m in bns]),
                          rv=torch.cat([m.running_var for m in bns]), names=names)
    a, b = res[False], res[True]
    nb = len(widths) - 1
    assert "bn_stats" in a["names"] and "bn_bwd_stats" in a["names"]
    fin = b["names"].get("bn_finalize", {}).get("launches", 0)
    bfin = b["names"].get("bn_bwd_finalize", {}).get("launches", 0)
    if all(w in (64, 128, 256, 512) for w in widths[1:]):       # (other widths may fall back under split-K: 256 % N != 0)
        assert fin == nb and "bn_stats" not in b["names"], b["names"].keys()
        # backward: every BatchNorm but the top one has a data-gradient pass above it
        assert bfin == nb - 1 and b["names"].get("bn_bwd_stats", {}).get("launches", 0) == 1
    if Bn == 32:                                                # the cases meant for k_pconv_patch_tr really ran it
        assert any(k.startswith("pconv_patch") for k in b["names"]), b["names"].keys()
    # (other shapes may fall back — e.g. split-K with 256 % N != 0 — and must simply agree)
    for k, tol in (("y", 2e-5), ("gx", 1e-4), ("g", 1e-4), ("rm", 2e-5), ("rv", 2e-5)):
        e = float((a[k] - b[k]).abs().max() / (a[k].abs().max() + 1e-30))
        assert e <= tol, (k, e)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(2, 3, 8, 8), (4, 12, 32, 32), (2, 48, 128, 128)])
def test_gdl_backward(shape, oracle, hipb):
    """vf_gdl_bwd (gdl_criterion.lua:47-53) against the oracle: same four pairings per element, so bit-exact up to the order of at
    most four additions; and through nn.GDLCriterion.backward."""
    from video_filler_amd import nn
    rng = np.random.default_rng(3)
    x = rng.standard_normal(shape).astype(np.float32)
    t = rng.standard_normal(shape).astype(np.float32)
    want = oracle.GDLCriterion(1).backward(x, t)
    crit = nn.GDLCriterion(1)
    dx, dt = to_dev(x, hipb), to_dev(t, hipb)
    got = to_np(crit.backward(dx, dt))
    norm = 1.0 / (shape[0] * shape[1] * (shape[2] - 1) * shape[3])
    assert np.abs(got - want).max() <= 1e-6 * norm * 4
    assert abs(float(crit.forward(dx, dt)) - oracle.GDLCriterion(1).forward(x, t)) < 1e-6
    with pytest.raises(Exception):
        hipb.gdl_bwd(hipb.empty_act(1, 3, 8, 4), hipb.empty_act(1, 3, 8, 4), hipb.empty_act(1, 3, 8, 4))


@pytest.mark.gpu
@pytest.mark.parametrize("Bn,nB,C8", [(4, 6400, 96), (16, 4000, 64), (3, 200, 32), (1, 128, 8), (32, 1000, 16)])
def test_bottleneck_weight_gradients_small_batch(Bn, nB, C8, hipb):
    """vf_wgrad_small.hip: the weight gradients of the bottleneck pair (train.lua:104,134) with K = batch <= 32 — fp32 MFMAs
    straight from the K-major operands — against fp64, overwrite and accumulate, ragged row tiles (nB % 64 != 0), odd batch."""
    g = torch.Generator().manual_seed(5)
    r = lambda *s: torch.randn(*s, generator=g).to(hipb.device)
    # conv C8 -> nB, 4x4 stride 1 pad 0 on a 4x4 map
    x = r(Bn, 4, 4, C8).permute(0, 3, 1, 2)
    gy = r(Bn, 1, 1, nB).permute(0, 3, 1, 2)
    want = torch.einsum("bn,bhwc->nchw", gy.double().reshape(Bn, nB).cpu(), x.permute(0, 2, 3, 1).double().cpu())
    for beta in (0.0, 1.0):
        gw0 = r(nB, 4, 4, C8) * 0.3
        gw = gw0.clone().permute(0, 3, 1, 2)
        hipb.conv2d_bwd_weight(x, gy, gw, None, 4, 1, 0, beta)
        ref = want + beta * gw0.permute(0, 3, 1, 2).double().cpu()
        assert float((gw.cpu().double() - ref).abs().max() / ref.abs().max()) < 2e-6
    # full-conv nB -> C8, 1x1 -> 4x4
    x2 = r(Bn, 1, 1, nB).permute(0, 3, 1, 2)
    gy2 = r(Bn, 4, 4, C8).permute(0, 3, 1, 2)
    want2 = torch.einsum("bn,bhwc->nchw", x2.double().reshape(Bn, nB).cpu(), gy2.permute(0, 2, 3, 1).double().cpu())
    gw2 = hipb.zeros(nB, 4, 4, C8).permute(0, 3, 1, 2)
    hipb.deconv2d_bwd_weight(x2, gy2, gw2, None, 4, 1, 0, 0.0)
    assert float((gw2.cpu().double() - want2).abs().max() / want2.abs().max()) < 2e-6


@pytest.mark.gpu
@pytest.mark.parametrize("Bn,nB,C8", [(4, 640, 64), (3, 128, 128), (8, 96, 64), (1, 64, 64), (5, 3200, 192), (4, 200, 32), (9, 128, 64)])
def test_bottleneck_forward_and_data_gradient_small_batch(Bn, nB, C8, hipb):
    """vf_smallm.hip: the four bottleneck passes (train.lua:104 conv on the 4x4 map, :134 full-conv from the 1x1 map; forward and
    data-gradient of each) at batchSize <= 8 — weight-streaming row-dot / axpy kernels + the split-K combine (bias, activation) —
    against fp64; odd batch sizes, and shapes / batch sizes the kernels do not take (those stay on the tiled kernel: same bars)."""
    g = torch.Generator().manual_seed(Bn * 100 + C8)
    r = lambda *s: torch.randn(*s, generator=g).to(hipb.device)
    tol = 2e-5
    rel = lambda got, want: float((got.double().cpu() - want).abs().max() / want.abs().max())
    # conv C8 -> nB on a 4x4 map (weights physical [nB][4][4][C8])
    x = r(Bn, 4, 4, C8).permute(0, 3, 1, 2)
    w = (r(nB, 4, 4, C8) * 0.05).permute(0, 3, 1, 2)
    bias = r(nB)
    y = hipb.empty_act(Bn, nB, 1, 1)
    hipb.conv2d_fwd(x, w, bias, y, 4, 1, 0, "lrelu", 0.2)
    want = torch.einsum("bchw,nchw->bn", x.double().cpu(), w.double().cpu()) + bias.double().cpu()
    want = torch.where(want > 0, want, 0.2 * want)
    assert rel(y.reshape(Bn, nB), want) < tol
    gy = r(Bn, 1, 1, nB).permute(0, 3, 1, 2)
    gx = hipb.empty_act(Bn, C8, 4, 4)
    hipb.conv2d_bwd_data(gy, w, gx, 4, 1, 0)
    want = torch.einsum("bn,nchw->bchw", gy.double().cpu().reshape(Bn, nB), w.double().cpu())
    assert rel(gx, want) < tol
    # full-conv nB -> C8 from the 1x1 map (weights physical [nB][4][4][C8])
    x2 = r(Bn, 1, 1, nB).permute(0, 3, 1, 2)
    w2 = (r(nB, 4, 4, C8) * 0.05).permute(0, 3, 1, 2)
    b2 = r(C8)
    y2 = hipb.empty_act(Bn, C8, 4, 4)
    hipb.deconv2d_fwd(x2, w2, b2, y2, 4, 1, 0, "relu", 0.0)
    want = torch.einsum("bi,iohw->bohw", x2.double().cpu().reshape(Bn, nB), w2.double().cpu()) + b2.double().cpu().view(1, C8, 1, 1)
    assert rel(y2, want.clamp(min=0)) < tol
    gy2 = r(Bn, 4, 4, C8).permute(0, 3, 1, 2)
    gx2 = hipb.empty_act(Bn, nB, 1, 1)
    hipb.deconv2d_bwd_data(gy2, w2, gx2, 4, 1, 0)
    want = torch.einsum("bohw,iohw->bi", gy2.double().cpu(), w2.double().cpu())
    assert rel(gx2.reshape(Bn, nB), want) < tol


@pytest.mark.gpu
def test_bce_forward_and_backward_in_one_launch(hipb):
    """vf_bce_fwd_bwd against vf_bce_fwd + vf_bce_bwd: one group, and two groups with their own labels (netD's real and fake
    halves) — element for element the same arithmetic, so bitwise."""
    g = torch.Generator().manual_seed(9)
    x = torch.rand(2 * 37, generator=g).to(hipb.device)
    x[0], x[1] = 0.0, 1.0                                     # the eps = 1e-12 edges (train.lua:204)
    loss = hipb.zeros(4, dtype=torch.float64)
    gx_ref, gx = hipb.zeros(2 * 37), hipb.zeros(2 * 37)
    hipb.bce_fwd(x[:37], 1.0, loss[0:1])
    hipb.bce_fwd(x[37:], 0.0, loss[1:2])
    hipb.bce_bwd(x[:37], 1.0, gx_ref[:37])
    hipb.bce_bwd(x[37:], 0.0, gx_ref[37:])
    hipb.bce_fwd_bwd(x, 1.0, 0.0, 37, 2, loss[2:3], loss[3:4], gx)
    assert torch.equal(gx, gx_ref) and torch.equal(loss[:2], loss[2:])
    hipb.bce_fwd_bwd(x[37:], 0.0, 0.0, 37, 1, loss[2:3], None, gx[:37])
    assert torch.equal(gx[:37], gx_ref[37:]) and float(loss[2]) == float(loss[1])
    with pytest.raises(Exception):
        hipb.bce_fwd_bwd(x, 1.0, 0.0, 37, 3, loss[2:3], loss[3:4], gx)
