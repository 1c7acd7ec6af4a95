"""The three matrix-core modes of vf_ctx_set_mfma_mode: 3 = fp32 operands as three exact bf16 planes (default, fp32-grade),
0 = native f32 MFMA, 1 = bf16-rounded operands (opt-in, never the default).  First the opt-in mode:

Stated tolerance: the operands of every conv / full-conv pass are rounded to bf16 (round-to-nearest-even) on their way
into LDS; products and sums are fp32.  So (a) against the oracle fed the SAME bf16-rounded operands the result is
fp32-exact up to summation order (2e-5 of the max-norm, as for the fp32 path), and (b) against the oracle on the
unrounded operands it differs by operand rounding, 2^-9 relative per operand: bounded here by 1e-2 of the max-norm.
Bias, BatchNorm, activations, criteria and Adam are untouched fp32."""
import numpy as np
import torch
import pytest

from helpers import assert_close, to_dev, to_np

pytestmark = pytest.mark.gpu


def bf16_round(a):
    u = np.ascontiguousarray(a, np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32).reshape(a.shape)


@pytest.fixture()
def bf16_backend(hipb):
    prev = hipb.mfma_mode
    hipb.set_mfma_mode("bf16")
    yield hipb
    hipb.set_mfma_mode(prev)


CASES = [  # (full, B, Cin, H, Cout, stride, pad)
    (False, 2, 64, 16, 64, 2, 1), (False, 3, 16, 8, 32, 2, 1), (False, 5, 64, 4, 128, 2, 1), (False, 2, 128, 4, 100, 1, 0),
    (False, 2, 3, 16, 64, 2, 1), (False, 1, 48, 8, 64, 2, 1), (False, 64, 64, 32, 128, 2, 1),
    (True, 2, 100, 1, 128, 1, 0), (True, 2, 128, 4, 64, 2, 1), (True, 2, 64, 8, 3, 2, 1), (True, 5, 16, 2, 16, 2, 1),
    (True, 64, 256, 8, 128, 2, 1),
]


@pytest.mark.parametrize("case", CASES)
def test_bf16_operand_mode(case, oracle, bf16_backend):
    hipb = bf16_backend
    full, B, Cin, H, Cout, s, p = case
    rng = np.random.default_rng(abs(hash(case)) % (2 ** 31))
    r = lambda *sh: rng.standard_normal(sh).astype(np.float32)
    mk = oracle.SpatialFullConvolution if full else oracle.SpatialConvolution
    w, bias, x = r(*mk(Cin, Cout, 4, 4, s, s, p, p).weight.shape) * 0.05, r(Cout), r(B, Cin, H, H)

    def ref_passes(xx, ww, gy_fn):
        m = mk(Cin, Cout, 4, 4, s, s, p, p)
        m.weight[...] = ww
        m.bias[...] = bias
        y = np.array(m.forward(xx), copy=True)
        gy = gy_fn(y.shape)
        return m, y, gy

    gy_holder = {}

    def gy_fn(shape):
        if "gy" not in gy_holder:
            gy_holder["gy"] = r(*shape)
        return gy_holder["gy"]

    # exact-operand reference
    m_exact, y_exact, gy = ref_passes(x, w, gy_fn)
    m_exact.gradWeight[...] = 0
    m_exact.gradBias[...] = 0
    m_exact.backward(x, gy)
    # rounded-operand reference: each pass rounds exactly the two operands the kernel stages
    xr, wr, gyr = bf16_round(x), bf16_round(w), bf16_round(gy)
    m_f, y_r, _ = ref_passes(xr, wr, gy_fn)
    m_d = mk(Cin, Cout, 4, 4, s, s, p, p)
    m_d.weight[...] = wr
    m_d.forward(xr)
    gx_r = np.array(m_d.updateGradInput(xr, gyr), copy=True)
    m_w = mk(Cin, Cout, 4, 4, s, s, p, p)
    m_w.weight[...] = wr
    m_w.forward(xr)
    m_w.gradWeight[...] = 0
    m_w.gradBias[...] = 0
    m_w.accGradParameters(xr, gyr)

    fwd, bwd_d, bwd_w = ((hipb.deconv2d_fwd, hipb.deconv2d_bwd_data, hipb.deconv2d_bwd_weight) if full else
                         (hipb.conv2d_fwd, hipb.conv2d_bwd_data, hipb.conv2d_bwd_weight))
    dx, dw, db, dgy = to_dev(x, hipb), to_dev(w, hipb), to_dev(bias, hipb), to_dev(gy, hipb)
    dy = hipb.empty_act(*y_exact.shape)
    fwd(dx, dw, db, dy, 4, s, p)
    assert_close(to_np(dy), y_r, 3e-5, "bf16 fwd vs rounded-operand oracle %s" % (case,))
    assert_close(to_np(dy), y_exact, 1e-2, "bf16 fwd vs exact oracle %s" % (case,))
    dgx = hipb.empty_act(*x.shape)
    bwd_d(dgy, dw, dgx, 4, s, p)
    assert_close(to_np(dgx), gx_r, 3e-5, "bf16 bwd_data vs rounded %s" % (case,))
    assert_close(to_np(dgx), m_exact.gradInput, 1e-2, "bf16 bwd_data vs exact %s" % (case,))
    dgw, dgb = hipb.zeros(*w.shape).contiguous(memory_format=__import__("torch").channels_last), hipb.zeros(Cout)
    dgw = to_dev(np.zeros_like(w), hipb)
    bwd_w(dx, dgy, dgw, dgb, 4, s, p, 0.0)
    assert_close(to_np(dgw), m_w.gradWeight, 3e-5, "bf16 bwd_weight vs rounded %s" % (case,))
    assert_close(to_np(dgw), m_exact.gradWeight, 1e-2, "bf16 bwd_weight vs exact %s" % (case,))
    assert_close(to_np(dgb), m_exact.gradBias, 2e-5, "bias gradient stays fp32 %s" % (case,))


def test_mode_is_per_context_and_validated(hipb):
    from video_filler_amd import _lib
    from video_filler_amd.backend import DEFAULT_MFMA_MODE
    prev = hipb.mfma_mode
    assert DEFAULT_MFMA_MODE == "f32_3xbf16"
    with pytest.raises(RuntimeError):
        _lib.check(hipb.lib.vf_ctx_set_mfma_mode(hipb.ctx, 7))
    side = hipb.fork(workspace_bytes=8 << 20)
    assert side.mfma_mode == prev
    hipb.set_mfma_mode("bf16")
    assert side.mfma_mode == "bf16"
    hipb.set_mfma_mode(prev)
    assert side.mfma_mode == prev and hipb.mfma_mode == prev


@pytest.fixture()
def native_backend(hipb):
    prev = hipb.mfma_mode
    hipb.set_mfma_mode("f32")
    yield hipb
    hipb.set_mfma_mode(prev)


@pytest.mark.parametrize("case", CASES[:7])
def test_native_f32_mfma_mode_still_matches(case, oracle, native_backend):
    """Mode 0 (v_mfma_f32_32x32x2_f32) stays available and correct at the same tolerance."""
    hipb = native_backend
    full, B, Cin, H, Cout, s, p = case
    rng = np.random.default_rng(abs(hash(case)) % (2 ** 31))
    r = lambda *sh: rng.standard_normal(sh).astype(np.float32)
    m = oracle.SpatialConvolution(Cin, Cout, 4, 4, s, s, p, p)
    m.weight[...] = r(*m.weight.shape) * 0.05
    m.bias[...] = r(Cout)
    x = r(B, Cin, H, H)
    y = np.array(m.forward(x), copy=True)
    gy = r(*y.shape)
    m.gradWeight[...] = 0
    m.gradBias[...] = 0
    m.backward(x, gy)
    dx, dw, db, dgy = to_dev(x, hipb), to_dev(m.weight, hipb), to_dev(m.bias, hipb), to_dev(gy, hipb)
    dy = hipb.empty_act(*y.shape)
    hipb.conv2d_fwd(dx, dw, db, dy, 4, s, p)
    assert_close(to_np(dy), y, 2e-5, "native fwd %s" % (case,))
    dgx = hipb.empty_act(*x.shape)
    hipb.conv2d_bwd_data(dgy, dw, dgx, 4, s, p)
    assert_close(to_np(dgx), m.gradInput, 2e-5, "native bwd_data %s" % (case,))
    dgw, dgb = to_dev(np.zeros_like(m.weight), hipb), hipb.zeros(Cout)
    hipb.conv2d_bwd_weight(dx, dgy, dgw, dgb, 4, s, p, 0.0)
    assert_close(to_np(dgw), m.gradWeight, 2e-5, "native bwd_weight %s" % (case,))


def test_bf16_training_iteration_tracks_fp32(oracle, bf16_backend):
    """One full iteration of the train.lua closures in bf16-operand mode stays within operand-rounding distance of the
    fp32 oracle: losses within 2e-2 relative, gradients within 5e-2 of their max-norm (smooth nets)."""
    import torch
    from video_filler_amd.trainers import CenterTrainer
    from test_gpu_trainers import _load
    opt = dict(nBottleneck=64, wtl2=0.999, overlapPred=4, smooth=True)
    ref = oracle.CenterTrainer(opt, np.random.default_rng(1))
    tr = CenterTrainer(opt, seed=3)
    _load(tr, ref)
    batch = oracle.synth_center_batch(4, np.random.default_rng(9))
    ref.set_batch(batch)
    tr.set_batch(torch.from_numpy(batch))
    ref.step()
    tr.step()
    got = tr.losses()
    for k in ("errD", "errG", "errG_l2"):
        assert abs(got[k] - getattr(ref, k)) < 2e-2 * max(1.0, abs(getattr(ref, k))), (k, got[k], getattr(ref, k))
    gG = tr.netG.reference_flat(grads=True).cpu().numpy()
    assert np.abs(gG - ref.gradParametersG).max() < 5e-2 * np.abs(ref.gradParametersG).max()


@pytest.fixture()
def x3_backend(hipb):
    prev = hipb.mfma_mode
    hipb.set_mfma_mode("f32_3xbf16")
    yield hipb
    hipb.set_mfma_mode(prev)


@pytest.mark.parametrize("case", CASES)
def test_three_plane_split_is_fp32_grade(case, oracle, x3_backend):
    """Mode 3: every fp32 operand is split EXACTLY into three bf16 planes (3 x 8 = 24 significand bits) and the six
    largest cross terms are accumulated in fp32 on the bf16 matrix pipe.  Stated tolerance: the fp32 path's own
    (2e-5 of the max-norm against the exact oracle) — no operand rounding is left."""
    hipb = x3_backend
    full, B, Cin, H, Cout, s, p = case
    rng = np.random.default_rng(abs(hash(case)) % (2 ** 31))
    r = lambda *sh: rng.standard_normal(sh).astype(np.float32)
    m = (oracle.SpatialFullConvolution if full else oracle.SpatialConvolution)(Cin, Cout, 4, 4, s, s, p, p)
    m.weight[...] = r(*m.weight.shape) * 0.05
    m.bias[...] = r(Cout)
    x = r(B, Cin, H, H)
    y = np.array(m.forward(x), copy=True)
    gy = r(*y.shape)
    m.gradWeight[...] = 0
    m.gradBias[...] = 0
    m.backward(x, gy)
    fwd, bwd_d, bwd_w = ((hipb.deconv2d_fwd, hipb.deconv2d_bwd_data, hipb.deconv2d_bwd_weight) if full else
                         (hipb.conv2d_fwd, hipb.conv2d_bwd_data, hipb.conv2d_bwd_weight))
    dx, dw, db, dgy = to_dev(x, hipb), to_dev(m.weight, hipb), to_dev(m.bias, hipb), to_dev(gy, hipb)
    dy = hipb.empty_act(*y.shape)
    fwd(dx, dw, db, dy, 4, s, p)
    assert_close(to_np(dy), y, 2e-5, "x3 fwd %s" % (case,))
    dgx = hipb.empty_act(*x.shape)
    bwd_d(dgy, dw, dgx, 4, s, p)
    assert_close(to_np(dgx), m.gradInput, 2e-5, "x3 bwd_data %s" % (case,))
    dgw, dgb = to_dev(np.zeros_like(m.weight), hipb), hipb.zeros(Cout)
    bwd_w(dx, dgy, dgw, dgb, 4, s, p, 0.0)
    assert_close(to_np(dgw), m.gradWeight, 2e-5, "x3 bwd_weight %s" % (case,))


def test_three_plane_training_iteration_matches_fp32_tolerances(oracle, x3_backend):
    """The whole train.lua iteration in mode 3 passes the fp32 path's end-to-end tolerances (smooth nets: losses 2e-5,
    gradients 1e-4 of the max-norm)."""
    import torch
    from video_filler_amd.trainers import CenterTrainer
    from test_gpu_trainers import _load
    opt = dict(nBottleneck=64, wtl2=0.999, overlapPred=4, smooth=True)
    ref = oracle.CenterTrainer(opt, np.random.default_rng(1))
    tr = CenterTrainer(opt, seed=3)
    _load(tr, ref)
    batch = oracle.synth_center_batch(4, np.random.default_rng(9))
    ref.set_batch(batch)
    tr.set_batch(torch.from_numpy(batch))
    ref.step()
    tr.step()
    got = tr.losses()
    for k in ("errD", "errG", "errG_l2"):
        assert abs(got[k] - getattr(ref, k)) < 2e-5 * max(1.0, abs(getattr(ref, k))), (k, got[k], getattr(ref, k))
    for net, want in ((tr.netG, ref.gradParametersG), (tr.netD, ref.gradParametersD)):
        g = net.reference_flat(grads=True).cpu().numpy()
        assert np.abs(g - want).max() < 1e-4 * np.abs(want).max()


@pytest.mark.parametrize("scale", [1e-18, 1.0, 1e18])
def test_three_plane_split_across_the_exponent_range(scale, oracle, x3_backend):
    """bf16 has fp32's exponent range, so the exact split must hold for very small and very large operands alike
    (products up to 1e36 and down to 1e-36 here); relative error stays at the fp32 level."""
    hipb = x3_backend
    rng = np.random.default_rng(5)
    r = lambda *sh: rng.standard_normal(sh).astype(np.float32)
    m = oracle.SpatialConvolution(64, 64, 4, 4, 2, 2, 1, 1)
    m.weight[...] = r(*m.weight.shape) * np.float32(scale)
    m.bias[...] = 0
    x = r(2, 64, 16, 16) * np.float32(scale)
    y = np.array(m.forward(x), copy=True)
    assert np.isfinite(y).all() and np.abs(y).max() > 0
    dy = hipb.empty_act(*y.shape)
    hipb.conv2d_fwd(to_dev(x, hipb), to_dev(m.weight, hipb), to_dev(m.bias, hipb), dy, 4, 2, 1)
    assert_close(to_np(dy), y, 2e-5, "x3 fwd at scale %g" % scale)


def test_three_plane_mode_outside_the_normal_range_is_as_documented(x3_backend):
    """include/vf_hip.h (vf_ctx_set_mfma_mode): what the default product mode does with operands no training run produces.
    An infinite operand element turns the outputs it reaches into NaN (the split forms inf - inf) where the native fp32
    mode gives inf; tiny operands lose their low planes, so they are carried to bf16-level relative accuracy, and fp32
    subnormals count as zero.  Both the igemm path (small layer) and the planes path (>= 1024 rows) are probed."""
    hipb = x3_backend
    rng = np.random.default_rng(6)
    for B_, H in ((1, 8), (4, 32)):
        x = rng.standard_normal((B_, 64, H, H)).astype(np.float32)
        w = rng.standard_normal((64, 64, 4, 4)).astype(np.float32)
        xi = x.copy()
        xi[0, 5, 3, 3] = np.inf
        y3 = hipb.empty_act(B_, 64, H // 2, H // 2)
        hipb.conv2d_fwd(to_dev(xi, hipb), to_dev(w, hipb), None, y3, 4, 2, 1)
        y3 = to_np(y3)
        hipb.set_mfma_mode("f32")
        try:
            y0 = hipb.empty_act(B_, 64, H // 2, H // 2)
            hipb.conv2d_fwd(to_dev(xi, hipb), to_dev(w, hipb), None, y0, 4, 2, 1)
            y0 = to_np(y0)
        finally:
            hipb.set_mfma_mode("f32_3xbf16")
        touched = ~np.isfinite(y0)
        assert touched.any() and np.isinf(y0[touched]).all(), "native fp32: +-inf where the infinite element is read"
        assert np.isnan(y3[touched]).all(), "three-plane mode: NaN in the same places"
        assert np.array_equal(np.isfinite(y3), ~touched), "and nowhere else"
        # tiny operands: bf16-level relative accuracy instead of fp32-level; subnormal operands vanish
        tiny = (x * np.float32(1e-36)).astype(np.float32)
        yt = hipb.empty_act(B_, 64, H // 2, H // 2)
        hipb.conv2d_fwd(to_dev(tiny, hipb), to_dev(w, hipb), None, yt, 4, 2, 1)
        ref = torch.nn.functional.conv2d(torch.from_numpy(tiny).double(), torch.from_numpy(w).double(), stride=2, padding=1).numpy()
        err = np.abs(to_np(yt) - ref).max() / np.abs(ref).max()
        assert err < 2e-2, err
        sub = (x * np.float32(1e-40)).astype(np.float32)
        ys = hipb.empty_act(B_, 64, H // 2, H // 2)
        hipb.conv2d_fwd(to_dev(sub, hipb), to_dev(w, hipb), None, ys, 4, 2, 1)
        assert np.abs(to_np(ys)).max() <= 1e-36
