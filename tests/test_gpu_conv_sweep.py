"""Seeded random sweep of conv / full-conv shapes against the oracle: ragged batches, channel counts that miss every
vector width and tile multiple (3, 6, 27, 33, 100, 130 ...), Cout = 1, both geometries the reference builds
(4x4 stride 2 pad 1; 4x4 stride 1 pad 0 bottleneck), every pass, accumulate and overwrite.  Plus the argument checks
of the C-ABI (the reference's THNN raises on the same misuse)."""
import numpy as np
import pytest

from helpers import assert_close, to_dev, to_np

pytestmark = pytest.mark.gpu

TOL = 2e-5


def _cases(n, seed):
    rng = np.random.default_rng(seed)
    cin = [3, 6, 12, 16, 20, 27, 32, 48, 64, 100, 132]
    cout = [1, 3, 12, 16, 27, 33, 64, 96, 130]
    out = []
    while len(out) < n:
        full = bool(rng.integers(0, 2))
        B, Ci, Co = int(rng.integers(1, 10)), int(rng.choice(cin)), int(rng.choice(cout))
        if rng.random() < 0.25:
            H, s, p = (1 if full else 4), 1, 0            # bottleneck geometry
        else:
            H, s, p = int(rng.choice([2, 4, 8, 16, 32] if full else [4, 8, 16, 32])), 2, 1
        if B * max(Ci, Co) * (H * (2 if full else 1)) ** 2 > 3_000_000:
            continue
        out.append((full, B, Ci, H, Co, s, p))
    return out


@pytest.mark.parametrize("case", _cases(48, 20261003))
def test_random_shapes(case, oracle, hipb):
    full, B, Cin, H, Cout, s, p = case
    rng = np.random.default_rng(abs(hash(case)) % (2 ** 31))
    r = lambda *sh: rng.standard_normal(sh).astype(np.float32)
    ref = (oracle.SpatialFullConvolution if full else oracle.SpatialConvolution)(Cin, Cout, 4, 4, s, s, p, p)
    ref.weight[...] = r(*ref.weight.shape) * 0.05
    ref.bias[...] = r(Cout)
    x = r(B, Cin, H, H)
    y = ref.forward(x)
    gy = r(*y.shape)
    ref.gradWeight[...] = r(*ref.weight.shape)
    ref.gradBias[...] = r(Cout)
    gw0, gb0 = ref.gradWeight.copy(), ref.gradBias.copy()
    ref.backward(x, gy)
    fwd, bwd_d, bwd_w = ((hipb.deconv2d_fwd, hipb.deconv2d_bwd_data, hipb.deconv2d_bwd_weight) if full else
                         (hipb.conv2d_fwd, hipb.conv2d_bwd_data, hipb.conv2d_bwd_weight))
    dx, dw, db = to_dev(x, hipb), to_dev(ref.weight, hipb), to_dev(ref.bias, hipb)
    dy = hipb.empty_act(*y.shape)
    fwd(dx, dw, db, dy, 4, s, p)
    assert_close(to_np(dy), y, TOL, "fwd %s" % (case,))
    dgy, dgx = to_dev(gy, hipb), hipb.empty_act(*x.shape)
    bwd_d(dgy, dw, dgx, 4, s, p)
    assert_close(to_np(dgx), ref.gradInput, TOL, "bwd_data %s" % (case,))
    dgw, dgb = to_dev(gw0, hipb), to_dev(gb0, hipb)
    bwd_w(dx, dgy, dgw, dgb, 4, s, p, 1.0)
    assert_close(to_np(dgw), ref.gradWeight, TOL, "bwd_weight(beta=1) %s" % (case,))
    assert_close(to_np(dgb), ref.gradBias, TOL, "bias grad %s" % (case,))
    bwd_w(dx, dgy, dgw, dgb, 4, s, p, 0.0)
    assert_close(to_np(dgw), ref.gradWeight - gw0, 2 * TOL, "bwd_weight(beta=0) %s" % (case,))
    assert_close(to_np(dgb), ref.gradBias - gb0, 2 * TOL, "bias grad(beta=0) %s" % (case,))


GENERIC = [  # (B, Cin, H, Cout, k, stride, pad) — the option branches' shapes (train.lua:109-113,158-170) and others
    (3, 3, 32, 64, 5, 2, 2),        # conditionAdv context branch (scaled: 128 -> 32)
    (3, 3, 16, 64, 5, 2, 2 + 8),    # conditionAdv prediction branch: pad 2 + 32 on 64x64, scaled to 2 + 8 on 16x16
    (2, 3, 64, 64, 5, 2, 2 + 32),   # the same at full size
    (5, 100, 1, 100, 1, 1, 0),      # noiseGen: nz -> nz, 1x1 on the 1x1 noise map
    (2, 16, 6, 32, 3, 1, 1),        # 3x3 on a map that is not a power of two
    (2, 6, 9, 12, 4, 2, 0),         # 4x4 stride 2 without padding
    (1, 20, 8, 130, 4, 3, 1),       # stride 3; Cout*K beyond the LDS weight stage
    (2, 8, 8, 4, 4, 1, 0),          # 4x4 stride 1 on a map larger than the bottleneck's 4x4
]


@pytest.mark.parametrize("case", GENERIC)
def test_generic_conv_shapes(case, oracle, hipb):
    """nn.SpatialConvolution of any kernel / stride / padding (vf_conv_generic.hip) against the oracle: every pass,
    accumulate and overwrite."""
    B, Cin, H, Cout, k, s, p = case
    rng = np.random.default_rng(abs(hash(case)) % (2 ** 31))
    r = lambda *sh: rng.standard_normal(sh).astype(np.float32)
    ref = oracle.SpatialConvolution(Cin, Cout, k, k, s, s, p, p)
    ref.weight[...] = r(*ref.weight.shape) * 0.05
    ref.bias[...] = r(Cout)
    x = r(B, Cin, H, H)
    y = ref.forward(x)
    gy = r(*y.shape)
    ref.gradWeight[...] = r(*ref.weight.shape)
    ref.gradBias[...] = r(Cout)
    gw0, gb0 = ref.gradWeight.copy(), ref.gradBias.copy()
    ref.backward(x, gy)
    dx, dw, db = to_dev(x, hipb), to_dev(ref.weight, hipb), to_dev(ref.bias, hipb)
    dy = hipb.empty_act(*y.shape)
    hipb.conv2d_fwd(dx, dw, db, dy, k, s, p)
    assert_close(to_np(dy), y, TOL, "fwd %s" % (case,))
    hipb.conv2d_fwd(dx, dw, db, dy, k, s, p, "lrelu", 0.2)
    assert_close(to_np(dy), np.where(y > 0, y, np.float32(0.2) * y), TOL, "fwd+lrelu %s" % (case,))
    dgy, dgx = to_dev(gy, hipb), hipb.empty_act(*x.shape)
    hipb.conv2d_bwd_data(dgy, dw, dgx, k, s, p)
    assert_close(to_np(dgx), ref.gradInput, TOL, "bwd_data %s" % (case,))
    dgw, dgb = to_dev(gw0, hipb), to_dev(gb0, hipb)
    hipb.conv2d_bwd_weight(dx, dgy, dgw, dgb, k, s, p, 1.0)
    assert_close(to_np(dgw), ref.gradWeight, TOL, "bwd_weight(beta=1) %s" % (case,))
    assert_close(to_np(dgb), ref.gradBias, TOL, "bias grad %s" % (case,))
    hipb.conv2d_bwd_weight(dx, dgy, dgw, dgb, k, s, p, 0.0)
    assert_close(to_np(dgw), ref.gradWeight - gw0, 2 * TOL, "bwd_weight(beta=0) %s" % (case,))
    assert_close(to_np(dgb), ref.gradBias - gb0, 2 * TOL, "bias grad(beta=0) %s" % (case,))


def test_c_abi_rejects_what_the_reference_never_builds(hipb):
    """Transposed convolutions: k = 4 only, (stride, pad) in {(2,1), (1,0)}, power-of-two maps (every
    nn.SpatialFullConvolution of the reference); a convolution whose kernel exceeds its padded input; BatchNorm over a
    channel count that misses the vector width: an error, never a wrong result or a fault."""
    x = hipb.empty_act(2, 16, 8, 8)
    w = hipb.empty(16, 4, 4, 32).permute(0, 3, 1, 2)
    b = hipb.zeros(32)
    y = hipb.empty_act(2, 32, 16, 16)
    with pytest.raises(RuntimeError):
        hipb.deconv2d_fwd(x, w, b, y, 3, 2, 1)            # 3x3 kernel
    with pytest.raises(RuntimeError):
        hipb.deconv2d_fwd(x, w, b, y, 4, 2, 0)            # stride 2 without pad 1
    with pytest.raises(RuntimeError):
        hipb.deconv2d_fwd(x, w, b, y, 4, 3, 1)            # stride 3
    x6 = hipb.empty_act(2, 16, 6, 6)                      # 6x6 map: not a power of two
    with pytest.raises(RuntimeError):
        hipb.deconv2d_fwd(x6, w, b, hipb.empty_act(2, 32, 12, 12), 4, 2, 1)
    wc = hipb.empty(32, 4, 4, 16).permute(0, 3, 1, 2)
    with pytest.raises(RuntimeError, match="larger than the padded input"):
        hipb.conv2d_fwd(hipb.empty_act(2, 16, 2, 2), wc, b, hipb.empty_act(2, 32, 1, 1), 4, 1, 0)
    bn_x = hipb.empty_act(2, 6, 4, 4)                     # BatchNorm over a channel count that is not a multiple of 4
    with pytest.raises(RuntimeError, match="multiple of 4"):
        hipb.bn_stats(bn_x, None, hipb.zeros(12, dtype=__import__("torch").float64))


def test_c_abi_rejects_empty_and_oversized_batches(hipb):
    """B = 0 is an error (THNN raises on empty tensors too); operands beyond the 2 GiB range of the hardware buffer
    descriptors the gathers rely on are refused before anything is launched (dimensions only: no such tensor is allocated)."""
    from video_filler_amd.backend import ACT, _ptr
    x = hipb.empty_act(1, 64, 8, 8)
    w = hipb.empty(64, 4, 4, 64).permute(0, 3, 1, 2)
    b = hipb.zeros(64)
    y = hipb.empty_act(1, 64, 4, 4)
    with pytest.raises(RuntimeError, match="bad sizes"):
        hipb._c("vf_conv2d_fwd", _ptr(x), _ptr(w), _ptr(b), _ptr(y), 0, 8, 8, 64, 64, 4, 2, 1, ACT["none"], 0.0)
    with pytest.raises(RuntimeError, match="2 GiB"):
        hipb._c("vf_conv2d_fwd", _ptr(x), _ptr(w), _ptr(b), _ptr(y), 4096, 128, 128, 64, 64, 4, 2, 1, ACT["none"], 0.0)
    with pytest.raises(RuntimeError, match="2 GiB"):
        hipb._c("vf_conv2d_bwd_weight", _ptr(x), _ptr(y), _ptr(w), None, 4096, 128, 128, 64, 64, 4, 2, 1, 0.0)
