"""The convolution passes from operands pre-split into three bf16 planes (csrc/vf_pgemm.hip) against the kernels that split
inside the GEMM (csrc/vf_conv.hip, themselves checked against the oracle in test_gpu_ops.py / test_gpu_conv_sweep.py):
the two form the same six-term products and differ only in the order the K dimension is summed in, so they agree to fp32
rounding — 2e-6 of the output's max-norm — and both sit within 2e-5 of an fp64 evaluation of the same convolution."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rand(shape, seed, dev, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dev)


def _act(Bn, C, H, seed, dev):
    return _rand((Bn, H, H, C), seed, dev).permute(0, 3, 1, 2)


def test_planes_are_an_exact_three_way_split(hipb):
    x = _rand((1 << 16,), 1, hipb.device)
    x[:8] = torch.tensor([0.0, -0.0, 1.0, -1.0, 3.0e38, -1.1754944e-38, 1e-30, 123456.789])
    x[8:4104] = x[8:4104] * torch.logspace(-30, 30, 4096).to(hipb.device)
    pl = hipb.planes_split(x)
    f = pl.float()
    assert torch.equal((f[0] + f[1]) + f[2], x)                  # hi + mid + lo == x, bit for bit
    assert torch.equal(f[0], (x.view(torch.int32) & -65536).view(torch.float32))      # hi = the top 16 bits
    # each plane carries at most 8 significant bits of what the planes before it left over
    assert float((f[1].abs() > f[0].abs() * 2.0 ** -7).sum()) == 0


def test_weight_planes_native_and_transposed(hipb):
    w = (_rand((96, 4, 4, 40), 2, hipb.device, 0.05)).permute(0, 3, 1, 2)          # logical [d0][d1][4][4]
    nat, tr = hipb.weight_planes(w)
    phys = w.permute(0, 2, 3, 1).contiguous()                                         # [d0][4][4][d1]
    f = nat.float().view(3, 96, 16, 40)
    assert torch.equal((f[0] + f[1]) + f[2], phys.view(96, 16, 40))
    t = tr.float().view(3, 40, 16, 96)
    assert torch.equal((t[0] + t[1]) + t[2], phys.view(96, 16, 40).permute(2, 1, 0).contiguous())


CASES = [(8, 64, 64, 64), (8, 64, 32, 128), (16, 128, 16, 256), (16, 256, 8, 512), (4, 192, 32, 384), (6, 32, 32, 96),
         (64, 64, 32, 128), (3, 96, 16, 36),
         (16, 64, 128, 64), (64, 64, 64, 128),      # scatter: >= 512 row tiles into 64 channels — four parity classes per patch block
         (32, 128, 32, 64), (64, 128, 16, 256)]     # scatter: 16 x 16 grid into 128 channels (two column slices per patch tile); 8 x 8 grid


@pytest.mark.parametrize("Bn,Cin,H,Cout", CASES, ids=lambda v: str(v))
def test_gather_pass_matches_the_in_kernel_split(Bn, Cin, H, Cout, hipb):
    """conv forward (with bias + LeakyReLU) and full-conv data-gradient"""
    dev = hipb.device
    x = _act(Bn, Cin, H, 3, dev)
    w = _rand((Cout, 4, 4, Cin), 4, dev, 0.05).permute(0, 3, 1, 2)
    bias = _rand((Cout,), 5, dev, 0.1)
    assert hipb.pconv_supported(Bn, H, H, Cin, Cout, 4, 2, 1, False)
    want = hipb.empty_act(Bn, Cout, H // 2, H // 2)
    hipb.conv2d_fwd(x, w, bias, want, 4, 2, 1, "lrelu", 0.2)
    xp = hipb.planes_split(x)
    wp, _ = hipb.weight_planes(w, want_transposed=False)
    got = hipb.empty_act(Bn, Cout, H // 2, H // 2)
    hipb.pconv_gather(xp, wp, bias, got, Bn, H, H, Cin, Cout, "lrelu", 0.2)
    e = float((got - want).abs().max() / want.abs().max())
    assert e <= 2e-6, e
    if Bn * H * H * Cin <= 1 << 21:
        ref = torch.nn.functional.conv2d(x.double().cpu().contiguous(), w.double().cpu().contiguous(), bias.double().cpu(), stride=2, padding=1)
        ref = torch.nn.functional.leaky_relu(ref, 0.2)
        assert float((got.cpu().double() - ref).abs().max() / ref.abs().max()) <= 2e-5


@pytest.mark.parametrize("Bn,Cin,H,Cout", CASES, ids=lambda v: str(v))
def test_scatter_pass_matches_the_in_kernel_split(Bn, Cin, H, Cout, hipb):
    """conv data-gradient (plain and with the activation-derivative epilogue) and full-conv forward (bias + ReLU): the
    low-res operand has Cout channels on an H/2 grid, the result Cin channels on the H grid"""
    dev = hipb.device
    Hl = H // 2
    gy = _act(Bn, Cout, Hl, 6, dev)
    w = _rand((Cout, 4, 4, Cin), 7, dev, 0.05).permute(0, 3, 1, 2)                   # conv weight [Cout][Cin][4][4]
    if not hipb.pconv_supported(Bn, Hl, Hl, Cout, Cin, 4, 2, 1, True):
        pytest.skip("the gathered operand's channel count is not a multiple of 32")
    want = hipb.empty_act(Bn, Cin, H, H)
    hipb.conv2d_bwd_data(gy, w, want, 4, 2, 1)
    gp = hipb.planes_split(gy)
    _, wt = hipb.weight_planes(w)                                                     # transposed planes [Cin][16][Cout]
    got = hipb.empty_act(Bn, Cin, H, H)
    hipb.pconv_scatter(gp, wt, None, got, Bn, Hl, Hl, Cout, Cin)
    e = float((got - want).abs().max() / want.abs().max())
    assert e <= 2e-6, e
    if Bn * H * H * Cin <= 1 << 21:
        ref = torch.nn.functional.conv_transpose2d(gy.double().cpu().contiguous(), w.double().cpu().contiguous(), None, stride=2, padding=1)
        assert float((got.cpu().double() - ref).abs().max() / ref.abs().max()) <= 2e-5
    # the derivative mask of the (leaky) ReLU that produced this conv's input, in the epilogue
    xact = _act(Bn, Cin, H, 8, dev)
    hipb.conv2d_bwd_data_act(gy, w, want, xact, "lrelu", 0.2, 4, 2, 1)
    hipb.pconv_scatter(gp, wt, None, got, Bn, Hl, Hl, Cout, Cin, dmask=xact, dact="lrelu", dslope=0.2)
    assert float((got - want).abs().max() / want.abs().max()) <= 2e-6
    # full-conv forward: weight [Cin_full = Cout here][Cout_full = Cin here]: the same physical tensor read the other way
    bias = _rand((Cin,), 9, dev, 0.1)
    hipb.deconv2d_fwd(gy, w, bias, want, 4, 2, 1, "relu", 0.0)
    hipb.pconv_scatter(gp, wt, bias, got, Bn, Hl, Hl, Cout, Cin, "relu", 0.0)
    assert float((got - want).abs().max() / want.abs().max()) <= 2e-6


def test_unsupported_shapes_are_reported(hipb):
    assert not hipb.pconv_supported(8, 64, 64, 3, 64, 4, 2, 1, False)       # thin input
    assert not hipb.pconv_supported(8, 64, 64, 48, 64, 4, 2, 1, False)      # 48 channels: not a multiple of 32
    assert not hipb.pconv_supported(8, 32, 32, 64, 3, 4, 2, 1, True)        # thin output
    assert not hipb.pconv_supported(64, 4, 4, 512, 4000, 4, 1, 0, False)    # the bottleneck GEMMs
    assert not hipb.pconv_supported(1, 8, 8, 64, 64, 4, 2, 1, False)        # 16 GEMM rows
    x = hipb.zeros(4 * 8 * 8 * 48)
    with pytest.raises(RuntimeError):
        hipb.pconv_gather(hipb.planes_split(x), hipb.planes_split(hipb.zeros(64 * 16 * 48)), None, hipb.zeros(4 * 4 * 4 * 64), 4, 8, 8, 48, 64)


# (Bn, Cin, H, Cout): conv Cin -> Cout on an H x H map.  The planes weight gradient needs the low-resolution operand's channels
# (conv: Cout; full-conv: its input's) % 128 == 0, the gathered one's % 64 == 0 and a multiple of 32 pixels — the other rows
# must come out of the fp32 kernels untouched by the planes arguments
WCASES = [(8, 64, 32, 128), (16, 128, 16, 256), (16, 256, 8, 512), (64, 64, 32, 128), (4, 192, 32, 384), (2, 64, 8, 128),
          (8, 64, 64, 64), (6, 32, 32, 96)]


@pytest.mark.parametrize("grouped", [False, True])
@pytest.mark.parametrize("Bn,Cin,H,Cout", WCASES, ids=lambda v: str(v))
def test_weight_gradient_from_planes(Bn, Cin, H, Cout, grouped, hipb):
    """vf_conv2d_bwd_weight_planes / vf_deconv2d_bwd_weight_planes against the fp32-operand kernels (same six-term products,
    another summation order: 2e-6) and fp64 torch; overwrite and accumulate; alone and inside a weight-gradient group."""
    dev = hipb.device
    x = _act(Bn, Cin, H, 3, dev)
    gy = _act(Bn, Cout, H // 2, 4, dev)
    xp, gp = hipb.planes_split(x), hipb.planes_split(gy)
    for full in (False, True):
        # conv: x is the high-resolution input, gy the low-resolution gradient; the full-conv with the SAME pair of maps has the
        # low-resolution tensor as its input and the high-resolution one as its gradOutput
        lo, hi, lop, hip_ = (gy, x, gp, xp)
        shape = (Cout, 4, 4, Cin)                       # physical [low-res channels][kh][kw][high-res channels] in both cases
        fn = hipb.deconv2d_bwd_weight if full else hipb.conv2d_bwd_weight
        args = (lo, hi) if full else (hi, lo)           # (input, gradOutput)
        pargs = (lop, hip_) if full else (hip_, lop)
        logical = (lambda t: t.permute(0, 3, 1, 2))
        for beta in (0.0, 1.0):
            want = logical(_rand(shape, 7, dev, 0.5))
            got = logical(want.permute(0, 2, 3, 1).clone())
            nb = Cin if full else Cout
            gb_w, gb_g = _rand((nb,), 8, dev), _rand((nb,), 8, dev)
            fn(*args, want, gb_w, 4, 2, 1, beta)
            if grouped:
                hipb.wgrad_group_begin()
            fn(*args, got, gb_g, 4, 2, 1, beta, *pargs)
            if grouped:
                hipb.wgrad_group_end()
            scale = float(want.abs().max())
            assert float((got - want).abs().max()) <= 2e-6 * scale, (full, beta)
            assert torch.equal(gb_w, gb_g)
    if Bn * H * H * Cin <= 1 << 21:
        xd, gd = x.double().cpu().contiguous().requires_grad_(False), gy.double().cpu().contiguous()
        wz = torch.zeros(Cout, Cin, 4, 4, dtype=torch.float64, requires_grad=True)
        torch.nn.functional.conv2d(xd, wz, stride=2, padding=1).backward(gd)
        got = hipb.zeros(Cout, 4, 4, Cin).permute(0, 3, 1, 2)
        hipb.conv2d_bwd_weight(x, gy, got, None, 4, 2, 1, 0.0, xp, gp)
        assert float((got.cpu().double() - wz.grad).abs().max() / wz.grad.abs().max()) <= 2e-5


# ------------------------------------------------------------------------------------------------ against the ORACLE
# (the comparisons above are HIP against HIP.)  The same passes against the CPU oracle's im2col + GEMM
# restatement of THNN (oracle.SpatialConvolution / SpatialFullConvolution), at the bar of tests/test_gpu_ops.py: 2e-5 of the
# tensor's max-norm.  Sizes the oracle finishes in seconds with 16 threads.
ORACLE_CASES = [(4, 64, 32, 128), (8, 64, 64, 64), (4, 128, 16, 256), (16, 256, 8, 512), (2, 192, 32, 384), (6, 32, 32, 96),
                (16, 64, 32, 64)]


def _to_dev(a, dev):
    t = torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    return t.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2) if t.dim() == 4 else t


def _np(t):
    return t.detach().cpu().contiguous().numpy()


def _err(got, want):
    return float(np.abs(_np(got).astype(np.float64) - want).max() / (np.abs(want).max() + 1e-30))


@pytest.mark.parametrize("Bn,Cin,H,Cout", ORACLE_CASES, ids=lambda v: str(v))
def test_planes_conv_passes_against_the_oracle(Bn, Cin, H, Cout, oracle, hipb):
    """nn.SpatialConvolution Cin -> Cout, 4x4 stride 2 pad 1, every pass from pre-split planes: forward (+ bias + LeakyReLU)
    on vf_pconv_gather, data-gradient on vf_pconv_scatter, weight gradient on vf_conv2d_bwd_weight_planes (k_pwgrad_group where
    the shape allows, the fp32-operand kernel otherwise) — against oracle.SpatialConvolution (train.lua:89-101)."""
    dev = hipb.device
    rng = np.random.default_rng(Bn * 1000 + Cin + Cout)
    oracle.set_num_threads(16)
    try:
        ref = oracle.SpatialConvolution(Cin, Cout, 4, 4, 2, 2, 1, 1)
        ref.weight[...] = rng.standard_normal(ref.weight.shape).astype(np.float32) * 0.05
        ref.bias[...] = rng.standard_normal(Cout).astype(np.float32) * 0.1
        x = rng.standard_normal((Bn, Cin, H, H)).astype(np.float32)
        y = ref.forward(x).copy()
        gy = rng.standard_normal(y.shape).astype(np.float32)
        ref.gradWeight[...] = 0
        ref.gradBias[...] = 0
        ref.backward(x, gy)
    finally:
        oracle.set_num_threads(1)
    dx, dw, db, dgy = _to_dev(x, dev), _to_dev(ref.weight, dev), _to_dev(ref.bias, dev), _to_dev(gy, dev)
    xp, gp = hipb.planes_split(dx), hipb.planes_split(dgy)
    wn, wt = hipb.weight_planes(dw)
    assert hipb.pconv_supported(Bn, H, H, Cin, Cout, 4, 2, 1, False)
    got = hipb.empty_act(Bn, Cout, H // 2, H // 2)
    hipb.pconv_gather(xp, wn, db, got, Bn, H, H, Cin, Cout, "lrelu", 0.2)
    assert _err(got, np.where(y > 0, y, 0.2 * y)) <= 2e-5
    if hipb.pconv_supported(Bn, H // 2, H // 2, Cout, Cin, 4, 2, 1, True):
        gx = hipb.empty_act(Bn, Cin, H, H)
        hipb.pconv_scatter(gp, wt, None, gx, Bn, H // 2, H // 2, Cout, Cin)
        assert _err(gx, ref.gradInput) <= 2e-5
    gw = torch.zeros_like(dw)
    gb = hipb.zeros(Cout)
    hipb.conv2d_bwd_weight(dx, dgy, gw, gb, 4, 2, 1, 0.0, xp, gp)
    assert _err(gw, ref.gradWeight) <= 2e-5
    assert _err(gb, ref.gradBias) <= 2e-5
    hipb.wgrad_group_begin()                                   # and recorded into a group, accumulating: exactly twice
    hipb.conv2d_bwd_weight(dx, dgy, gw, gb, 4, 2, 1, 1.0, xp, gp)
    hipb.wgrad_group_end()
    assert _err(gw, 2 * ref.gradWeight) <= 2e-5


@pytest.mark.parametrize("Bn,Cin,H,Cout", ORACLE_CASES, ids=lambda v: str(v))
def test_planes_full_conv_passes_against_the_oracle(Bn, Cin, H, Cout, oracle, hipb):
    """nn.SpatialFullConvolution Cout -> Cin (the decoder direction over the same pair of maps: H/2 -> H), every pass from
    planes: forward (+ bias + ReLU) on vf_pconv_scatter, data-gradient on vf_pconv_gather, weight gradient on
    vf_deconv2d_bwd_weight_planes — against oracle.SpatialFullConvolution (train.lua:134-146)."""
    dev = hipb.device
    rng = np.random.default_rng(Bn * 1000 + Cin + Cout + 1)
    Hl = H // 2
    if not hipb.pconv_supported(Bn, Hl, Hl, Cout, Cin, 4, 2, 1, True):
        pytest.skip("the low-resolution operand's channel count is not a multiple of 32")
    oracle.set_num_threads(16)
    try:
        ref = oracle.SpatialFullConvolution(Cout, Cin, 4, 4, 2, 2, 1, 1)
        ref.weight[...] = rng.standard_normal(ref.weight.shape).astype(np.float32) * 0.05
        ref.bias[...] = rng.standard_normal(Cin).astype(np.float32) * 0.1
        x = rng.standard_normal((Bn, Cout, Hl, Hl)).astype(np.float32)
        y = ref.forward(x).copy()
        gy = rng.standard_normal(y.shape).astype(np.float32)
        ref.gradWeight[...] = 0
        ref.gradBias[...] = 0
        ref.backward(x, gy)
    finally:
        oracle.set_num_threads(1)
    dx, dw, db, dgy = _to_dev(x, dev), _to_dev(ref.weight, dev), _to_dev(ref.bias, dev), _to_dev(gy, dev)
    xp, gp = hipb.planes_split(dx), hipb.planes_split(dgy)
    wn, wt = hipb.weight_planes(dw)                            # physical [Cout][16][Cin]; transposed [Cin][16][Cout]
    got = hipb.empty_act(Bn, Cin, H, H)
    hipb.pconv_scatter(xp, wt, db, got, Bn, Hl, Hl, Cout, Cin, "relu", 0.0)
    assert _err(got, np.maximum(y, 0)) <= 2e-5
    gx = hipb.empty_act(Bn, Cout, Hl, Hl)
    hipb.pconv_gather(gp, wn, None, gx, Bn, H, H, Cin, Cout)
    assert _err(gx, ref.gradInput) <= 2e-5
    gw = torch.zeros_like(dw)
    gb = hipb.zeros(Cin)
    hipb.deconv2d_bwd_weight(dx, dgy, gw, gb, 4, 2, 1, 0.0, xp, gp)
    assert _err(gw, ref.gradWeight) <= 2e-5
    assert _err(gb, ref.gradBias) <= 2e-5


#  which kernel a transposed pass lands on is a tiling decision of launch_pconv (vf_pgemm.hip): the patch kernel
# with four parity classes per block from 512 (row tile x column slice) blocks, with two below.  The ORACLE_CASES above reach only the
# two-class form and do not say so; here every case names the symbol it is meant for and the launch list is asserted.
PATCH_ORACLE_CASES = [
    (64, 64, 64, 64, "pconv_patch_128x64_t4_c4"),      # E2's data-gradient at batchSize 64 (train.lua:92): 512 row tiles x 1 slice
    (32, 128, 64, 64, "pconv_patch_128x64_t4_c4"),     # into N = 128 channels: 256 row tiles x 2 column slices
    (16, 64, 64, 64, "pconv_patch_128x64_t4_c2"),      # 128 row tiles: two classes per block
    (8, 128, 32, 64, "pconv_patch_128x64_t4_c2"),      # N = 128 on a 16 x 16 low-res grid
]


@pytest.mark.parametrize("Bn,Cin,H,Cout,symbol", PATCH_ORACLE_CASES, ids=lambda v: str(v))
def test_patch_fed_transposed_passes_against_the_oracle(Bn, Cin, H, Cout, symbol, oracle, hipb):
    """k_pconv_patch_tr<4> / <2> directly against oracle.SpatialConvolution.updateGradInput (THNN SpatialConvolutionMM's
    data-gradient, train.lua:92 for the first case) — plain, and with the LeakyReLU derivative mask of the layer below in the
    epilogue — asserting through vf_prof_begin / _end that the named symbol is what served the pass."""
    dev = hipb.device
    rng = np.random.default_rng(Bn * 1000 + Cin + Cout + 7)
    Hl = H // 2
    oracle.set_num_threads(16)
    try:
        ref = oracle.SpatialConvolution(Cin, Cout, 4, 4, 2, 2, 1, 1)
        ref.weight[...] = rng.standard_normal(ref.weight.shape).astype(np.float32) * 0.05
        gy = rng.standard_normal((Bn, Cout, Hl, Hl)).astype(np.float32)
        x = rng.standard_normal((Bn, Cin, H, H)).astype(np.float32)      # updateGradInput reads it for its shape only
        want = ref.updateGradInput(x, gy).copy()
    finally:
        oracle.set_num_threads(1)
    dw, dgy, dx = _to_dev(ref.weight, dev), _to_dev(gy, dev), _to_dev(x, dev)
    gp = hipb.planes_split(dgy)
    _, wt = hipb.weight_planes(dw)
    got = hipb.empty_act(Bn, Cin, H, H)
    hipb.prof_begin()
    hipb.pconv_scatter(gp, wt, None, got, Bn, Hl, Hl, Cout, Cin)
    names = hipb.prof_end()
    assert symbol in names, (symbol, list(names))
    assert _err(got, want) <= 2e-5
    # the derivative mask of the LeakyReLU whose output is this conv's input (conv -> LeakyReLU -> this conv): dx *= (x > 0 ? 1 : 0.2)
    hipb.prof_begin()
    hipb.pconv_scatter(gp, wt, None, got, Bn, Hl, Hl, Cout, Cin, dmask=dx, dact="lrelu", dslope=0.2)
    names = hipb.prof_end()
    assert symbol in names, (symbol, list(names))
    assert _err(got, want * np.where(x > 0, 1.0, 0.2)) <= 2e-5


# ---- the gather passes from a patch (k_pconv_patch_g): against k_pconv_dma BIT FOR BIT (same products, same K order
# per output element), and against the oracle with the symbol asserted.  Modes: m0 = 8 x 16 tiles of a map at least 8 x 16 (Wo = 16 and
# Wo = 32 take different tile walks), m1 = whole 8 x 8 maps, m2 = whole 4 x 4 maps; several cases split K over channel chunks.
GATHER_PATCH_CASES = [
    (8, 64, 32, 128, "m0"), (4, 64, 64, 64, "m0"), (16, 128, 16, 256, "m1"), (32, 256, 8, 512, "m2"), (64, 64, 32, 128, "m0"),
    (6, 192, 32, 384, "m0"), (64, 128, 16, 256, "m1"), (64, 256, 8, 512, "m2"), (2, 64, 128, 64, "m0"),
]


@pytest.mark.parametrize("Bn,Cin,H,Cout,mode", GATHER_PATCH_CASES, ids=lambda v: str(v))
def test_patch_fed_gather_pass_equals_the_tap_staged_kernel_bit_for_bit(Bn, Cin, H, Cout, mode, hipb):
    dev = hipb.device
    x = _act(Bn, Cin, H, 31, dev)
    w = _rand((Cout, 4, 4, Cin), 32, dev, 0.05).permute(0, 3, 1, 2)
    bias = _rand((Cout,), 33, dev, 0.1)
    xp = hipb.planes_split(x)
    wp, _ = hipb.weight_planes(w, want_transposed=False)
    outs = {}
    try:
        for route in (1, 0):           # 1: k_pconv_patch_g (the default), 0: k_pconv_dma
            hipb.pconv_set_routing(gather_patch=route)
            y = hipb.empty_act(Bn, Cout, H // 2, H // 2)
            y.fill_(float("nan"))
            hipb.prof_begin()
            hipb.pconv_gather(xp, wp, bias, y, Bn, H, H, Cin, Cout, "lrelu", 0.2)
            names = hipb.prof_end()
            if route:
                assert "pconv_patchg_128x64_t16_" + mode in names, list(names)
            else:
                assert not any(k.startswith("pconv_patchg") for k in names) and any(k.startswith("pconv_dma") for k in names), list(names)
            # the data-gradient form (nn.SpatialFullConvolution: no bias, the ReLU derivative of the layer below in the epilogue and the
            # BatchNorm-backward sums of that layer as a by-product)
            gx = hipb.empty_act(Bn, Cout, H // 2, H // 2)
            below = _act(Bn, Cout, H // 2, 34, dev)
            sm = _rand((Cout,), 35, dev, 0.1)
            rows = max(Bn * (H // 2) ** 2 // 64, 512) + 8
            part = hipb.zeros(rows * 2 * Cout, dtype=torch.float64)
            hipb.bn_fuse_next_bwd(below, below, "relu", 0.0, sm, part, 1)
            hipb.pconv_gather(xp, wp, None, gx, Bn, H, H, Cin, Cout)
            nrows = hipb.bn_fuse_result()
            # ... and the forward sums of a BatchNorm behind the convolution
            part2 = hipb.zeros(rows * 2 * Cout, dtype=torch.float64)
            y2 = hipb.empty_act(Bn, Cout, H // 2, H // 2)
            hipb.bn_fuse_next_fwd(sm, part2, 1)
            hipb.pconv_gather(xp, wp, bias, y2, Bn, H, H, Cin, Cout)
            nrows2 = hipb.bn_fuse_result()
            outs[route] = (y, gx, part.view(-1, 2 * Cout)[:max(nrows, 0)].clone(), nrows, y2, part2.view(-1, 2 * Cout)[:max(nrows2, 0)].clone(), nrows2)
    finally:
        hipb.pconv_set_routing(gather_patch=1)
    y0, g0, s0, n0, z0, f0, m0 = outs[0]
    for route in (1,):
        y1, g1, s1, n1, z1, f1, m1 = outs[route]
        assert torch.equal(y1, y0), (route, float((y1 - y0).abs().max()))
        assert torch.equal(g1, g0), (route, float((g1 - g0).abs().max()))
        assert torch.equal(z1, z0), route
        assert n1 == n0 and m1 == m0
        for a, b in ((s1, s0), (f1, f0)):
            if a.numel() == 0:
                continue
            if H // 2 >= 32:      # 8 x 16 tiles walk a 32-wide map in another order than 128 consecutive GEMM rows: the totals agree (fp32 partials)
                ta, tb = a.sum(0), b.sum(0)
                assert float((ta - tb).abs().max() / (tb.abs().max() + 1e-30)) <= 2e-6, route
            else:                 # same rows per tile, same order inside a wave's strip: the partial rows themselves, bit for bit
                assert torch.equal(a, b), route


# ... and the transposed passes whose low-resolution grid is a whole 8 x 8 / 4 x 4 map (k_pconv_patch_g<., ., TR>: E4 / E5 and netD's deeper
# data-gradients, D2 / D3 forward), against k_pconv_dma<., ., 4> bit for bit; (Bn, Cin, H, Cout) as the scatter tests above: the low-res
# operand has Cout channels on an H/2 grid
TR_PATCH_CASES = [(16, 128, 16, 256, "m1"), (32, 256, 8, 512, "m2"), (64, 128, 16, 256, "m1"), (64, 256, 8, 512, "m2"), (128, 256, 8, 512, "m2")]


@pytest.mark.parametrize("Bn,Cin,H,Cout,mode", TR_PATCH_CASES, ids=lambda v: str(v))
def test_patch_fed_transposed_whole_map_pass_equals_the_tap_staged_kernel_bit_for_bit(Bn, Cin, H, Cout, mode, hipb):
    dev = hipb.device
    Hl = H // 2
    gy = _act(Bn, Cout, Hl, 41, dev)
    w = _rand((Cout, 4, 4, Cin), 42, dev, 0.05).permute(0, 3, 1, 2)
    gp = hipb.planes_split(gy)
    _, wt = hipb.weight_planes(w)
    bias = _rand((Cin,), 43, dev, 0.1)
    below = _act(Bn, Cin, H, 44, dev)
    sm = _rand((Cin,), 45, dev, 0.1)
    rows = max(Bn * H * H // 64, 512) + 8
    outs = {}
    try:
        for route in (1, 0):
            hipb.pconv_set_routing(scatter_patch=route)
            res = []
            # conv data-gradient, plain; with the LeakyReLU derivative mask; with mask + BatchNorm-backward sums; full-conv forward with
            # bias + ReLU + BatchNorm forward sums
            for kind in ("plain", "mask", "bwd_sums", "fwd_sums"):
                out = hipb.empty_act(Bn, Cin, H, H)
                out.fill_(float("nan"))
                part = hipb.zeros(rows * 2 * Cin, dtype=torch.float64)
                hipb.prof_begin()
                if kind == "plain":
                    hipb.pconv_scatter(gp, wt, None, out, Bn, Hl, Hl, Cout, Cin)
                elif kind == "mask":
                    hipb.pconv_scatter(gp, wt, None, out, Bn, Hl, Hl, Cout, Cin, dmask=below, dact="lrelu", dslope=0.2)
                elif kind == "bwd_sums":
                    hipb.bn_fuse_next_bwd(below, below, "lrelu", 0.2, sm, part, 1)
                    hipb.pconv_scatter(gp, wt, None, out, Bn, Hl, Hl, Cout, Cin)
                else:
                    hipb.bn_fuse_next_fwd(sm, part, 1)
                    hipb.pconv_scatter(gp, wt, bias, out, Bn, Hl, Hl, Cout, Cin, "relu", 0.0)
                nrows = hipb.bn_fuse_result() if kind.endswith("sums") else 0
                names = hipb.prof_end()
                if route:
                    assert "pconv_patchg_128x64_t4_" + mode in names, (kind, list(names))
                else:
                    assert not any(k.startswith("pconv_patch") for k in names), (kind, list(names))
                res.append((out, part.view(-1, 2 * Cin)[:max(nrows, 0)].clone(), nrows))
            outs[route] = res
    finally:
        hipb.pconv_set_routing(scatter_patch=1)
    for (o1, p1, n1), (o0, p0, n0) in zip(outs[1], outs[0]):
        assert torch.equal(o1, o0), float((o1 - o0).abs().max())
        assert n1 == n0 and torch.equal(p1, p0)


GATHER_ORACLE_CASES = [(4, 64, 32, 128, "m0"), (2, 64, 64, 64, "m0"), (4, 128, 16, 256, "m1"), (16, 256, 8, 512, "m2")]


@pytest.mark.parametrize("Bn,Cin,H,Cout,mode", GATHER_ORACLE_CASES, ids=lambda v: str(v))
def test_patch_fed_gather_passes_against_the_oracle(Bn, Cin, H, Cout, mode, oracle, hipb):
    """k_pconv_patch_g directly against oracle.SpatialConvolution.updateOutput (+ LeakyReLU, train.lua:89-101) and against
    oracle.SpatialFullConvolution.updateGradInput over the same pair of maps (train.lua:134-146), the launch name asserted."""
    dev = hipb.device
    rng = np.random.default_rng(Bn * 1000 + Cin + Cout + 11)
    symbol = "pconv_patchg_128x64_t16_" + mode
    oracle.set_num_threads(16)
    try:
        ref = oracle.SpatialConvolution(Cin, Cout, 4, 4, 2, 2, 1, 1)
        ref.weight[...] = rng.standard_normal(ref.weight.shape).astype(np.float32) * 0.05
        ref.bias[...] = rng.standard_normal(Cout).astype(np.float32) * 0.1
        x = rng.standard_normal((Bn, Cin, H, H)).astype(np.float32)
        y = ref.forward(x).copy()
        full = oracle.SpatialFullConvolution(Cout, Cin, 4, 4, 2, 2, 1, 1)      # weight [Cout][Cin][4][4]: the same tensor read the other way
        full.weight[...] = ref.weight
        lo = rng.standard_normal((Bn, Cout, H // 2, H // 2)).astype(np.float32)
        want_gx = full.updateGradInput(lo, x).copy()                            # "gradOutput" = x: H x H maps with Cin channels
    finally:
        oracle.set_num_threads(1)
    dx, dw, db = _to_dev(x, dev), _to_dev(ref.weight, dev), _to_dev(ref.bias, dev)
    xp = hipb.planes_split(dx)
    wn, _ = hipb.weight_planes(dw, want_transposed=False)
    got = hipb.empty_act(Bn, Cout, H // 2, H // 2)
    hipb.prof_begin()
    hipb.pconv_gather(xp, wn, db, got, Bn, H, H, Cin, Cout, "lrelu", 0.2)
    names = hipb.prof_end()
    assert symbol in names, (symbol, list(names))
    assert _err(got, np.where(y > 0, y, 0.2 * y)) <= 2e-5
    hipb.prof_begin()
    hipb.pconv_gather(xp, wn, None, got, Bn, H, H, Cin, Cout)
    names = hipb.prof_end()
    assert symbol in names, (symbol, list(names))
    assert _err(got, want_gx) <= 2e-5


# ------------------------------------------------------------------------------------------------ producer-written planes
def _same_planes(a, b):
    return torch.equal(a.view(torch.int16), b.view(torch.int16))


@pytest.mark.parametrize("Bn,C,H,groups,act", [(4, 64, 32, 1, "lrelu"), (8, 128, 16, 2, "lrelu"), (6, 256, 8, 1, "relu"),
                                               (4, 64, 16, 2, "none"), (64, 64, 32, 1, "lrelu")])
def test_batchnorm_written_planes_are_the_split_of_its_output(Bn, C, H, groups, act, hipb):
    """The planes a BatchNorm writes beside its output (forward: y_planes; backward: gx_planes) for a planes-fed consumer must
    be BIT FOR BIT vf_planes_split of the fp32 tensor it wrote — otherwise the consumer convolves another tensor than the one
    the weight gradient, the next BatchNorm and the oracle comparison see.  Both forms: statistics pass included
    (vf_bn_train_fwd_planes / vf_bn_bwd_planes) and statistics from the neighbouring GEMM (vf_bn_train_fwd_pre / vf_bn_bwd_pre)."""
    from video_filler_amd import nn
    dev = hipb.device
    x = _act(Bn, C, H, 11, dev) * 1.3 + 0.2
    gy = _act(Bn, C, H, 12, dev)
    gamma, beta = _rand((C,), 13, dev, 0.1) + 1.0, _rand((C,), 14, dev, 0.1)
    rm, rv = hipb.zeros(C), hipb.zeros(C) + 1.0
    sm, si = hipb.zeros(groups * C), hipb.zeros(groups * C)
    sums = hipb.zeros(groups * 2 * C, dtype=torch.float64)
    y = hipb.empty_act(Bn, C, H, H)
    yp = torch.empty((3, y.numel()), dtype=torch.bfloat16, device=dev)
    hipb.bn_train_fwd_groups(x, y, gamma, beta, rm, rv, sm, si, sums, groups, 0.1, 1e-5, act, 0.2, y_planes=yp)
    assert _same_planes(yp, hipb.planes_split(y))
    gx = hipb.empty_act(Bn, C, H, H)
    gxp = torch.empty((3, gx.numel()), dtype=torch.bfloat16, device=dev)
    gg, gb = hipb.zeros(C), hipb.zeros(C)
    hipb.bn_bwd_groups(x, y if act != "none" else None, gy, gx, gg, gb, gamma, sm, si, sums, groups, act, 0.2, 0.0, gx_planes=gxp)
    assert _same_planes(gxp, hipb.planes_split(gx))
    # the same two through nn.Sequential with the statistics coming out of the GEMMs on either side (the _pre forms):
    # conv -> BN -> act -> conv, gate dropped so that the second conv is planes-fed and asks the BatchNorm for planes
    old = nn._PCONV_MIN_GFLOP
    nn._PCONV_MIN_GFLOP = 0.0
    try:
        A = (lambda: nn.LeakyReLU(0.2, True)) if act != "relu" else (lambda: nn.ReLU(True))
        net = nn.Sequential()
        net.add(nn.SpatialConvolution(C, C, 4, 4, 2, 2, 1, 1)).add(nn.SpatialBatchNormalization(C))
        if act != "none":
            net.add(A())
        net.add(nn.SpatialConvolution(C, C, 4, 4, 2, 2, 1, 1)).add(nn.SpatialBatchNormalization(C))
        if act != "none":
            net.add(A())
        net.add(nn.SpatialConvolution(C, 64, 4, 4, 2, 2, 1, 1))
        net.getParameters()
        gen = torch.Generator().manual_seed(3)
        for m in net.leaves():
            if isinstance(m, nn.SpatialConvolution):
                m.weight.copy_((torch.randn(m.weight.shape, generator=gen) * 0.05).to(dev))
        net.setBatchGroups(groups)
        xin = _act(Bn, C, 4 * H, 15, dev)
        out = net.forward(xin)
        bns = [m for m in net.leaves() if isinstance(m, nn.SpatialBatchNormalization)]
        seen = 0
        for m in bns:
            if m.output_planes is not None:
                assert _same_planes(m.output_planes, hipb.planes_split(m.output)), "forward planes of a BatchNorm (pre form)"
                seen += 1
        net.zeroGradParameters()
        net.backward(xin, _act(Bn, 64, out.shape[2], 16, dev))
        for m in bns:
            if m.grad_planes is not None:
                assert _same_planes(m.grad_planes, hipb.planes_split(m.gradInput)), "gradient planes of a BatchNorm (pre form)"
                seen += 1
        if Bn * H * H >= 1024:
            assert seen >= 2, "the planes hand-off did not happen: nothing was checked"
    finally:
        nn._PCONV_MIN_GFLOP = old


@pytest.mark.parametrize("Bn,H,Cout,act", [(4, 64, 64, "lrelu"), (8, 128, 64, "lrelu"), (2, 32, 128, "none"), (64, 64, 64, "lrelu")])
def test_thin_input_conv_written_planes_are_the_split_of_its_output(Bn, H, Cout, act, hipb):
    """vf_conv2d_fwd_planes (the 3-channel image-side layers, train.lua:89,183): planes from the epilogue == split of y, and y
    itself == vf_conv2d_fwd's, bit for bit."""
    dev = hipb.device
    x = _act(Bn, 3, H, 21, dev)
    w = _rand((Cout, 4, 4, 3), 22, dev, 0.05).permute(0, 3, 1, 2)
    b = _rand((Cout,), 23, dev, 0.1)
    y = hipb.empty_act(Bn, Cout, H // 2, H // 2)
    yp = torch.empty((3, y.numel()), dtype=torch.bfloat16, device=dev)
    hipb.conv2d_fwd_planes(x, w, b, y, yp, 4, 2, 1, act, 0.2)
    assert _same_planes(yp, hipb.planes_split(y))
    y2 = hipb.empty_act(Bn, Cout, H // 2, H // 2)
    hipb.conv2d_fwd(x, w, b, y2, 4, 2, 1, act, 0.2)
    assert torch.equal(y, y2)
