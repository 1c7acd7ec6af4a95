"""Full-size (BASELINE.json configs[1]: batchSize 64, nBottleneck 4000) checks through size-independent properties —
the oracle would need minutes per layer at these sizes, the properties need none:

  * adjointness:  <gy, conv(x; W)> = <conv_bwd_data(gy; W), x>          (forward and data-grad kernels are transposes)
  * bilinearity:  <gy, conv(x; W')> = <conv_bwd_weight(x, gy), W'>      (weight-grad kernel is the W-derivative)
  * linearity:    conv(a*x1 + b*x2) = a*conv(x1) + b*conv(x2)
  * BatchNorm:    per-channel mean 0 / variance 1 of the normalised output; sum_p gx = 0 and sum_p gx*xhat = 0
  * determinism:  two runs of the whole iteration from the same state are bitwise equal (no float atomics anywhere
                  on the value path)
Tolerance: 2e-5 relative on the inner products (fp32 sums of up to 1.7e7 terms, accumulated in fp64 by torch).
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

B = 64
# every conv / full-conv of the train.lua nets at fineSize 128 (train.lua:87-199): (kind, Cin, H, Cout, stride, pad)
LAYERS = [("conv", 3, 128, 64, 2, 1), ("conv", 64, 64, 64, 2, 1), ("conv", 64, 32, 128, 2, 1), ("conv", 128, 16, 256, 2, 1),
          ("conv", 256, 8, 512, 2, 1), ("conv", 512, 4, 4000, 1, 0), ("full", 4000, 1, 512, 1, 0), ("full", 512, 4, 256, 2, 1),
          ("full", 256, 8, 128, 2, 1), ("full", 128, 16, 64, 2, 1), ("full", 64, 32, 3, 2, 1),
          ("conv", 3, 64, 64, 2, 1), ("conv", 512, 4, 1, 1, 0)]


def _dot(a, b):
    return float((a.double() * b.double()).sum().item())


def _close(a, b, tol=2e-5):
    assert abs(a - b) <= tol * max(abs(a), abs(b), 1e-30), (a, b, abs(a - b) / max(abs(a), abs(b)))


@pytest.mark.parametrize("layer", LAYERS, ids=lambda l: "%s%d-%d@%d" % (l[0], l[1], l[3], l[2]))
def test_conv_adjoint_bilinear_linear(layer, hipb):
    kind, Cin, H, Cout, s, p = layer
    full = kind == "full"
    Ho = (H - 1) * s - 2 * p + 4 if full else (H + 2 * p - 4) // s + 1
    g = torch.Generator(device="cpu").manual_seed(Cin * 131 + Cout)
    rnd = lambda *sh: torch.randn(*sh, generator=g).to(hipb.device)
    x = rnd(B, H, H, Cin).permute(0, 3, 1, 2)
    x2 = rnd(B, H, H, Cin).permute(0, 3, 1, 2)
    gy = rnd(B, Ho, Ho, Cout).permute(0, 3, 1, 2)
    d0, d1 = (Cin, Cout) if full else (Cout, Cin)
    w = (rnd(d0, 4, 4, d1) * 0.05).permute(0, 3, 1, 2)
    w2 = (rnd(d0, 4, 4, d1) * 0.05).permute(0, 3, 1, 2)
    fwd = hipb.deconv2d_fwd if full else hipb.conv2d_fwd
    bwd_d = hipb.deconv2d_bwd_data if full else hipb.conv2d_bwd_data
    bwd_w = hipb.deconv2d_bwd_weight if full else hipb.conv2d_bwd_weight
    y = hipb.empty_act(B, Cout, Ho, Ho)
    fwd(x, w, None, y, 4, s, p)
    gx = hipb.empty_act(B, Cin, H, H)
    bwd_d(gy, w, gx, 4, s, p)
    _close(_dot(gy, y), _dot(gx, x))                                    # adjointness
    gw = torch.zeros_like(w)
    bwd_w(x, gy, gw, None, 4, s, p, 0.0)
    y2 = hipb.empty_act(B, Cout, Ho, Ho)
    fwd(x, w2, None, y2, 4, s, p)
    _close(_dot(gy, y2), _dot(gw, w2))                                  # bilinearity in W
    a, b = 0.75, -1.25
    xm = (a * x + b * x2).contiguous(memory_format=torch.channels_last) if x.dim() == 4 else a * x + b * x2
    xm = xm.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
    ym = hipb.empty_act(B, Cout, Ho, Ho)
    fwd(xm, w, None, ym, 4, s, p)
    fwd(x2, w, None, y2, 4, s, p)
    lin = a * y + b * y2
    err = float((ym - lin).abs().max().item()) / (float(lin.abs().max().item()) + 1e-30)
    assert err <= 2e-5, err                                             # linearity in x
    # accumulate form: beta = 1 adds exactly one more copy
    bwd_w(x, gy, gw, None, 4, s, p, 1.0)
    _close(_dot(gy, y2 * 0 + y), 0.5 * _dot(gw, w), 4e-5)


@pytest.mark.parametrize("C,HW", [(64, 32), (128, 16), (256, 8), (512, 4), (4000, 1)])
def test_batchnorm_invariants_full_size(C, HW, hipb):
    g = torch.Generator(device="cpu").manual_seed(C)
    x = (torch.randn(B, HW, HW, C, generator=g) * 1.7 + 0.6).to(hipb.device).permute(0, 3, 1, 2)
    gy = torch.randn(B, HW, HW, C, generator=g).to(hipb.device).permute(0, 3, 1, 2)
    gamma, beta = hipb.zeros(C) + 1.0, hipb.zeros(C)
    rm, rv, sm, si = hipb.zeros(C), hipb.zeros(C) + 1, hipb.zeros(C), hipb.zeros(C)
    sums = hipb.zeros(2 * C, dtype=torch.float64)
    y = hipb.empty_act(B, C, HW, HW)
    hipb.bn_train_fwd(x, y, gamma, beta, rm, rv, sm, si, sums, 0.1, 1e-5)
    n = B * HW * HW
    yd = y.double()
    mean = yd.mean(dim=(0, 2, 3))
    var = (yd * yd).mean(dim=(0, 2, 3)) - mean * mean
    assert float(mean.abs().max()) < 2e-5
    xv = x.double().var(dim=(0, 2, 3), unbiased=False)
    assert float((var - xv / (xv + 1e-5)).abs().max()) < 1e-4        # var = sigma^2/(sigma^2+eps)
    gx = hipb.empty_act(B, C, HW, HW)
    gg, gb = hipb.zeros(C), hipb.zeros(C)
    hipb.bn_bwd(x, None, gy, gx, gg, gb, gamma, sm, si, sums, "none", 0.0, 0.0)
    scale = float(gy.abs().max())
    assert float(gx.double().sum(dim=(0, 2, 3)).abs().max()) / n < 1e-6 * scale      # sum_p gx = 0
    assert float((gx.double() * yd).sum(dim=(0, 2, 3)).abs().max()) / n < 1e-5 * scale   # sum_p gx*xhat = 0
    assert float((gb.double() - gy.double().sum(dim=(0, 2, 3))).abs().max()) < 1e-3 * (n ** 0.5)


def test_full_config_iteration_is_deterministic_and_finite(hipb):
    """configs[1] itself: batchSize 64, nBottleneck 4000, wtl2 0.999, overlapPred 4; three iterations, twice."""
    from video_filler_amd.trainers import CenterTrainer
    opt = dict(batchSize=B, nBottleneck=4000, wtl2=0.999, overlapPred=4)
    gen = torch.Generator().manual_seed(99)
    batch = torch.rand((B, 3, 128, 128), generator=gen) * 2 - 1
    runs = []
    for _ in range(2):
        tr = CenterTrainer(opt, seed=7)
        assert abs(tr.netG.n_parameters() - 71.13e6) < 0.01e6        # 71.13 M with nBottleneck = 4000
        assert tr.netD.n_parameters() == 2766529
        tr.set_batch(batch)
        for _ in range(3):
            tr.step()
        l = tr.losses()
        assert all(np.isfinite(v) for v in l.values() if v is not None)
        runs.append((l, tr.parametersG.clone(), tr.parametersD.clone()))
        del tr
    # parameters: bitwise (every reduction on the value path has a fixed order)
    assert torch.equal(runs[0][1], runs[1][1]) and torch.equal(runs[0][2], runs[1][2])
    # reported loss scalars are summed with one double atomic per block: equal to ~1e-15, not bitwise
    for k, v in runs[0][0].items():
        if v is not None:
            assert abs(v - runs[1][0][k]) <= 1e-12 * max(1.0, abs(v))
    l = runs[0][0]
    assert 0 < l["errG_l2"] < 1.0 and 0 < l["errD"] < 20 and 0 < l["errG"] < 40
