"""Pins for the CPU oracle (the reference has none — SURVEY 8(c), "parity unpinned"):
  (i)  agreement with PyTorch-CPU functional ops wherever PyTorch implements the same maths as Torch7
       (conv2d, conv_transpose2d, batch_norm incl. unbiased running_var, leaky_relu, mse_loss, l1_loss);
  (ii) independent numpy restatements + central finite differences in double for what PyTorch does differently
       (BCE eps = 1e-12, optim.adam's eps placement) or does not have (GDL's flattened pairing, MaskedMSE).
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

TOL = 2e-5


def _r(rng, *s):
    return rng.standard_normal(s).astype(np.float32)


def _close(a, b, tol=TOL):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    assert np.abs(a - b).max() <= tol * (np.abs(b).max() + 1e-30), np.abs(a - b).max() / (np.abs(b).max() + 1e-30)


@pytest.mark.parametrize("B,Cin,H,Cout,s,p", [(2, 3, 16, 8, 2, 1), (1, 12, 8, 5, 2, 1), (3, 8, 4, 10, 1, 0), (2, 16, 4, 1, 1, 0)])
def test_conv_matches_torch(B, Cin, H, Cout, s, p, oracle):
    rng = np.random.default_rng(Cin * 100 + Cout)
    m = oracle.SpatialConvolution(Cin, Cout, 4, 4, s, s, p, p)
    m.weight[...] = _r(rng, *m.weight.shape) * 0.1
    m.bias[...] = _r(rng, Cout)
    x = _r(rng, B, Cin, H, H)
    y = m.forward(x)
    xt = torch.from_numpy(x).requires_grad_()
    wt = torch.from_numpy(m.weight.copy()).requires_grad_()
    bt = torch.from_numpy(m.bias.copy()).requires_grad_()
    yt = F.conv2d(xt, wt, bt, stride=s, padding=p)
    _close(y, yt.detach().numpy())
    gy = _r(rng, *y.shape)
    yt.backward(torch.from_numpy(gy))
    m.backward(x, gy)
    _close(m.gradInput, xt.grad.numpy())
    _close(m.gradWeight, wt.grad.numpy())
    _close(m.gradBias, bt.grad.numpy())
    m.backward(x, gy)                      # accGradParameters accumulates (scale = 1)
    _close(m.gradWeight, 2 * wt.grad.numpy())


@pytest.mark.parametrize("B,Cin,H,Cout,s,p", [(2, 8, 4, 3, 2, 1), (3, 10, 1, 6, 1, 0), (1, 6, 8, 12, 2, 1)])
def test_fullconv_matches_torch(B, Cin, H, Cout, s, p, oracle):
    rng = np.random.default_rng(Cin * 10 + Cout)
    m = oracle.SpatialFullConvolution(Cin, Cout, 4, 4, s, s, p, p)
    m.weight[...] = _r(rng, *m.weight.shape) * 0.1
    m.bias[...] = _r(rng, Cout)
    x = _r(rng, B, Cin, H, H)
    y = m.forward(x)
    xt = torch.from_numpy(x).requires_grad_()
    wt = torch.from_numpy(m.weight.copy()).requires_grad_()
    bt = torch.from_numpy(m.bias.copy()).requires_grad_()
    yt = F.conv_transpose2d(xt, wt, bt, stride=s, padding=p)
    assert y.shape == tuple(yt.shape)
    _close(y, yt.detach().numpy())
    gy = _r(rng, *y.shape)
    yt.backward(torch.from_numpy(gy))
    m.backward(x, gy)
    _close(m.gradInput, xt.grad.numpy())
    _close(m.gradWeight, wt.grad.numpy())
    _close(m.gradBias, bt.grad.numpy())


@pytest.mark.parametrize("shape", [(4, 6, 5, 5), (8, 10, 1, 1), (2, 3, 16, 16)])
def test_batchnorm_matches_torch(shape, oracle):
    B, C, H, W = shape
    rng = np.random.default_rng(C)
    m = oracle.SpatialBatchNormalization(C)
    m.weight[...] = 1 + 0.1 * _r(rng, C)
    m.bias[...] = 0.1 * _r(rng, C)
    x = (_r(rng, *shape) * 2 + 0.5).astype(np.float32)
    rm, rv = torch.zeros(C), torch.ones(C)
    xt = torch.from_numpy(x).requires_grad_()
    g = torch.from_numpy(m.weight.copy()).requires_grad_()
    b = torch.from_numpy(m.bias.copy()).requires_grad_()
    for _ in range(2):                     # two updates of the running statistics
        y = m.forward(x).copy()
        yt = F.batch_norm(xt, rm, rv, g, b, True, 0.1, 1e-5)
    _close(y, yt.detach().numpy())
    _close(m.running_mean, rm.numpy())
    _close(m.running_var, rv.numpy())       # unbiased estimate in both
    gy = _r(rng, *shape)
    yt.backward(torch.from_numpy(gy))
    m.backward(x, gy)
    _close(m.gradInput, xt.grad.numpy(), 5e-5)
    _close(m.gradWeight, g.grad.numpy())
    _close(m.gradBias, b.grad.numpy())
    m.train = False
    _close(m.forward(x), F.batch_norm(torch.from_numpy(x), rm, rv, g.detach(), b.detach(), False, 0.1, 1e-5).numpy())


def test_activations_inplace_semantics(oracle):
    rng = np.random.default_rng(0)
    x = _r(rng, 2, 3, 4, 4)
    gy = _r(rng, 2, 3, 4, 4)
    a = oracle.LeakyReLU(0.2, True)
    buf = x.copy()
    y = a.forward(buf)
    assert y is buf                                                  # the producer's output was overwritten
    _close(y, F.leaky_relu(torch.from_numpy(x), 0.2).numpy(), 1e-7)
    g = gy.copy()
    gi = a.updateGradInput(buf, g)                                   # derivative from the ACTIVATED values
    assert gi is g
    _close(gi, np.where(x > 0, gy, 0.2 * gy), 1e-7)
    r = oracle.ReLU(True)
    buf = x.copy()
    r.forward(buf)
    _close(r.updateGradInput(buf, gy.copy()), np.where(x > 0, gy, 0), 1e-7)
    t = oracle.Tanh()
    yt = t.forward(x)
    _close(t.updateGradInput(x, gy), gy * (1 - yt * yt), 1e-6)
    s = oracle.Sigmoid()
    ys = s.forward(x)
    _close(ys, torch.sigmoid(torch.from_numpy(x)).numpy(), 1e-6)
    _close(s.updateGradInput(x, gy), gy * ys * (1 - ys), 1e-6)


def test_mse_and_abs_match_torch(oracle):
    rng = np.random.default_rng(1)
    x, t = _r(rng, 2, 3, 8, 8), _r(rng, 2, 3, 8, 8)
    c = oracle.MSECriterion()
    assert abs(c.forward(x, t) - F.mse_loss(torch.from_numpy(x), torch.from_numpy(t)).item()) < 1e-6
    xt = torch.from_numpy(x).requires_grad_()
    F.mse_loss(xt, torch.from_numpy(t)).backward()
    _close(c.backward(x, t), xt.grad.numpy(), 1e-6)
    l1 = oracle.lib().vfo_abs_fwd(oracle._p(x), oracle._p(t), x.size)
    assert abs(l1 - F.l1_loss(torch.from_numpy(x), torch.from_numpy(t)).item()) < 1e-6


def test_bce_eps_semantics_and_gradient(oracle):
    """nn.BCECriterion: -(1/N) sum t log(x+1e-12) + (1-t) log(1-x+1e-12) — finite at x in {0,1}, unlike a log clamp."""
    rng = np.random.default_rng(2)
    x = rng.random(40).astype(np.float32)
    x[:2] = (0.0, 1.0)
    c = oracle.BCECriterion()
    for label in (0.0, 1.0):
        t = np.full(40, label, np.float32)
        xd = x.astype(np.float64)
        want = -np.mean(t * np.log(xd + 1e-12) + (1 - t) * np.log(1 - xd + 1e-12))
        assert abs(c.forward(x, t) - want) < 1e-9 * max(1, abs(want))
        g = c.backward(x, t)
        want_g = -(1.0 / 40) * (t - xd) / ((1 - xd + 1e-12) * (xd + 1e-12))
        _close(g, want_g, 1e-6)
        # central finite difference in double on interior points
        h = 1e-6
        for i in (5, 17, 33):
            f = lambda v: -np.mean(t * np.log(np.where(np.arange(40) == i, v, xd) + 1e-12) + (1 - t) * np.log(1 - np.where(np.arange(40) == i, v, xd) + 1e-12))
            fd = (f(xd[i] + h) - f(xd[i] - h)) / (2 * h)
            assert abs(fd - g[i]) < 1e-4 * max(1, abs(fd))


def _gdl_numpy(yhat, y):
    """independent restatement of gdl_criterion.lua:12-30 with Torch7's flattened CSubTable pairing"""
    B, C, H, W = y.shape
    tot12 = tot34 = 0.0
    for X, sign in ((y, +1), (yhat, -1)):
        pass
    def crops(X):
        i1 = X[:, :, 0:H - 1, :].reshape(B, C, -1)
        j1 = X[:, :, 1:H, :].reshape(B, C, -1)
        i2 = X[:, :, :, 0:W - 1].reshape(B, C, -1)
        j2 = X[:, :, :, 1:W].reshape(B, C, -1)
        return i1, j1, i2, j2
    yi1, yj1, yi2, yj2 = crops(y.astype(np.float64))
    hi1, hj1, hi2, hj2 = crops(yhat.astype(np.float64))
    t12 = np.abs(yi2 - yi1) - np.abs(hi2 - hi1)
    t34 = np.abs(yj2 - yj1) - np.abs(hj2 - hj1)
    return np.abs(t12).mean() + np.abs(t34).mean()


def test_gdl_flattened_pairing(oracle):
    rng = np.random.default_rng(3)
    x, t = _r(rng, 2, 3, 8, 8), _r(rng, 2, 3, 8, 8)
    got = oracle.GDLCriterion(1).forward(x, t)
    assert abs(got - _gdl_numpy(x, t)) < 1e-6
    # it is NOT the image-gradient difference a reader might expect
    true_gdl = (np.abs(np.abs(np.diff(t, axis=3)) - np.abs(np.diff(x, axis=3))).mean()
                + np.abs(np.abs(np.diff(t, axis=2)) - np.abs(np.diff(x, axis=2))).mean())
    assert abs(got - true_gdl) > 1e-3
    assert np.isnan(oracle.lib().vfo_gdl_fwd(oracle._p(_r(rng, 1, 1, 4, 6)), oracle._p(_r(rng, 1, 1, 4, 6)), 1, 1, 4, 6))


def test_masked_mse(oracle):
    rng = np.random.default_rng(4)
    x, t = _r(rng, 2, 3, 6, 6), _r(rng, 2, 3, 6, 6)
    m = (rng.random(x.shape) > 0.5).astype(np.uint8)
    w = 0.05
    c = oracle.MaskedMSECriterion(w)
    c.setMask(m)
    wm = (1 - w) * m + w
    assert abs(c.forward(x, t) - np.mean(wm * (x.astype(np.float64) - t) ** 2)) < 1e-7
    _close(c.backward(x, t), 2.0 / x.size * wm * (x.astype(np.float64) - t), 1e-6)
    with pytest.raises(AssertionError):
        c.setMask(m.astype(np.float32))        # MaskedMSECriterion.lua:25 wants a ByteTensor


def test_adam_follows_optim_adam_not_torch(oracle):
    rng = np.random.default_rng(5)
    n = 1000
    x0, g = _r(rng, n), _r(rng, n) * 1e-4     # small gradients make the eps placement visible
    lr, b1, b2, eps = 0.002, 0.5, 0.999, 1e-8
    x = x0.copy()
    state = {"learningRate": lr, "beta1": b1}
    xt = torch.from_numpy(x0.copy()).requires_grad_()
    opt = torch.optim.Adam([xt], lr=lr, betas=(b1, b2), eps=eps)
    m = np.zeros(n)
    v = np.zeros(n)
    xd = x0.astype(np.float64)
    for t in range(1, 4):
        oracle.adam(lambda _x: (0.0, g), x, state)
        m = b1 * m + (1 - b1) * g
        v = b2 * v + (1 - b2) * g.astype(np.float64) ** 2
        step = lr * np.sqrt(1 - b2 ** t) / (1 - b1 ** t)
        xd = xd - step * m / (np.sqrt(v) + eps)            # optim/adam.lua: eps added to sqrt(v) BEFORE bias correction
        xt.grad = torch.from_numpy(g.copy())
        opt.step()
    np.testing.assert_allclose(x, xd, rtol=0, atol=2e-6)
    assert state["t"] == 3
    # torch.optim.Adam divides sqrt(v) by sqrt(1-b2^t) before adding eps: visibly different here
    assert np.abs(xt.detach().numpy() - x).max() > 10 * np.abs(x - xd).max()


def test_sequential_protocol_and_flat_parameters(oracle):
    rng = np.random.default_rng(6)
    net = oracle.build_netD(3, 8, False)
    oracle.weights_init(net, rng)
    flat, gflat = net.getParameters()
    leaves = []
    def walk(s):
        for m in s.modules:
            walk(m) if hasattr(m, "modules") else leaves.append(m)
    walk(net)
    # A.11: depth-first {weight, bias}; views alias the flat storage
    off = 0
    for m in leaves:
        if hasattr(m, "weight"):
            assert np.shares_memory(m.weight, flat) and np.shares_memory(m.gradWeight, gflat)
            np.testing.assert_array_equal(flat[off:off + m.weight.size], m.weight.ravel())
            off += m.weight.size + m.bias.size
    assert off == flat.size
    x = _r(rng, 2, 3, 64, 64)
    y = net.forward(x)
    assert y.shape == (2, 1)
    # in-place LeakyReLU: the conv's .output holds the activated values afterwards
    assert (leaves[0].output >= 0).mean() > 0.4 and np.array_equal(leaves[0].output, leaves[1].output)
    gy = _r(rng, 2, 1)
    net.backward(x, gy)
    g1 = gflat.copy()
    gi = net.updateGradInput(x, gy)            # data-grad only: parameters untouched (train.lua:371)
    np.testing.assert_array_equal(gflat, g1)
    assert gi.shape == x.shape


def test_gdl_backward_against_autograd_of_the_flattened_pairing(oracle):
    """gdl_criterion.lua:47-53 (never called by a driver; part of the nn.Criterion protocol).  The forward restated with
    torch ops — crops flattened and paired element by element, as CSubTable does on tensors of equal element count —
    and differentiated by autograd; random data keeps every |.| away from its kink, where THNN's +1 and torch's 0 differ."""
    rng = np.random.default_rng(12)
    B, Cc, H = 3, 4, 8
    x = rng.standard_normal((B, Cc, H, H)).astype(np.float32)
    t = rng.standard_normal((B, Cc, H, H)).astype(np.float32)
    xt = torch.from_numpy(x).double().requires_grad_(True)
    tt = torch.from_numpy(t).double()
    flat = lambda a: a.reshape(B, Cc, -1)
    i1 = lambda a: flat(a[:, :, :H - 1, :])
    j1 = lambda a: flat(a[:, :, 1:, :])
    i2 = lambda a: flat(a[:, :, :, :H - 1])
    j2 = lambda a: flat(a[:, :, :, 1:])
    t12 = (i2(tt) - i1(tt)).abs() - (i2(xt) - i1(xt)).abs()
    t34 = (j2(tt) - j1(tt)).abs() - (j2(xt) - j1(xt)).abs()
    loss = t12.abs().mean() + t34.abs().mean()
    loss.backward()
    crit = oracle.GDLCriterion(1)
    assert abs(crit.forward(x, t) - float(loss)) < 1e-6
    g = crit.backward(x, t)
    np.testing.assert_allclose(g, xt.grad.numpy(), rtol=1e-5, atol=1e-9)
    # the convention at the kinks: input == target makes every term12 / term34 exactly 0 -> AbsCriterion's +1 branch
    g0 = crit.backward(t, t)
    assert np.isfinite(g0).all() and np.abs(g0).max() <= 4.0 / (B * Cc * (H - 1) * H) + 1e-12
    # (the first W-1 pairings of the i kind pair an element with itself — i2[k] and i1[k] are both X[0][k] — so their d is
    #  exactly 0 in every input; the +g and -g they hand out land on the same element and cancel, whatever the sign choice)
