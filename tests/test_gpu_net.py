"""vf_net_* — nn.Sequential behind the C-ABI (include/vf_hip.h, csrc/vf_net.hip) — against the Python mirror of the same protocol
(video-filler_amd/nn.py) on discriminator- and generator-shaped stacks of the reference's layers (train.lua:87-199): forward,
backward (gradInput + every parameter gradient), updateGradInput, gradient accumulation and zeroGradParameters, evaluate mode."""
import ctypes as C

import numpy as np
import pytest
import torch

from helpers import rel_err

pytestmark = pytest.mark.gpu

CONV, FULL, BN, ACT, VIEW = 1, 2, 3, 4, 5
LRELU, RELU, TANH, SIGMOID = 1, 2, 3, 4


class Desc(C.Structure):
    _fields_ = [("kind", C.c_int), ("nin", C.c_int), ("nout", C.c_int), ("k", C.c_int), ("stride", C.c_int), ("pad", C.c_int),
                ("act", C.c_int), ("slope", C.c_float), ("eps", C.c_float), ("momentum", C.c_float)]


def _stack(kind):
    from video_filler_amd import nn
    if kind == "netD":      # train.lua:183-199 at quarter width: conv+LReLU, 2 x (conv+BN+LReLU), 4x4 conv to 1x1, Sigmoid, View
        descs = [(CONV, 3, 16, 4, 2, 1), (ACT, LRELU, 0.2), (CONV, 16, 32, 4, 2, 1), (BN, 32), (ACT, LRELU, 0.2),
                 (CONV, 32, 64, 4, 2, 1), (BN, 64), (ACT, LRELU, 0.2), (CONV, 64, 1, 4, 1, 0), (ACT, SIGMOID, 0.0), (VIEW,)]
        shape = (4, 3, 32, 32)
    else:                   # the decoder end of netG (train.lua:134-146): full-conv from 1x1, 2 x (full-conv+BN+ReLU), full-conv, Tanh
        descs = [(BN, 32), (ACT, LRELU, 0.2), (FULL, 32, 64, 4, 1, 0), (BN, 64), (ACT, RELU, 0.0), (FULL, 64, 32, 4, 2, 1), (BN, 32),
                 (ACT, RELU, 0.0), (FULL, 32, 3, 4, 2, 1), (ACT, TANH, 0.0)]
        shape = (4, 32, 1, 1)
    arr = (Desc * len(descs))()
    seq = nn.Sequential(True, True)
    for i, d in enumerate(descs):
        if d[0] in (CONV, FULL):
            arr[i] = Desc(d[0], d[1], d[2], d[3], d[4], d[5], 0, 0.0, 0.0, 0.0)
            cls = nn.SpatialConvolution if d[0] == CONV else nn.SpatialFullConvolution
            seq.add(cls(d[1], d[2], d[3], d[3], d[4], d[4], d[5], d[5]))
        elif d[0] == BN:
            arr[i] = Desc(BN, 0, d[1], 0, 0, 0, 0, 0.0, 0.0, 0.0)
            seq.add(nn.SpatialBatchNormalization(d[1]))
        elif d[0] == ACT:
            arr[i] = Desc(ACT, 0, 0, 0, 0, 0, d[1], d[2], 0.0, 0.0)
            seq.add({LRELU: lambda: nn.LeakyReLU(d[2], True), RELU: lambda: nn.ReLU(True), TANH: nn.Tanh, SIGMOID: nn.Sigmoid}[d[1]]())
        else:
            arr[i] = Desc(VIEW, 0, 0, 0, 0, 0, 0, 0.0, 0.0, 0.0)
            seq.add(nn.View(1).setNumInputDims(3))
    return arr, seq, shape


@pytest.mark.parametrize("kind", ["netD", "netG_decoder"])
def test_net_object_matches_the_module_mirror(kind, hipb):
    lib, ctx, dev = hipb.lib, hipb.ctx, hipb.device
    arr, seq, shape = _stack(kind)
    Bn, Cc, H, W = shape
    net = C.c_void_p()
    assert lib.vf_net_create(ctx, C.byref(net), arr, len(arr), Bn, Cc, H, W) == 0, lib.vf_last_error()
    p, g, cnt = C.c_void_p(), C.c_void_p(), C.c_int64()
    assert lib.vf_net_parameters(net, C.byref(p), C.byref(g), C.byref(cnt)) == 0

    def d2d(dst_ptr, src_tensor):
        assert lib.vf_memcpy_h2d(ctx, C.c_void_p(dst_ptr), C.c_void_p(src_tensor.data_ptr()), src_tensor.numel() * 4) == 0

    def read(ptr, n):
        out = torch.empty(n, dtype=torch.float32)
        assert lib.vf_memcpy_d2h(ctx, C.c_void_p(out.data_ptr()), C.c_void_p(ptr), n * 4) == 0
        return out

    # the mirror's parameters (random), copied module by module into the net object's flat buffer
    seq.getParameters()
    gen = torch.Generator().manual_seed(3)
    mods = seq.leaves()
    for i, m in enumerate(mods):
        if not m.parameters():
            continue
        for which, t in enumerate(m.parameters()[0]):
            v = (torch.randn(t.shape, generator=gen) * (0.1 if which == 0 and t.dim() == 4 else 0.5) + (1.0 if which == 0 and t.dim() == 1 else 0.0))
            t.copy_(v.to(dev))
            ln = C.c_int64()
            off = lib.vf_net_param_offset(net, i, which, C.byref(ln))
            assert off >= 0
            phys = t.permute(0, 2, 3, 1).contiguous() if t.dim() == 4 else t.contiguous()      # channels-last storage order
            assert ln.value == phys.numel()
            host = phys.cpu().contiguous()
            d2d(p.value + 4 * off, host)
    x = torch.randn(Bn, H, W, Cc, generator=gen).to(dev).permute(0, 3, 1, 2)
    y_ref = seq.forward(x)
    yp = C.c_void_p()
    assert lib.vf_net_forward(net, C.c_void_p(x.data_ptr()), C.byref(yp)) == 0, lib.vf_last_error()
    y = read(yp.value, y_ref.numel())
    y_ref_phys = (y_ref.permute(0, 2, 3, 1) if y_ref.dim() == 4 else y_ref).contiguous().cpu().reshape(-1)
    assert rel_err(y.numpy(), y_ref_phys.numpy()) < 1e-5
    # backward twice (accumulation), against the mirror doing the same
    gy = torch.randn(y_ref_phys.shape, generator=gen).to(dev)
    gy_log = gy.view(y_ref.permute(0, 2, 3, 1).shape).permute(0, 3, 1, 2) if y_ref.dim() == 4 else gy.view(y_ref.shape)
    seq.zeroGradParameters()
    assert lib.vf_net_zero_grad(net) == 0
    gxp = C.c_void_p()
    for _ in range(2):
        gx_ref = seq.backward(x, gy_log)
        assert lib.vf_net_backward(net, C.c_void_p(x.data_ptr()), C.c_void_p(gy.data_ptr()), C.byref(gxp)) == 0, lib.vf_last_error()
    gx = read(gxp.value, x.numel())
    assert rel_err(gx.numpy(), gx_ref.permute(0, 2, 3, 1).contiguous().cpu().reshape(-1).numpy()) < 2e-5
    for i, m in enumerate(mods):
        if not m.parameters():
            continue
        # a module's tensors share one scale: the bias of a convolution in front of a BatchNorm has a TRUE gradient of exactly 0
        # (what both sides hold there is rounding noise, 1e-6 of the weight gradient)
        scale = max(float(t.abs().max()) for t in m.parameters()[1])
        for which, t in enumerate(m.parameters()[1]):
            ln = C.c_int64()
            off = lib.vf_net_param_offset(net, i, which, C.byref(ln))
            got = read(g.value + 4 * off, ln.value)
            want = (t.permute(0, 2, 3, 1) if t.dim() == 4 else t).contiguous().cpu().reshape(-1)
            assert float((got - want).abs().max()) <= 5e-5 * scale, (i, which)
    # updateGradInput: the same gradInput, parameter gradients untouched
    before = read(g.value, cnt.value)
    gx2p = C.c_void_p()
    assert lib.vf_net_update_grad_input(net, C.c_void_p(x.data_ptr()), C.c_void_p(gy.data_ptr()), C.byref(gx2p)) == 0
    assert torch.equal(read(gx2p.value, x.numel()), gx)
    assert torch.equal(read(g.value, cnt.value), before)
    # evaluate mode: running statistics (both sides saw two... one forward; same momentum update)
    seq.evaluate()
    assert lib.vf_net_training(net, 0) == 0
    y_ref = seq.forward(x)
    assert lib.vf_net_forward(net, C.c_void_p(x.data_ptr()), C.byref(yp)) == 0
    y = read(yp.value, y_ref.numel())
    assert rel_err(y.numpy(), (y_ref.permute(0, 2, 3, 1) if y_ref.dim() == 4 else y_ref).contiguous().cpu().reshape(-1).numpy()) < 1e-5
    # a shape mismatch is an error with a message
    bad = (Desc * 1)(Desc(CONV, 5, 8, 4, 2, 1, 0, 0.0, 0.0, 0.0))
    n2 = C.c_void_p()
    assert lib.vf_net_create(ctx, C.byref(n2), bad, 1, 2, 3, 8, 8) != 0 and b"input planes" in lib.vf_last_error()
    assert lib.vf_net_destroy(net) == 0
