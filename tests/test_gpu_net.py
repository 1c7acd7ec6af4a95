"""vf_net_* — the C-ABI's own net object (include/vf_hip.h, csrc/vf_net.hip), driven through raw ctypes exactly as a foreign host
would (no Python mirror in between), against the CPU ORACLE's nn.Sequential on the reference's nets (train.lua:87-199,
train_vid_weighted.lua:112-236 at reduced width): forward, backward (gradInput + every parameter gradient), accumulation and
zeroGradParameters, updateGradInput, evaluate mode, the conv-bias sweep, netD's [real; fake] batch in two BatchNorm groups with
the generator's pass over the fake half, the cut backward walk of the data-parallel step, reshape, host-bound storage.

Tolerances: forward 2e-5 of the output's max-norm on the REAL nets; gradients 1e-4 of their max-norm on the SMOOTH nets
(LeakyReLU(1.0) everywhere — same graph, same kernels; on the real nets a pre-activation within rounding distance of its kink
moves small-batch gradients by ~1e-3, tests/test_gpu_trainers.py pins those through the trainers).  Both routings of the 4x4
stride-2 passes run: `planes` (gate dropped: k_pconv_dma / k_pwgrad_group fed by producer-written planes) and `shipped`."""
import ctypes as C

import numpy as np
import pytest
import torch

from helpers import rel_err

pytestmark = pytest.mark.gpu

CONV, FULL, BN, ACT, VIEW = 1, 2, 3, 4, 5
ACTS = {"LeakyReLU": 1, "ReLU": 2, "Tanh": 3, "Sigmoid": 4}


class Desc(C.Structure):
    _fields_ = [("kind", C.c_int), ("nin", C.c_int), ("nout", C.c_int), ("k", C.c_int), ("stride", C.c_int), ("pad", C.c_int),
                ("act", C.c_int), ("slope", C.c_float), ("eps", C.c_float), ("momentum", C.c_float)]


def _oleaves(m):
    if hasattr(m, "modules"):
        out = []
        for c in m.modules:
            out += _oleaves(c)
        return out
    return [m]


def _descs(onet):
    """the vf_layer_desc array of an ORACLE net (the same module list the reference builds)"""
    mods = _oleaves(onet)
    arr = (Desc * len(mods))()
    for i, m in enumerate(mods):
        t = type(m).__name__
        if t in ("SpatialConvolution", "SpatialFullConvolution"):
            arr[i] = Desc(FULL if t == "SpatialFullConvolution" else CONV, m.nInputPlane, m.nOutputPlane, m.kH, m.dH, m.padH, 0, 0.0, 0.0, 0.0)
        elif t == "SpatialBatchNormalization":
            arr[i] = Desc(BN, 0, m.weight.shape[0], 0, 0, 0, 0, 0.0, 0.0, 0.0)
        elif t in ACTS:
            arr[i] = Desc(ACT, 0, 0, 0, 0, 0, ACTS[t], float(getattr(m, "negval", 0.0)), 0.0, 0.0)
        else:
            assert t == "View", t
            arr[i] = Desc(VIEW, 0, 0, 0, 0, 0, 0, 0.0, 0.0, 0.0)
    return arr, mods


class Host:
    """a foreign host of vf_net_*: ctypes and device pointers only"""

    def __init__(self, hipb, onet, shape):
        self.lib, self.ctx, self.dev = hipb.lib, hipb.ctx, hipb.device
        self.arr, self.mods = _descs(onet)
        self.net = C.c_void_p()
        Bn, Cc, H, W = shape
        assert self.lib.vf_net_create(self.ctx, C.byref(self.net), self.arr, len(self.arr), Bn, Cc, H, W) == 0, self.lib.vf_last_error()
        self.shape = shape
        p, g, cnt = C.c_void_p(), C.c_void_p(), C.c_int64()
        assert self.lib.vf_net_parameters(self.net, C.byref(p), C.byref(g), C.byref(cnt)) == 0
        self.p, self.g, self.count = p.value, g.value, cnt.value

    def close(self):
        assert self.lib.vf_net_destroy(self.net) == 0

    def ok(self, rc):
        assert rc == 0, self.lib.vf_last_error()

    def upload(self, ptr, arr):
        t = torch.from_numpy(np.ascontiguousarray(arr, dtype=np.float32))
        self.ok(self.lib.vf_memcpy_h2d(self.ctx, C.c_void_p(ptr), C.c_void_p(t.data_ptr()), t.numel() * 4))
        self.ok(self.lib.vf_stream_synchronize(self.ctx))

    def download(self, ptr, n):
        out = torch.empty(n, dtype=torch.float32)
        self.ok(self.lib.vf_memcpy_d2h(self.ctx, C.c_void_p(out.data_ptr()), C.c_void_p(ptr), n * 4))
        return out.numpy()

    def load_from_oracle(self):
        """the oracle's parameters into the flat buffer, tensor by tensor (logical NCHW -> channels-last storage order)"""
        for i, m in enumerate(self.mods):
            if not hasattr(m, "weight"):
                continue
            for which, t in enumerate((m.weight, m.bias)):
                ln = C.c_int64()
                off = self.lib.vf_net_param_offset(self.net, i, which, C.byref(ln))
                assert off >= 0 and ln.value == t.size
                phys = t.transpose(0, 2, 3, 1) if t.ndim == 4 else t
                self.upload(self.p + 4 * off, phys)
            if hasattr(m, "running_mean"):
                rm, rv = C.c_void_p(), C.c_void_p()
                self.ok(self.lib.vf_net_bn_running(self.net, i, C.byref(rm), C.byref(rv)))
                self.upload(rm.value, m.running_mean)
                self.upload(rv.value, m.running_var)

    def grads(self):
        """parameter gradients in the oracle's order and layout: [(layer, which, array)]"""
        out = []
        for i, m in enumerate(self.mods):
            if not hasattr(m, "weight"):
                continue
            for which, t in enumerate((m.gradWeight, m.gradBias)):
                ln = C.c_int64()
                off = self.lib.vf_net_param_offset(self.net, i, which, C.byref(ln))
                a = self.download(self.g + 4 * off, ln.value)
                if t.ndim == 4:
                    d0, d1, kH, kW = t.shape
                    a = a.reshape(d0, kH, kW, d1).transpose(0, 3, 1, 2)
                out.append((i, which, a.reshape(t.shape), t))
        return out

    def dev_in(self, a):
        """NCHW numpy -> NHWC device tensor (kept alive by the caller)"""
        t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(self.dev)
        return t.permute(0, 2, 3, 1).contiguous() if t.dim() == 4 else t.contiguous()

    def read_act(self, ptr, shape_nchw):
        Bn, Cc, H, W = shape_nchw
        return self.download(ptr, Bn * Cc * H * W).reshape(Bn, H, W, Cc).transpose(0, 3, 1, 2)

    def forward(self, x_dev):
        yp = C.c_void_p()
        self.ok(self.lib.vf_net_forward(self.net, C.c_void_p(x_dev.data_ptr()), C.byref(yp)))
        return yp.value


def _nets(kind, oracle, smooth, seed=3):
    rng = np.random.default_rng(seed)
    if kind == "netD":        # train_vid_weighted.lua:213-236 at quarter width, 6-channel clips, 64x64 ... 128 needs 5 levels
        net = oracle.build_netD(6, 16, True, smooth)
        shape = (4, 6, 128, 128)
    elif kind == "netD64":    # train.lua:183-199 at quarter width
        net = oracle.build_netD(3, 16, False, smooth)
        shape = (4, 3, 64, 64)
    else:                     # train_vid_weighted.lua:112-176 at quarter width: encoder, 1x1 bottleneck, decoder to 128x128
        net = oracle.build_netG(6, 6, 16, 16, 48, True, smooth)
        shape = (4, 6, 128, 128)
    oracle.weights_init(net, rng)
    for m in _oleaves(net):     # biases / BatchNorm shifts away from zero so that they matter
        if hasattr(m, "bias"):
            m.bias[...] = 0.05 * rng.standard_normal(m.bias.shape).astype(np.float32)
    return net, shape, rng


@pytest.fixture(params=["planes", "shipped"])
def gate(request, hipb):
    lib = hipb.lib
    assert lib.vf_net_set_planes_gate(0.0 if request.param == "planes" else 3.0, 1024) == 0
    yield request.param
    assert lib.vf_net_set_planes_gate(3.0, 1024) == 0


@pytest.mark.parametrize("kind", ["netD", "netD64", "netG"])
def test_net_object_forward_backward_against_the_oracle(kind, gate, oracle, hipb):
    oracle.set_num_threads(16)
    try:
        # ---- forward on the REAL nets, training and evaluate mode
        onet, shape, rng = _nets(kind, oracle, smooth=False)
        h = Host(hipb, onet, shape)
        h.load_from_oracle()
        x = rng.uniform(-1, 1, shape).astype(np.float32)
        want = onet.forward(x.copy())
        xd = h.dev_in(x)
        yp = h.forward(xd)
        got = h.download(yp, want.size)
        w = want.transpose(0, 2, 3, 1) if want.ndim == 4 else want
        assert rel_err(got.reshape(w.shape), w) <= 2e-5, "training-mode forward"
        for i, m in enumerate(h.mods):      # running statistics moved exactly as the oracle's
            if hasattr(m, "running_mean"):
                rm, rv = C.c_void_p(), C.c_void_p()
                h.ok(h.lib.vf_net_bn_running(h.net, i, C.byref(rm), C.byref(rv)))
                scale = max(np.abs(m.running_mean).max(), np.sqrt(m.running_var).max())
                assert np.abs(h.download(rm.value, m.running_mean.size) - m.running_mean).max() <= 1e-5 * scale
                assert rel_err(h.download(rv.value, m.running_var.size), m.running_var) <= 1e-5
        onet.evaluate()
        h.ok(h.lib.vf_net_training(h.net, 0))
        want = onet.forward(x.copy())
        got = h.download(h.forward(xd), want.size)
        w = want.transpose(0, 2, 3, 1) if want.ndim == 4 else want
        assert rel_err(got.reshape(w.shape), w) <= 5e-5, "evaluate-mode forward"
        h.close()

        # ---- backward on the SMOOTH nets: twice (accumulation), then updateGradInput, then after zeroGradParameters
        onet, shape, rng = _nets(kind, oracle, smooth=True)
        h = Host(hipb, onet, shape)
        h.load_from_oracle()
        x = rng.uniform(-1, 1, shape).astype(np.float32)
        y = onet.forward(x.copy())
        gy = rng.standard_normal(y.shape).astype(np.float32)
        for m in _oleaves(onet):
            if hasattr(m, "gradWeight"):
                m.gradWeight[...] = 0
                m.gradBias[...] = 0
        xd = h.dev_in(x)
        gyd = h.dev_in(gy)
        gy_keep = gyd.clone()
        h.forward(xd)
        h.ok(h.lib.vf_net_zero_grad(h.net))
        gxp = C.c_void_p()
        for _ in range(2):
            gx_want = onet.backward(x.copy(), gy.copy()).copy()
            h.ok(h.lib.vf_net_backward(h.net, C.c_void_p(xd.data_ptr()), C.c_void_p(gyd.data_ptr()), C.byref(gxp)))
        assert torch.equal(gyd, gy_keep), "the caller's gradOutput must stay intact"
        assert rel_err(h.read_act(gxp.value, shape), gx_want) <= 1e-4, "gradInput"
        for i, which, got, t in h.grads():
            # a module's tensors share one scale: the bias of a convolution in front of a BatchNorm has a TRUE gradient of 0
            scale = max(np.abs(h.mods[i].gradWeight).max(), np.abs(h.mods[i].gradBias).max())
            assert np.abs(got - t).max() <= 1e-4 * scale, ("accumulated gradient", i, which, np.abs(got - t).max() / scale)
        before = h.download(h.g, h.count)
        gx2 = C.c_void_p()
        h.ok(h.lib.vf_net_update_grad_input(h.net, C.c_void_p(xd.data_ptr()), C.c_void_p(gyd.data_ptr()), C.byref(gx2)))
        assert rel_err(h.read_act(gx2.value, shape), gx_want) <= 1e-4
        np.testing.assert_array_equal(h.download(h.g, h.count), before)       # parameter gradients untouched
        # zeroGradParameters (lazy: the next backward overwrites) -> one backward = half of the accumulated two
        h.ok(h.lib.vf_net_zero_grad(h.net))
        h.ok(h.lib.vf_net_backward(h.net, C.c_void_p(xd.data_ptr()), C.c_void_p(gyd.data_ptr()), C.byref(gxp)))
        for i, which, got, t in h.grads():
            scale = max(np.abs(h.mods[i].gradWeight).max(), np.abs(h.mods[i].gradBias).max())
            assert np.abs(2 * got - t).max() <= 1e-4 * scale, ("after zeroGradParameters", i, which)
        # skip-input-grad: no gradInput, the same parameter gradients
        h.ok(h.lib.vf_net_set_skip_input_grad(h.net, 1))
        h.ok(h.lib.vf_net_zero_grad(h.net))
        h.ok(h.lib.vf_net_backward(h.net, C.c_void_p(xd.data_ptr()), C.c_void_p(gyd.data_ptr()), C.byref(gxp)))
        assert not gxp.value
        for i, which, got, t in h.grads():
            scale = max(np.abs(h.mods[i].gradWeight).max(), np.abs(h.mods[i].gradBias).max())
            assert np.abs(2 * got - t).max() <= 1e-4 * scale
        # the conv-bias sweep: conv biases zero, BatchNorm betas and every weight untouched (train.lua:279)
        p0 = h.download(h.p, h.count)
        h.ok(h.lib.vf_net_zero_conv_biases(h.net, None))
        p1 = h.download(h.p, h.count)
        for i, m in enumerate(h.mods):
            if not hasattr(m, "weight"):
                continue
            for which in (0, 1):
                ln = C.c_int64()
                off = h.lib.vf_net_param_offset(h.net, i, which, C.byref(ln))
                seg0, seg1 = p0[off:off + ln.value], p1[off:off + ln.value]
                if which == 1 and "Convolution" in type(m).__name__:
                    assert np.all(seg1 == 0) and np.any(seg0 != 0)
                else:
                    np.testing.assert_array_equal(seg0, seg1)
        h.close()
    finally:
        oracle.set_num_threads(1)


def test_net_object_batch_groups_and_group_pass(gate, oracle, hipb):
    """netD over [real; fake] as ONE batch with two BatchNorm groups (train.lua:331-349 run as a 2B batch) == the oracle's two
    separate passes: outputs, running statistics after both, accumulated parameter gradients; then fGx's
    netD:updateGradInput over the fake half only (train.lua:366) == the oracle's pass on the fake batch."""
    oracle.set_num_threads(16)
    try:
        onet, shape, rng = _nets("netD", oracle, smooth=True)
        Bn = shape[0]
        h = Host(hipb, onet, (2 * Bn,) + shape[1:])
        h.load_from_oracle()
        h.ok(h.lib.vf_net_set_batch_groups(h.net, 2))
        real = rng.uniform(-1, 1, shape).astype(np.float32)
        fake = rng.uniform(-1, 1, shape).astype(np.float32)
        for m in _oleaves(onet):
            if hasattr(m, "gradWeight"):
                m.gradWeight[...] = 0
                m.gradBias[...] = 0
        y_r = onet.forward(real.copy()).copy()
        gy_r = rng.standard_normal(y_r.shape).astype(np.float32)
        onet.backward(real.copy(), gy_r.copy())
        y_f = onet.forward(fake.copy()).copy()
        gy_f = rng.standard_normal(y_f.shape).astype(np.float32)
        onet.backward(fake.copy(), gy_f.copy())
        cat = h.dev_in(np.concatenate([real, fake], 0))
        gcat = h.dev_in(np.concatenate([gy_r, gy_f], 0))
        yp = h.forward(cat)
        got = h.download(yp, 2 * y_r.size).reshape(2 * Bn, -1)
        assert rel_err(got, np.concatenate([y_r, y_f], 0).reshape(2 * Bn, -1)) <= 2e-5
        h.ok(h.lib.vf_net_zero_grad(h.net))
        h.ok(h.lib.vf_net_set_skip_input_grad(h.net, 1))
        gxp = C.c_void_p()
        h.ok(h.lib.vf_net_backward(h.net, C.c_void_p(cat.data_ptr()), C.c_void_p(gcat.data_ptr()), C.byref(gxp)))
        for i, which, g, t in h.grads():
            scale = max(np.abs(h.mods[i].gradWeight).max(), np.abs(h.mods[i].gradBias).max())
            assert np.abs(g - t).max() <= 1e-4 * scale, ("2B backward", i, which, np.abs(g - t).max() / scale)
        for i, m in enumerate(h.mods):
            if hasattr(m, "running_mean"):
                rm, rv = C.c_void_p(), C.c_void_p()
                h.ok(h.lib.vf_net_bn_running(h.net, i, C.byref(rm), C.byref(rv)))
                scale = max(np.abs(m.running_mean).max(), np.sqrt(m.running_var).max())
                assert np.abs(h.download(rm.value, m.running_mean.size) - m.running_mean).max() <= 1e-5 * scale
                assert rel_err(h.download(rv.value, m.running_var.size), m.running_var) <= 1e-5
        # the generator's pass: gradient w.r.t. the fake half with the fake pass's saved statistics
        df = rng.standard_normal(y_f.shape).astype(np.float32)
        want = onet.updateGradInput(fake.copy(), df.copy())
        fk, dfd = h.dev_in(fake), h.dev_in(df)
        h.ok(h.lib.vf_net_update_grad_input_group(h.net, C.c_void_p(fk.data_ptr()), C.c_void_p(dfd.data_ptr()), 1, 2, C.byref(gxp)))
        assert rel_err(h.read_act(gxp.value, shape), want) <= 1e-4
        h.close()
    finally:
        oracle.set_num_threads(1)


def test_net_object_cut_walk_reshape_and_bound_storage(gate, oracle, hipb):
    """vf_net_bucket_split + vf_net_backward_range: the walk cut at the bucket boundary equals the uncut walk bit for bit and the
    first part leaves the head of the flat gradient untouched; vf_net_reshape keeps parameters; vf_net_bind_parameters runs the
    same net on host-owned storage with identical results."""
    oracle.set_num_threads(16)
    try:
        onet, shape, rng = _nets("netG", oracle, smooth=True)
        h = Host(hipb, onet, shape)
        h.load_from_oracle()
        x = h.dev_in(rng.uniform(-1, 1, shape).astype(np.float32))
        yp = h.forward(x)
        gy = h.dev_in(rng.standard_normal(shape).astype(np.float32))       # netG's output has the input's shape here (6 -> 6)
        gxp = C.c_void_p()
        h.ok(h.lib.vf_net_zero_grad(h.net))
        h.ok(h.lib.vf_net_backward(h.net, C.c_void_p(x.data_ptr()), C.c_void_p(gy.data_ptr()), C.byref(gxp)))
        full = h.download(h.g, h.count).copy()
        gx_full = h.download(gxp.value, int(np.prod(shape))).copy()
        k, off = C.c_int(), C.c_int64()
        h.ok(h.lib.vf_net_bucket_split(h.net, 0.9, C.byref(k), C.byref(off)))
        n_plan = h.lib.vf_net_plan_size(h.net)
        assert 0 < k.value < n_plan and 0 < off.value < h.count
        h.ok(h.lib.vf_zero(h.ctx, C.c_void_p(h.g), h.count * 4))
        h.ok(h.lib.vf_net_zero_grad(h.net))
        # (no second forward: it would take its statistics about the MOVED running mean and round differently; the walk itself
        #  is repeatable from the saved state of the one forward above)
        mid = C.c_void_p()
        h.ok(h.lib.vf_net_backward_range(h.net, C.c_void_p(x.data_ptr()), C.c_void_p(gy.data_ptr()), -1, k.value, 1, C.byref(mid)))
        part = h.download(h.g, h.count)
        assert np.all(part[:off.value] == 0), "the first part of the walk touched the head bucket"
        np.testing.assert_array_equal(part[off.value:], full[off.value:])
        h.ok(h.lib.vf_net_backward_range(h.net, C.c_void_p(x.data_ptr()), mid, k.value, 0, 1, C.byref(gxp)))
        np.testing.assert_array_equal(h.download(h.g, h.count), full)
        np.testing.assert_array_equal(h.download(gxp.value, int(np.prod(shape))), gx_full)
        # the same cut WITHOUT interrupting the data-gradient chain: one walk, the tail bucket's gradients at its end, the rest
        # at vf_net_backward_finish (what the data-parallel step runs)
        h.ok(h.lib.vf_zero(h.ctx, C.c_void_p(h.g), h.count * 4))
        h.ok(h.lib.vf_net_zero_grad(h.net))
        h.ok(h.lib.vf_net_backward_split(h.net, C.c_void_p(x.data_ptr()), C.c_void_p(gy.data_ptr()), k.value, 1, C.byref(gxp)))
        part = h.download(h.g, h.count)
        np.testing.assert_array_equal(part[off.value:], full[off.value:])
        pending = 0                           # below the cut: BatchNorm gains / shifts are the walk's own, the recorded conv gradients
        for i, m in enumerate(h.mods):       # (all but the thin-channel layers', which launch at once) are still to come
            if type(m).__name__ in ("SpatialConvolution", "SpatialFullConvolution"):
                ln = C.c_int64()
                o = h.lib.vf_net_param_offset(h.net, i, 0, C.byref(ln))
                if o + ln.value <= off.value and np.all(part[o:o + ln.value] == 0):
                    pending += 1
        assert pending >= 2, "the head bucket's weight gradients were launched before vf_net_backward_finish"
        np.testing.assert_array_equal(h.download(gxp.value, int(np.prod(shape))), gx_full)
        assert h.lib.vf_net_backward(h.net, C.c_void_p(x.data_ptr()), C.c_void_p(gy.data_ptr()), C.byref(mid)) != 0      # pending
        h.ok(h.lib.vf_net_backward_finish(h.net))
        np.testing.assert_array_equal(h.download(h.g, h.count), full)
        # reshape: another batch size, same parameters -> the oracle's forward on that batch
        small = (2,) + shape[1:]
        h.ok(h.lib.vf_net_reshape(h.net, *small))
        xs = rng.uniform(-1, 1, small).astype(np.float32)
        want = onet.forward(xs.copy())
        xsd = h.dev_in(xs)
        got = h.download(h.forward(xsd), want.size)
        assert rel_err(got.reshape(want.transpose(0, 2, 3, 1).shape), want.transpose(0, 2, 3, 1)) <= 2e-5
        # host-owned storage: copy the parameters out, bind, same forward bit for bit
        own = torch.from_numpy(h.download(h.p, h.count).copy()).to(h.dev)
        owng = torch.zeros_like(own)
        h.ok(h.lib.vf_net_training(h.net, 0))       # (evaluate mode: a training forward moves the running mean its sums are shifted by)
        before = h.download(h.forward(xsd), want.size).copy()
        h.ok(h.lib.vf_net_bind_parameters(h.net, C.c_void_p(own.data_ptr()), C.c_void_p(owng.data_ptr()), own.numel()))
        np.testing.assert_array_equal(h.download(h.forward(xsd), want.size), before)
        assert h.lib.vf_net_bind_parameters(h.net, C.c_void_p(own.data_ptr()), None, own.numel()) != 0
        h.close()
        # a shape mismatch is an error with a message
        bad = (Desc * 1)(Desc(CONV, 5, 8, 4, 2, 1, 0, 0.0, 0.0, 0.0))
        n2 = C.c_void_p()
        assert hipb.lib.vf_net_create(hipb.ctx, C.byref(n2), bad, 1, 2, 3, 8, 8) != 0 and b"input planes" in hipb.lib.vf_last_error()
    finally:
        oracle.set_num_threads(1)


def test_net_object_activation_observer(oracle, hipb):
    """the observer sees every fused (Leaky)ReLU output of a forward, in order, with the right sizes, and an edit it makes is
    what the layers behind it read"""
    onet, shape, rng = _nets("netD64", oracle, smooth=False)
    h = Host(hipb, onet, shape)
    h.load_from_oracle()
    seen = []
    CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int64)

    def cb(user, layer, ptr, numel):
        seen.append((layer, numel))
        return 0

    fn = CB(cb)
    h.ok(h.lib.vf_net_set_act_observer(h.net, C.cast(fn, C.c_void_p), None))
    x = h.dev_in(rng.uniform(-1, 1, shape).astype(np.float32))
    h.forward(x)
    acts = [i for i, m in enumerate(h.mods) if type(m).__name__ == "LeakyReLU"]
    assert [l for l, _ in seen] == acts and all(n > 0 for _, n in seen)
    h.ok(h.lib.vf_net_set_act_observer(h.net, None, None))
    seen.clear()
    h.forward(x)
    assert not seen
    h.close()


def test_net_object_conv_relu_then_batchnorm_chain(oracle, hipb):
    """conv(4x4 s2) + LeakyReLU -> BatchNorm -> conv: an ordering the reference nets do not have but vf_net / hipnn.Net accept.
    The BatchNorm's backward writes the planes of its gradInput for the convolution below; that convolution first undoes its fused
    LeakyReLU on the gradient IN PLACE, so the planes are stale and must not be used (the data-gradient and the weight
    gradient of that layer were formed from the unmasked gradient).  Real LeakyReLU(0.2): the bug is an O(1) error, a kink flip
    moves 1e-3, so 2e-2 separates them.  Planes gate dropped so that the planes path is what runs."""
    lib = hipb.lib
    assert lib.vf_net_set_planes_gate(0.0, 1) == 0
    oracle.set_num_threads(16)
    try:
        rng = np.random.default_rng(11)
        net = oracle.Sequential()
        for m in (oracle.SpatialConvolution(16, 64, 4, 4, 2, 2, 1, 1), oracle.LeakyReLU(0.2, True),
                  oracle.SpatialConvolution(64, 64, 4, 4, 2, 2, 1, 1), oracle.LeakyReLU(0.2, True), oracle.SpatialBatchNormalization(64),
                  oracle.SpatialConvolution(64, 128, 4, 4, 2, 2, 1, 1), oracle.LeakyReLU(0.2, True)):
            net.add(m)
        oracle.weights_init(net, rng)
        shape = (16, 16, 64, 64)
        h = Host(hipb, net, shape)
        h.load_from_oracle()
        x = rng.uniform(-1, 1, shape).astype(np.float32)
        y = net.forward(x.copy())
        gy = rng.standard_normal(y.shape).astype(np.float32)
        for m in _oleaves(net):
            if hasattr(m, "gradWeight"):
                m.gradWeight[...] = 0
                m.gradBias[...] = 0
        gx_want = net.backward(x.copy(), gy.copy()).copy()
        xd, gyd = h.dev_in(x), h.dev_in(gy)
        got = h.download(h.forward(xd), y.size)
        assert rel_err(got.reshape(y.transpose(0, 2, 3, 1).shape), y.transpose(0, 2, 3, 1)) <= 2e-5
        h.ok(h.lib.vf_net_zero_grad(h.net))
        gxp = C.c_void_p()
        h.ok(h.lib.vf_net_backward(h.net, C.c_void_p(xd.data_ptr()), C.c_void_p(gyd.data_ptr()), C.byref(gxp)))
        assert rel_err(h.read_act(gxp.value, shape), gx_want) <= 2e-2, "gradInput"
        for i, which, got, t in h.grads():
            scale = max(np.abs(h.mods[i].gradWeight).max(), np.abs(h.mods[i].gradBias).max())
            assert np.abs(got - t).max() <= 2e-2 * scale, ("gradient", i, which, np.abs(got - t).max() / scale)
        h.close()
    finally:
        oracle.set_num_threads(1)
        assert lib.vf_net_set_planes_gate(3.0, 1024) == 0


def test_syncbn_without_a_communicator_is_never_silently_local(oracle, hipb):
    """with the C-ABI host and no vf_comm communicator (bench.py --comm torch, or the --comm auto fallback) a SyncBN
    request used to reach vf_net_set_sync_bn with a NULL communicator and every rank normalised with its OWN statistics while the
    bench line said `sync`.  Now: the library refuses world > 1 without a communicator, and the trainers keep such nets on the
    module-by-module host, which exchanges the sums over torch.distributed."""
    from video_filler_amd import nn, trainers
    saved, hipb.comm = hipb.comm, None        # (an earlier test of the session may have attached a one-rank communicator)
    try:
        onet, shape, rng = _nets("netD64", oracle, smooth=True)
        h = Host(hipb, onet, shape)
        assert h.lib.vf_net_set_sync_bn(h.net, None, 2, 0) != 0 and b"communicator" in h.lib.vf_last_error()
        assert h.lib.vf_net_set_sync_bn(h.net, None, 1, 0) == 0
        h.close()
        netG = trainers.build_netG(3, 3, 16, 16, 32, False, True, True, False)
        netD = trainers.build_netD(3, 16, False, True, True, False)
        g1, d1, host = trainers._host_nets("cabi", netG, netD, sync_world=2)
        assert host == "mirror" and g1 is netG and d1 is netD
        g2, d2, host = trainers._host_nets("cabi", netG, netD, sync_world=1)
        assert host == "cabi" and type(g2).__name__ == "CNet"
    finally:
        hipb.comm = saved


def test_derivative_mask_from_sign_bits_equals_the_fp32_mask(hipb):
    """The data-gradient of the second convolution masks with the derivative of the LeakyReLU below it (train.lua:183-186: conv ->
    LeakyReLU -> conv).  vf_net reads that mask as SIGN BITS which the thin-input convolution leaves beside its output (2 MB instead
    of the 67 MB activation at batchSize 64); the module-by-module host reads the fp32 activation.  Same kernels otherwise: gradInput
    and every parameter gradient of the two hosts must be the same bits — on the REAL net (LeakyReLU(0.2): a wrong mask bit is an
    O(1) error), for the whole batch and for a group pass over the second half of a two-group batch."""
    from video_filler_amd import nn, trainers
    from video_filler_amd.cnet import CNet, adopt_if_chain
    old = (nn._PCONV_MIN_GFLOP, nn._PCONV_MIN_ROWS)
    nn._PCONV_MIN_GFLOP, nn._PCONV_MIN_ROWS = 0.0, 1
    try:
        g = torch.Generator().manual_seed(21)
        mirror = trainers.build_netD(3, 64, False, True, True, False)
        cab = adopt_if_chain(trainers.build_netD(3, 64, False, True, True, False))
        assert isinstance(cab, CNet)
        pm, _ = mirror.getParameters()
        pc, _ = cab.getParameters()
        assert pm.shape == pc.shape
        pm.copy_(torch.randn(pm.shape, generator=g).to(pm.device) * 0.05)
        pc.copy_(pm)
        x = (torch.rand((16, 3, 64, 64), generator=g) * 2 - 1).to(hipb.device)
        for net in (mirror, cab):
            net.setBatchGroups(2)
            net.zeroGradParameters()
        ym, yc = mirror.forward(x), cab.forward(x)
        assert torch.equal(ym, yc)
        assert hipb.lib.vf_net_layer_has_act_bits(cab._net, 0) == 1, "the first conv did not leave its sign bits: the test would compare nothing"
        gy = torch.randn(ym.shape, generator=g).to(hipb.device)
        gm, gc = mirror.backward(x, gy).clone(), cab.backward(x, gy).clone()
        torch.cuda.synchronize()
        assert torch.equal(gm, gc), float((gm - gc).abs().max())
        assert torch.equal(mirror.reference_flat(grads=True), cab.reference_flat(grads=True))
        # the generator's pass over the fake half only (fGx: netD:updateGradInput with the saved activations of group 2 of 2)
        h = x.shape[0] // 2
        gm2 = mirror.updateGradInput(x[h:], gy[h:], group=(1, 2)).clone()
        gc2 = cab.updateGradInput(x[h:], gy[h:], group=(1, 2)).clone()
        torch.cuda.synchronize()
        assert torch.equal(gm2, gc2), float((gm2 - gc2).abs().max())
    finally:
        nn._PCONV_MIN_GFLOP, nn._PCONV_MIN_ROWS = old
