"""CPU oracle: numpy orchestration over oracle/vf_oracle.cpp.

TEST INFRASTRUCTURE ONLY — importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  Nothing under video-filler_amd/ imports this.

PARITY UNPINNED (see vf_oracle.cpp header): the reference has no golden vectors and
Torch7 cannot run here; this file restates, module by module, the Torch7 `nn` object
protocol the reference drivers use, in the reference's own NCHW fp32 layout:

  * containers / module protocol ........ SURVEY A.5  (nn.Sequential:forward/backward/updateGradInput)
  * net topologies ....................... train.lua:87-199, train_vid_weighted.lua:112-236,
                                           train_wholeim_input.lua:137-260
  * iteration algebra fDx / fGx .......... train.lua:278-410, train_vid_weighted.lua:373-537
  * flat parameters ...................... SURVEY A.11 (net:getParameters())
  * optim.adam ........................... SURVEY A.10 (train.lua:219-226,421-424)
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

f32p = C.POINTER(C.c_float)
u8p = C.POINTER(C.c_uint8)


def build(force=False):
    """Compile libvf_oracle.so (g++).  Falls back to generic flags if the host lacks AVX2."""
    so = os.path.join(_HERE, "libvf_oracle.so")
    src = os.path.join(_HERE, "vf_oracle.cpp")
    have_avx2 = False
    try:
        with open("/proc/cpuinfo") as fh:
            have_avx2 = " avx2 " in fh.read().replace("\n", " ")
    except OSError:
        pass
    stale = (not os.path.exists(so)) or os.path.getmtime(so) < os.path.getmtime(src)
    if force or stale or not have_avx2:
        target = "all" if have_avx2 else "generic"
        if stale or force or target == "generic":
            subprocess.check_call(["make", "-C", _HERE, "-B", target], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        L = _LIB
        for name in ("vfo_bce_fwd", "vfo_mse_fwd", "vfo_abs_fwd", "vfo_gdl_fwd", "vfo_masked_mse_fwd"):
            getattr(L, name).restype = C.c_double
        L.vfo_get_num_threads.restype = C.c_int
    return _LIB


def set_num_threads(n):
    lib().vfo_set_num_threads(int(n))


def _p(a):
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"], (a.dtype, a.flags)
    return a.ctypes.data_as(f32p)


def _pn(a):
    return None if a is None else _p(a)


def _f(x):
    return C.c_float(float(x))


def _sz(n):
    return C.c_size_t(int(n))


# ------------------------------------------------------------------ modules
class Module:
    act_trace = None      # tests: a list that receives (module, activated output copy) from every (Leaky)ReLU forward

    def __init__(self):
        self.output = None
        self.gradInput = None
        self.train = True

    def forward(self, x):
        return self.updateOutput(x)

    def backward(self, x, gy, scale=1.0):
        self.updateGradInput(x, gy)
        self.accGradParameters(x, gy, scale)
        return self.gradInput

    def accGradParameters(self, x, gy, scale=1.0):
        pass

    def parameters(self):
        return [], []

    def apply(self, fn):
        fn(self)

    def training(self):
        self.apply(lambda m: setattr(m, "train", True))

    def evaluate(self):
        self.apply(lambda m: setattr(m, "train", False))

    def type_name(self):
        return "nn." + type(self).__name__


class SpatialConvolution(Module):
    def __init__(self, nIn, nOut, kW, kH, dW=1, dH=1, padW=0, padH=0):
        super().__init__()
        self.nInputPlane, self.nOutputPlane = nIn, nOut
        self.kW, self.kH, self.dW, self.dH, self.padW, self.padH = kW, kH, dW, dH, padW, padH
        self.weight = np.zeros((nOut, nIn, kH, kW), np.float32)
        self.bias = np.zeros((nOut,), np.float32)
        self.gradWeight = np.zeros_like(self.weight)
        self.gradBias = np.zeros_like(self.bias)

    def _dims(self, x):
        B, Cin, H, W = x.shape
        assert Cin == self.nInputPlane
        return (B, Cin, H, W, self.nOutputPlane, self.kH, self.kW, self.dH, self.dW, self.padH, self.padW)

    def out_hw(self, H, W):
        return ((H + 2 * self.padH - self.kH) // self.dH + 1, (W + 2 * self.padW - self.kW) // self.dW + 1)

    def updateOutput(self, x):
        B, _, H, W = x.shape
        Ho, Wo = self.out_hw(H, W)
        if self.output is None or self.output.shape != (B, self.nOutputPlane, Ho, Wo):
            self.output = np.empty((B, self.nOutputPlane, Ho, Wo), np.float32)
        lib().vfo_conv2d_fwd(_p(x), _p(self.weight), _p(self.bias), _p(self.output), *self._dims(x))
        return self.output

    def updateGradInput(self, x, gy):
        if self.gradInput is None or self.gradInput.shape != x.shape:
            self.gradInput = np.empty(x.shape, np.float32)
        lib().vfo_conv2d_bwd_input(_p(gy), _p(self.weight), _p(self.gradInput), *self._dims(x))
        return self.gradInput

    def accGradParameters(self, x, gy, scale=1.0):
        lib().vfo_conv2d_acc_grad(_p(x), _p(gy), _p(self.gradWeight), _p(self.gradBias), *self._dims(x), _f(scale))

    def parameters(self):
        return [self.weight, self.bias], [self.gradWeight, self.gradBias]


class SpatialFullConvolution(SpatialConvolution):
    def __init__(self, nIn, nOut, kW, kH, dW=1, dH=1, padW=0, padH=0):
        super().__init__(nIn, nOut, kW, kH, dW, dH, padW, padH)
        self.weight = np.zeros((nIn, nOut, kH, kW), np.float32)
        self.gradWeight = np.zeros_like(self.weight)

    def out_hw(self, H, W):
        return ((H - 1) * self.dH - 2 * self.padH + self.kH, (W - 1) * self.dW - 2 * self.padW + self.kW)

    def updateOutput(self, x):
        B, _, H, W = x.shape
        Ho, Wo = self.out_hw(H, W)
        if self.output is None or self.output.shape != (B, self.nOutputPlane, Ho, Wo):
            self.output = np.empty((B, self.nOutputPlane, Ho, Wo), np.float32)
        lib().vfo_fullconv2d_fwd(_p(x), _p(self.weight), _p(self.bias), _p(self.output), *self._dims(x))
        return self.output

    def updateGradInput(self, x, gy):
        if self.gradInput is None or self.gradInput.shape != x.shape:
            self.gradInput = np.empty(x.shape, np.float32)
        lib().vfo_fullconv2d_bwd_input(_p(gy), _p(self.weight), _p(self.gradInput), *self._dims(x))
        return self.gradInput

    def accGradParameters(self, x, gy, scale=1.0):
        lib().vfo_fullconv2d_acc_grad(_p(x), _p(gy), _p(self.gradWeight), _p(self.gradBias), *self._dims(x), _f(scale))


class SpatialBatchNormalization(Module):
    def __init__(self, C_, eps=1e-5, momentum=0.1):
        super().__init__()
        self.nOutputPlane = C_
        self.eps, self.momentum = eps, momentum
        self.weight = np.ones((C_,), np.float32)
        self.bias = np.zeros((C_,), np.float32)
        self.gradWeight = np.zeros_like(self.weight)
        self.gradBias = np.zeros_like(self.bias)
        self.running_mean = np.zeros((C_,), np.float32)
        self.running_var = np.ones((C_,), np.float32)
        self.save_mean = np.zeros((C_,), np.float32)
        self.save_std = np.zeros((C_,), np.float32)  # holds invstd, as THNN's save_std does

    def updateOutput(self, x):
        B, Cc, H, W = x.shape
        if self.output is None or self.output.shape != x.shape:
            self.output = np.empty(x.shape, np.float32)
        if self.train:
            lib().vfo_bn_train_fwd(_p(x), _p(self.output), _p(self.weight), _p(self.bias), _p(self.running_mean),
                                   _p(self.running_var), _p(self.save_mean), _p(self.save_std), B, Cc, H * W,
                                   _f(self.momentum), _f(self.eps))
        else:
            lib().vfo_bn_eval_fwd(_p(x), _p(self.output), _p(self.weight), _p(self.bias), _p(self.running_mean),
                                  _p(self.running_var), B, Cc, H * W, _f(self.eps))
        return self.output

    def _bwd(self, x, gy, want_gx, want_gp, scale):
        B, Cc, H, W = x.shape
        assert self.train, "reference never back-propagates in evaluate mode"
        if want_gx and (self.gradInput is None or self.gradInput.shape != x.shape):
            self.gradInput = np.empty(x.shape, np.float32)
        lib().vfo_bn_bwd(_p(x), _p(gy), _p(self.gradInput) if want_gx else None,
                         _p(self.gradWeight) if want_gp else None, _p(self.gradBias) if want_gp else None,
                         _p(self.weight), _p(self.save_mean), _p(self.save_std), B, Cc, H * W, _f(scale))

    def updateGradInput(self, x, gy):
        self._bwd(x, gy, True, False, 1.0)
        return self.gradInput

    def accGradParameters(self, x, gy, scale=1.0):
        self._bwd(x, gy, False, True, scale)

    def backward(self, x, gy, scale=1.0):
        self._bwd(x, gy, True, True, scale)
        return self.gradInput

    def parameters(self):
        return [self.weight, self.bias], [self.gradWeight, self.gradBias]


class LeakyReLU(Module):
    def __init__(self, negval=0.01, inplace=False):
        super().__init__()
        self.negval, self.inplace = negval, inplace

    def updateOutput(self, x):
        self.output = x if self.inplace else x.copy()
        lib().vfo_lrelu_fwd(_p(self.output), _sz(self.output.size), _f(self.negval))
        if Module.act_trace is not None:
            Module.act_trace.append((self, self.output.copy()))
        return self.output

    def updateGradInput(self, x, gy):
        # in-place form: `x` IS the activated output (the producer's .output was overwritten)
        self.gradInput = gy if self.inplace else gy.copy()
        ref = x if self.inplace else self.output
        lib().vfo_lrelu_bwd(_p(ref), _p(self.gradInput), _sz(gy.size), _f(self.negval))
        return self.gradInput


class ReLU(Module):
    def __init__(self, inplace=False):
        super().__init__()
        self.inplace = inplace

    def updateOutput(self, x):
        self.output = x if self.inplace else x.copy()
        lib().vfo_relu_fwd(_p(self.output), _sz(self.output.size))
        if Module.act_trace is not None:
            Module.act_trace.append((self, self.output.copy()))
        return self.output

    def updateGradInput(self, x, gy):
        self.gradInput = gy if self.inplace else gy.copy()
        ref = x if self.inplace else self.output
        lib().vfo_relu_bwd(_p(ref), _p(self.gradInput), _sz(gy.size))
        return self.gradInput


class Tanh(Module):
    def updateOutput(self, x):
        if self.output is None or self.output.shape != x.shape:
            self.output = np.empty(x.shape, np.float32)
        lib().vfo_tanh_fwd(_p(x), _p(self.output), _sz(x.size))
        return self.output

    def updateGradInput(self, x, gy):
        if self.gradInput is None or self.gradInput.shape != x.shape:
            self.gradInput = np.empty(x.shape, np.float32)
        lib().vfo_tanh_bwd(_p(self.output), _p(gy), _p(self.gradInput), _sz(x.size))
        return self.gradInput


class Sigmoid(Module):
    def updateOutput(self, x):
        if self.output is None or self.output.shape != x.shape:
            self.output = np.empty(x.shape, np.float32)
        lib().vfo_sigmoid_fwd(_p(x), _p(self.output), _sz(x.size))
        return self.output

    def updateGradInput(self, x, gy):
        if self.gradInput is None or self.gradInput.shape != x.shape:
            self.gradInput = np.empty(x.shape, np.float32)
        lib().vfo_sigmoid_bwd(_p(self.output), _p(gy), _p(self.gradInput), _sz(x.size))
        return self.gradInput


class View(Module):
    """nn.View(1):setNumInputDims(3): B x 1 x 1 x 1 -> B x 1 (shares storage)."""

    def __init__(self, *sizes):
        super().__init__()
        self.sizes = sizes
        self.numInputDims = None

    def setNumInputDims(self, n):
        self.numInputDims = n
        return self

    def updateOutput(self, x):
        self.output = x.reshape(x.shape[0], *self.sizes)
        return self.output

    def updateGradInput(self, x, gy):
        self.gradInput = gy.reshape(x.shape)
        return self.gradInput


class JoinTable(Module):
    """nn.JoinTable(2) on a table of NCHW tensors (train.lua:119,172)."""

    def __init__(self, dimension):
        super().__init__()
        assert dimension == 2
        self.dimension = dimension

    def updateOutput(self, xs):
        self.output = np.ascontiguousarray(np.concatenate(xs, axis=1))
        return self.output

    def updateGradInput(self, xs, gy):
        out, off = [], 0
        for x in xs:
            out.append(np.ascontiguousarray(gy[:, off:off + x.shape[1]]))
            off += x.shape[1]
        self.gradInput = out
        return out

    def parameters(self):
        return [], []


class ParallelTable(Module):
    """nn.ParallelTable (train.lua:115-117,168-170): member i on table element i."""

    def __init__(self):
        super().__init__()
        self.modules = []

    def add(self, m):
        self.modules.append(m)
        return self

    def apply(self, fn):
        fn(self)
        for m in self.modules:
            m.apply(fn)

    def updateOutput(self, xs):
        self.output = [m.updateOutput(x) for m, x in zip(self.modules, xs)]
        return self.output

    def updateGradInput(self, xs, gys):
        self.gradInput = [m.updateGradInput(x, g) for m, x, g in zip(self.modules, xs, gys)]
        return self.gradInput

    def backward(self, xs, gys, scale=1.0):
        self.gradInput = [m.backward(x, g, scale) for m, x, g in zip(self.modules, xs, gys)]
        return self.gradInput

    def parameters(self):
        ws, gs = [], []
        for m in self.modules:
            w, g = m.parameters()
            ws += w
            gs += g
        return ws, gs


class Sequential(Module):
    def __init__(self):
        super().__init__()
        self.modules = []

    def add(self, m):
        self.modules.append(m)
        return self

    def apply(self, fn):
        fn(self)
        for m in self.modules:
            m.apply(fn)

    def updateOutput(self, x):
        cur = x
        for m in self.modules:
            cur = m.updateOutput(cur)
        self.output = cur
        return cur

    def _walk(self, x, gy, method, *extra):
        g = gy
        cur = self.modules[-1]
        for i in range(len(self.modules) - 2, -1, -1):
            prev = self.modules[i]
            g = getattr(cur, method)(prev.output, g, *extra)
            cur.gradInput = g
            cur = prev
        g = getattr(cur, method)(x, g, *extra)
        self.gradInput = g
        return g

    def updateGradInput(self, x, gy):
        return self._walk(x, gy, "updateGradInput")

    def backward(self, x, gy, scale=1.0):
        return self._walk(x, gy, "backward", scale)

    def accGradParameters(self, x, gy, scale=1.0):
        raise NotImplementedError("drivers never call Sequential:accGradParameters directly")

    def parameters(self):
        ws, gs = [], []
        for m in self.modules:
            w, g = m.parameters()
            ws += w
            gs += g
        return ws, gs

    def getParameters(self):
        """depth-first {weight, bias} per module, flattened into ONE storage; the
        modules' tensors become views of it."""
        owners = []

        def collect(m):
            if isinstance(m, (Sequential, ParallelTable)):
                for c in m.modules:
                    collect(c)
            elif hasattr(m, "weight"):
                owners.append(m)

        collect(self)
        n = sum(m.weight.size + m.bias.size for m in owners)
        flat, gflat = np.zeros((n,), np.float32), np.zeros((n,), np.float32)
        off = 0
        for m in owners:
            for name, gname in (("weight", "gradWeight"), ("bias", "gradBias")):
                t = getattr(m, name)
                g = getattr(m, gname)
                flat[off:off + t.size] = t.ravel()
                gflat[off:off + t.size] = g.ravel()
                setattr(m, name, flat[off:off + t.size].reshape(t.shape))
                setattr(m, gname, gflat[off:off + t.size].reshape(t.shape))
                off += t.size
        return flat, gflat


# ------------------------------------------------------------------ criteria
class BCECriterion:
    def forward(self, x, t):
        return lib().vfo_bce_fwd(_p(x.reshape(-1)), _p(t), _sz(x.size))

    def backward(self, x, t):
        g = np.empty(x.shape, np.float32)
        lib().vfo_bce_bwd(_p(x.reshape(-1)), _p(t), _p(g.reshape(-1)), _sz(x.size))
        return g


class MSECriterion:
    def forward(self, x, t):
        return lib().vfo_mse_fwd(_p(x), _p(t), _sz(x.size))

    def backward(self, x, t):
        g = np.empty(x.shape, np.float32)
        lib().vfo_mse_bwd(_p(x), _p(t), _p(g), _sz(x.size))
        return g


class GDLCriterion:
    def __init__(self, alpha=1):
        assert alpha == 1  # gdl_criterion.lua:9

    def forward(self, x, t):
        B, Cc, H, W = x.shape
        return lib().vfo_gdl_fwd(_p(x), _p(t), B, Cc, H, W)

    def backward(self, x, t):
        """gdl_criterion.lua:47-53 (no driver calls it)"""
        B, Cc, H, W = x.shape
        g = np.empty(x.shape, np.float32)
        rc = lib().vfo_gdl_bwd(_p(x), _p(t), _p(g), B, Cc, H, W)
        assert rc == 0, "GDLCriterion needs square maps"
        return g


class MaskedMSECriterion:
    def __init__(self, mWeight=1.0):
        self.mWeight = mWeight
        self.mask = None

    def setMask(self, m):
        assert m.dtype == np.uint8  # MaskedMSECriterion.lua:25
        self.mask = np.ascontiguousarray(m)

    def forward(self, x, t):
        return lib().vfo_masked_mse_fwd(_p(x), _p(t), self.mask.ctypes.data_as(u8p), _f(self.mWeight), _sz(x.size))

    def backward(self, x, t):
        g = np.empty(x.shape, np.float32)
        lib().vfo_masked_mse_bwd(_p(x), _p(t), self.mask.ctypes.data_as(u8p), _f(self.mWeight), _p(g), _sz(x.size))
        return g


# ------------------------------------------------------------------ optim.adam
def adam(opfunc, x, state):
    """optim.adam(opfunc, x, config) with state kept in `state`."""
    lr = state.get("learningRate", 0.001)
    beta1 = state.get("beta1", 0.9)
    beta2 = state.get("beta2", 0.999)
    eps = state.get("epsilon", 1e-8)
    fx, dfdx = opfunc(x)
    if "t" not in state:
        state["t"] = 0
        state["m"] = np.zeros_like(dfdx)
        state["v"] = np.zeros_like(dfdx)
        state["denom"] = np.zeros_like(dfdx)
    state["t"] += 1
    lib().vfo_adam_step(_p(x), _p(dfdx), _p(state["m"]), _p(state["v"]), _p(state["denom"]), _sz(x.size),
                        C.c_double(lr), C.c_double(beta1), C.c_double(beta2), C.c_double(eps), state["t"])
    return x, fx


# ------------------------------------------------------------------ nets
def _conv(nIn, nOut, s2=True):
    return SpatialConvolution(nIn, nOut, 4, 4, 2, 2, 1, 1) if s2 else SpatialConvolution(nIn, nOut, 4, 4)


def _full(nIn, nOut, s2=True):
    return SpatialFullConvolution(nIn, nOut, 4, 4, 2, 2, 1, 1) if s2 else SpatialFullConvolution(nIn, nOut, 4, 4)


def build_netG(nc_in, nc_out, nef, ngf, nBottleneck, extra_decoder_layer, smooth=False, noise_nz=0, half_last=False,
               extra_bottleneck_stage=False):
    """noise_nz > 0: the noiseGen generator of train.lua:109-124 (input {context, noise[B, nz, 1, 1]}).
    half_last: train_logo_withmask.lua:95-98 — the extra decoder layer is ngf -> ngf/2 (BN over ngf/2).
    train.lua:87-148 (extra_decoder_layer=False, output nc x 64 x 64) and
    train_vid_weighted.lua:112-176 / train_wholeim_input.lua:137-199 (True, output nc_out x 128 x 128).
    smooth=True (tests only) replaces every LeakyReLU(0.2)/ReLU by LeakyReLU(1.0): same graph and kernels, but no
    derivative discontinuity, so gradients can be compared at fp32 precision."""
    LeakyReLU, ReLU = _acts(smooth)
    netE = Sequential()
    netE.add(_conv(nc_in, nef)).add(LeakyReLU(0.2, True))
    netE.add(_conv(nef, nef)).add(SpatialBatchNormalization(nef)).add(LeakyReLU(0.2, True))
    netE.add(_conv(nef, nef * 2)).add(SpatialBatchNormalization(nef * 2)).add(LeakyReLU(0.2, True))
    netE.add(_conv(nef * 2, nef * 4)).add(SpatialBatchNormalization(nef * 4)).add(LeakyReLU(0.2, True))
    netE.add(_conv(nef * 4, nef * 8)).add(SpatialBatchNormalization(nef * 8)).add(LeakyReLU(0.2, True))
    if extra_bottleneck_stage:      # the labelled 256x256 extension (not in the reference: see video-filler_amd/trainers.py)
        netE.add(_conv(nef * 8, nef * 8)).add(SpatialBatchNormalization(nef * 8)).add(LeakyReLU(0.2, True))
    netE.add(_conv(nef * 8, nBottleneck, s2=False))
    netG = Sequential()
    nz_size = nBottleneck
    if noise_nz:
        netG_noise = Sequential().add(SpatialConvolution(noise_nz, noise_nz, 1, 1, 1, 1, 0, 0))
        netG.add(ParallelTable().add(netE).add(netG_noise))
        netG.add(JoinTable(2))
        nz_size = nBottleneck + noise_nz
    else:
        netG.add(netE)
    netG.add(SpatialBatchNormalization(nz_size)).add(LeakyReLU(0.2, True))
    netG.add(_full(nz_size, ngf * 8, s2=False)).add(SpatialBatchNormalization(ngf * 8)).add(ReLU(True))
    if extra_bottleneck_stage:
        netG.add(_full(ngf * 8, ngf * 8)).add(SpatialBatchNormalization(ngf * 8)).add(ReLU(True))
    netG.add(_full(ngf * 8, ngf * 4)).add(SpatialBatchNormalization(ngf * 4)).add(ReLU(True))
    netG.add(_full(ngf * 4, ngf * 2)).add(SpatialBatchNormalization(ngf * 2)).add(ReLU(True))
    netG.add(_full(ngf * 2, ngf)).add(SpatialBatchNormalization(ngf)).add(ReLU(True))
    last = ngf
    if extra_decoder_layer:
        last = ngf // 2 if half_last else ngf
        netG.add(_full(ngf, last)).add(SpatialBatchNormalization(last)).add(ReLU(True))
    netG.add(_full(last, nc_out)).add(Tanh())
    return netG


def _acts(smooth):
    if not smooth:
        return globals()["LeakyReLU"], globals()["ReLU"]
    return (lambda negval, inplace: globals()["LeakyReLU"](1.0, inplace)), (lambda inplace: globals()["LeakyReLU"](1.0, inplace))


def build_netD(nc, ndf, extra_first_layer, smooth=False, conditionAdv=False, extra_last_layer=False):
    """train.lua:157-199 (64x64 input; conditionAdv: input {context 128x128, prediction 64x64}, :158-180) and
    train_vid_weighted.lua:213-236 (128x128 input, extra floor(ndf/2)-wide first layer)."""
    LeakyReLU, ReLU = _acts(smooth)
    netD = Sequential()
    if conditionAdv:
        assert not extra_first_layer
        netD_ctx = Sequential().add(SpatialConvolution(nc, ndf, 5, 5, 2, 2, 2, 2))
        netD_pred = Sequential().add(SpatialConvolution(nc, ndf, 5, 5, 2, 2, 2 + 32, 2 + 32))
        netD.add(ParallelTable().add(netD_ctx).add(netD_pred))
        netD.add(JoinTable(2))
        netD.add(LeakyReLU(0.2, True))
        netD.add(_conv(ndf * 2, ndf)).add(SpatialBatchNormalization(ndf)).add(LeakyReLU(0.2, True))
    elif extra_first_layer:
        mylayer = ndf // 2
        netD.add(_conv(nc, mylayer)).add(LeakyReLU(0.2, True))
        netD.add(_conv(mylayer, ndf)).add(LeakyReLU(0.2, True))
    else:
        netD.add(_conv(nc, ndf)).add(LeakyReLU(0.2, True))
    netD.add(_conv(ndf, ndf * 2)).add(SpatialBatchNormalization(ndf * 2)).add(LeakyReLU(0.2, True))
    netD.add(_conv(ndf * 2, ndf * 4)).add(SpatialBatchNormalization(ndf * 4)).add(LeakyReLU(0.2, True))
    netD.add(_conv(ndf * 4, ndf * 8)).add(SpatialBatchNormalization(ndf * 8)).add(LeakyReLU(0.2, True))
    if extra_last_layer:      # the labelled 256x256 extension (not in the reference: its netD fails at that size)
        netD.add(_conv(ndf * 8, ndf * 8)).add(SpatialBatchNormalization(ndf * 8)).add(LeakyReLU(0.2, True))
    netD.add(_conv(ndf * 8, 1, s2=False)).add(Sigmoid())
    netD.add(View(1).setNumInputDims(3))
    return netD


def weights_init(net, rng):
    """train.lua:58-67: Convolution -> N(0,0.02), bias 0; BatchNormalization -> N(1,0.02), bias 0.
    Torch's MT19937 stream is not reproducible here, so `rng` is a numpy Generator."""

    def init(m):
        name = m.type_name()
        if "Convolution" in name:
            m.weight[...] = rng.normal(0.0, 0.02, m.weight.shape).astype(np.float32)
            m.bias[...] = 0
        elif "BatchNormalization" in name:
            m.weight[...] = rng.normal(1.0, 0.02, m.weight.shape).astype(np.float32)
            m.bias[...] = 0

    net.apply(init)


def zero_conv_biases(net):
    def z(m):
        if "Convolution" in m.type_name():
            m.bias[...] = 0

    net.apply(z)


DEFAULT_OPT_TRAIN = dict(batchSize=64, fineSize=128, nBottleneck=100, nef=64, ngf=64, ndf=64, nc=3, wtl2=0.0,
                         overlapPred=0, lr=0.0002, beta1=0.5, nz=100, conditionAdv=False, noiseGen=False,
                         noisetype="normal")
DEFAULT_OPT_VID = dict(batchSize=16, fineSize=128, nBottleneck=4000, nef=64, ngf=64, ndf=64, nc=3, predLen=4,
                       wtl2=0.999, weight_nomask=0.05, wtgdl=0.0, overlapPred=0, lr=0.0002, beta1=0.5,
                       nc_in=None, nc_out=None)


def _solver(opt):
    wt = opt["wtl2"]
    lrG = opt["lr"] * 10 if (wt > 0 and wt < 1) else opt["lr"]
    return ({"learningRate": lrG, "beta1": opt["beta1"]}, {"learningRate": opt["lr"], "beta1": opt["beta1"]})


class CenterTrainer:
    """train.lua (configs 1-2): centre-square inpainting, D on the 64x64 centre."""

    def __init__(self, opt, rng):
        o = dict(DEFAULT_OPT_TRAIN)
        o.update(opt)
        self.opt = o
        self.netG = build_netG(o["nc"], o["nc"], o["nef"], o["ngf"], o["nBottleneck"], False, o.get("smooth", False),
                               noise_nz=o["nz"] if o["noiseGen"] else 0)
        self.netD = build_netD(o["nc"], o["ndf"], False, o.get("smooth", False), conditionAdv=bool(o["conditionAdv"]))
        weights_init(self.netG, rng)
        weights_init(self.netD, rng)
        self.noise = None                 # [B, nz, 1, 1]; set_noise() or drawn with noise_fill(seed, t)
        self.noise_seed = 1234
        self.criterion = BCECriterion()
        self.criterionMSE = MSECriterion() if o["wtl2"] != 0 else None
        self.optimStateG, self.optimStateD = _solver(o)
        self.parametersD, self.gradParametersD = self.netD.getParameters()
        self.parametersG, self.gradParametersG = self.netG.getParameters()
        self.errD = self.errG = self.errG_l2 = None
        self.batch = None

    def set_batch(self, real_ctx):
        self.batch = np.ascontiguousarray(real_ctx, np.float32).copy()

    def set_noise(self, noise):
        """Fix the noise of the next iterations (tests); None: draw noise_fill(seed, Adam step of D) per iteration."""
        self.noise_fixed = None if noise is None else np.ascontiguousarray(noise, np.float32).copy()

    noise_fixed = None

    def _d_in(self):
        return [self.input_ctx, self.input_center] if self.opt["conditionAdv"] else self.input_center

    def _g_in(self):
        return [self.input_ctx, self.noise] if self.opt["noiseGen"] else self.input_ctx

    def fDx(self, x):
        o = self.opt
        fs, ov = o["fineSize"], o["overlapPred"]
        zero_conv_biases(self.netD)
        zero_conv_biases(self.netG)
        self.gradParametersD[...] = 0
        real_ctx = self.batch
        lo, hi = fs // 4, fs // 2 + fs // 4                      # 1-based 1+fs/4 .. fs/2+fs/4
        real_center = real_ctx[:, :, lo:hi, lo:hi].copy()
        for ch, mean in enumerate((117.0, 104.0, 123.0)):         # train.lua:287-289
            real_ctx[:, ch, lo + ov:hi - ov, lo + ov:hi - ov] = 2 * mean / 255.0 - 1.0
        self.input_ctx = real_ctx.copy()
        self.input_center = real_center.copy()
        if o["wtl2"] != 0:
            self.input_real_center = real_center.copy()
        B = real_ctx.shape[0]
        label = np.full((B,), 1.0, np.float32)
        output = self.netD.forward(self._d_in())                 # train.lua:300-305
        errD_real = self.criterion.forward(output, label)
        df_do = self.criterion.backward(output, label)
        self.netD.backward(self._d_in(), df_do)
        if o["noiseGen"]:                                        # train.lua:319-323: regenerate random noise
            if self.noise_fixed is not None:
                self.noise = self.noise_fixed
            else:
                self.noise = noise_fill((B, o["nz"], 1, 1), self.noise_seed, self.optimStateD.get("t", 0),
                                        o["noisetype"] == "normal")
        fake = self.netG.forward(self._g_in())
        self.input_center[...] = fake
        label[...] = 0.0
        output = self.netD.forward(self._d_in())
        errD_fake = self.criterion.forward(output, label)
        df_do = self.criterion.backward(output, label)
        self.netD.backward(self._d_in(), df_do)
        self.errD = errD_real + errD_fake
        return self.errD, self.gradParametersD

    def fGx(self, x):
        o = self.opt
        fs, ov, wt = o["fineSize"], o["overlapPred"], o["wtl2"]
        zero_conv_biases(self.netD)
        zero_conv_biases(self.netG)
        self.gradParametersG[...] = 0
        B = self.input_center.shape[0]
        label = np.full((B,), 1.0, np.float32)
        output = self.netD.output                                  # stale w.r.t. D's Adam step (train.lua:363)
        self.errG = self.criterion.forward(output, label)
        df_do = self.criterion.backward(output, label)
        df_dg = self.netD.updateGradInput(self._d_in(), df_do)
        if o["conditionAdv"]:
            df_dg = df_dg[1]                                       # df_dg[2] because conditional GAN (train.lua:371)
        errG_total = self.errG
        if wt != 0:
            self.errG_l2 = self.criterionMSE.forward(self.input_center, self.input_real_center)
            df_dg_l2 = self.criterionMSE.backward(self.input_center, self.input_real_center)
            f32 = np.float32
            if ov == 0:
                if 0 < wt < 1:
                    df_dg *= f32(1 - wt)
                    df_dg += f32(wt) * df_dg_l2
                    errG_total = (1 - wt) * self.errG + wt * self.errG_l2
                else:
                    df_dg += f32(wt) * df_dg_l2
                    errG_total = self.errG + wt * self.errG_l2
            else:
                wtl2Matrix = np.full(df_dg_l2.shape, f32(10 * wt), np.float32)   # train.lua:389-391
                wtl2Matrix[:, :, ov:fs // 2 - ov, ov:fs // 2 - ov] = f32(wt)
                if 0 < wt < 1:
                    df_dg *= f32(1 - wt)
                    df_dg += wtl2Matrix * df_dg_l2
                    errG_total = (1 - wt) * self.errG + wt * self.errG_l2
                else:
                    df_dg += wtl2Matrix * df_dg_l2
                    errG_total = self.errG + wt * self.errG_l2
        self.netG.backward(self._g_in(), df_dg)                   # train.lua:403-407
        return errG_total, self.gradParametersG

    def step(self):
        adam(self.fDx, self.parametersD, self.optimStateD)
        adam(self.fGx, self.parametersG, self.optimStateG)
        return dict(errD=self.errD, errG=self.errG, errG_l2=self.errG_l2)


class VidTrainer:
    """train_vid_weighted.lua (configs 3-4) and, with nc_in/nc_out/widths overridden,
    train_wholeim_input.lua (config 5): full-frame output, D on the whole frame."""

    def __init__(self, opt, rng):
        o = dict(DEFAULT_OPT_VID)
        o.update(opt)
        self.opt = o
        nc = o["nc"] * o["predLen"]
        self.nc_in = o["nc_in"] or nc
        self.nc_out = o["nc_out"] or nc
        # logoNet: train_logo_withmask.lua:95-98 — the last decoder stage is ngf -> ngf/2 -> nc; its closures are this
        # class's with predLen = 1, weight_nomask = 1 (weights of ones), wtgdl = 0
        self.netG = build_netG(self.nc_in, self.nc_out, o["nef"], o["ngf"], o["nBottleneck"], True, o.get("smooth", False),
                               half_last=bool(o.get("logoNet", False)), extra_bottleneck_stage=bool(o.get("ext256", False)))
        self.netD = build_netD(self.nc_out, o["ndf"], True, o.get("smooth", False),
                               extra_last_layer=bool(o.get("ext256", False)))
        weights_init(self.netG, rng)
        weights_init(self.netD, rng)
        self.netI = None                  # withInit: the initializer net (train_vid_weighted.lua:260-264), set by the caller
        self.criterion = BCECriterion()
        self.criterionMSE = MSECriterion() if o["wtl2"] != 0 else None
        self.criterionGDL = GDLCriterion(1) if o["wtgdl"] != 0 else None
        self.optimStateG, self.optimStateD = _solver(o)
        self.parametersD, self.gradParametersD = self.netD.getParameters()
        self.parametersG, self.gradParametersG = self.netG.getParameters()
        self.errD = self.errG = self.errG_l2 = self.errG_gdl = None

    def set_batch(self, real_ctx, real_full, real_mask):
        self.batch = (np.ascontiguousarray(real_ctx, np.float32), np.ascontiguousarray(real_full, np.float32),
                      np.ascontiguousarray(real_mask, np.uint8))

    def fDx(self, x):
        o = self.opt
        zero_conv_biases(self.netD)
        zero_conv_biases(self.netG)
        self.gradParametersD[...] = 0
        real_ctx, real_full, real_mask = self.batch
        self.input_ctx = real_ctx.copy()
        self.input_real = real_full.copy()
        self.input_mask = real_mask.astype(np.float32)            # input_mask:copy(real_mask), Byte -> Float
        if self.netI is not None:                                  # train_vid_weighted.lua:401-405 (netI as loaded:
            fake_init = self.netI.forward(self.input_ctx)          #  training mode, batch statistics)
            sel = self.input_mask != 0                             # inpaint_utils.fillIn, 4-D mask: maskedCopy
            self.input_ctx[sel] = fake_init[sel]
        B = real_ctx.shape[0]
        label = np.full((B,), 1.0, np.float32)
        output = self.netD.forward(self.input_real)
        errD_real = self.criterion.forward(output, label)
        df_do = self.criterion.backward(output, label)
        self.netD.backward(self.input_real, df_do)
        fake = self.netG.forward(self.input_ctx)
        if o["weight_nomask"] == 0:                                # train_vid_weighted.lua:429-432
            self.input_inpainted = np.empty_like(self.input_real)
            lib().vfo_masked_compose(_p(self.input_inpainted), _p(self.input_real), _p(fake), _p(self.input_mask),
                                     _sz(fake.size))
        else:
            self.input_inpainted = fake.copy()
        label[...] = 0.0
        output = self.netD.forward(self.input_inpainted)
        errD_fake = self.criterion.forward(output, label)
        df_do = self.criterion.backward(output, label)
        self.netD.backward(self.input_inpainted, df_do)
        self.errD = errD_real + errD_fake
        return self.errD, self.gradParametersD

    def fGx(self, x):
        o = self.opt
        wt = o["wtl2"]
        f32 = np.float32
        zero_conv_biases(self.netD)
        zero_conv_biases(self.netG)
        self.gradParametersG[...] = 0
        B = self.input_real.shape[0]
        label = np.full((B,), 1.0, np.float32)
        output = self.netD.output
        self.errG = self.criterion.forward(output, label)
        df_do = self.criterion.backward(output, label)
        df_dg = self.netD.updateGradInput(self.input_real, df_do)
        errG_total = self.errG
        if wt != 0:
            if o["weight_nomask"] == 0:
                self.errG_l2 = self.criterionMSE.forward(self.input_inpainted, self.input_real)
                df_dg_l2 = self.criterionMSE.backward(self.input_inpainted, self.input_real)
            else:
                lam = o["weight_nomask"]
                lib().vfo_mask_to_weights(_p(self.input_mask), _f(lam), _sz(self.input_mask.size))  # in place (:494)
                weights = self.input_mask
                self.errG_l2 = self.criterionMSE.forward(self.input_inpainted, self.input_real)
                df_dg_l2 = self.criterionMSE.backward(self.input_inpainted, self.input_real)
                df_dg_l2 *= weights
            assert o["overlapPred"] == 0, "train_vid_weighted.lua:499 'overlapPred should should should be 0'"
            if 0 < wt < 1:
                df_dg *= f32(1 - wt)
                df_dg += f32(wt) * df_dg_l2
                errG_total = (1 - wt) * self.errG + wt * self.errG_l2
            else:
                df_dg += f32(wt) * df_dg_l2
                errG_total = self.errG + wt * self.errG_l2
        if o["wtgdl"] != 0:                                        # train_vid_weighted.lua:523-528 (sic: MSE grad)
            self.errG_gdl = self.criterionGDL.forward(self.input_inpainted, self.input_real)
            df_dg_gdl = self.criterionMSE.backward(self.input_inpainted, self.input_real)
            errG_total = errG_total + o["wtgdl"] * self.errG_gdl
            df_dg += f32(o["wtgdl"]) * df_dg_gdl
        self.df_dg = df_dg
        self.netG.backward(self.input_ctx, df_dg)
        return errG_total, self.gradParametersG

    def step(self):
        adam(self.fDx, self.parametersD, self.optimStateD)
        adam(self.fGx, self.parametersG, self.optimStateG)
        return dict(errD=self.errD, errG=self.errG, errG_l2=self.errG_l2, errG_gdl=self.errG_gdl)


# ------------------------------------------------------------------ synthetic batches
_M64 = (1 << 64) - 1


def _splitmix64(z):
    z = (z + 0x9E3779B97F4A7C15) & _M64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
    return z ^ (z >> 31)


def noise_fill(shape, seed, counter, normal=True):
    """The backend's counter-based stand-in for noise:normal(0,1) / noise:uniform(-1,1) (train.lua:319-323; Torch7's
    generator stream is not reproducible): element i of draw `counter` = splitmix64 of (seed, counter, i), 24-bit
    uniforms, Box-Muller.  include/vf_hip.h: vf_noise_fill."""
    n = int(np.prod(shape))
    base = _splitmix64((seed ^ _splitmix64(counter & 0xFFFFFFFF)) & _M64)
    out = np.empty(n, np.float32)
    scale = np.float32(2.0 ** -24)
    for i in range(n):
        z = _splitmix64((base + i * 0xD1342543DE82EF95) & _M64)
        u1 = np.float32((z >> 40) + 1) * scale
        u2 = np.float32((z >> 16) & 0xFFFFFF) * scale
        if normal:
            out[i] = np.sqrt(np.float32(-2.0) * np.log(u1)) * np.cos(np.float32(6.283185307179586) * u2)
        else:
            out[i] = np.float32(2.0) * u2 - np.float32(1.0)
    return out.reshape(shape)


def synth_center_batch(B, rng, nc=3, fineSize=128):
    return rng.uniform(-1.0, 1.0, (B, nc, fineSize, fineSize)).astype(np.float32)


def synth_vid_batch(B, rng, nc_in, nc_out=None, fineSize=128, maskValue=110.0 / 255.0):
    """(ctx, full, mask) per datavid/dataset.lua:426: centre fineSize/2 square mask replicated over channels;
    ctx = full with masked pixels = 2*maskValue-1 (datavid/donkey_folder.lua:166,183)."""
    nc_out = nc_out or nc_in
    full = rng.uniform(-1.0, 1.0, (B, nc_out, fineSize, fineSize)).astype(np.float32)
    mask = np.zeros((B, nc_out, fineSize, fineSize), np.uint8)
    lo, hi = fineSize // 4, fineSize // 2 + fineSize // 4
    mask[:, :, lo:hi, lo:hi] = 1
    if nc_in == nc_out:
        ctx = full.copy()
        ctx[mask != 0] = np.float32(2 * maskValue - 1)
    else:
        ctx = rng.uniform(-1.0, 1.0, (B, nc_in, fineSize, fineSize)).astype(np.float32)
        ctx[:, :, lo:hi, lo:hi] = np.float32(2 * maskValue - 1)
    return ctx, full, mask


# ------------------------------------------------------------------ batch preparation (loader side of the path)
def center_prepare(batch, overlapPred=0, fill=(117.0, 104.0, 123.0)):
    """train.lua:284-298 on the loader's batch (B x nc x fs x fs in [-1,1]): returns (input_ctx, real_center).
    The centre crop is cloned BEFORE the hole (minus the overlap band) is painted with the channel means."""
    real_ctx = np.ascontiguousarray(batch, np.float32).copy()
    fs = real_ctx.shape[-1]
    lo, hi, ov = fs // 4, fs // 2 + fs // 4, overlapPred
    real_center = real_ctx[:, :, lo:hi, lo:hi].copy()
    for ch, mean in enumerate(fill):
        real_ctx[:, ch, lo + ov:hi - ov, lo + ov:hi - ov] = np.float32(2 * mean / 255.0 - 1.0)
    return real_ctx, real_center


def clip_train_hook(clip, mask, fineSize, w1, h1, flip, maskValue=110.0 / 255.0, blocks=None, blockSize=None):
    """datavid/donkey_folder.lua:135-189 (trainHook, withMask) with the random draws passed in.

    clip: (predLen*nc) x iH x iW in [0,1] (loadContImages' result); mask: 1 x iH x iW Byte 0/1 (:33-35, scaled :103).
    (w1, h1): 0-based crop corner (image.crop(input, w1, h1, w1+oW, h1+oH), :147).  If the cropped mask has a set
    pixel (:165) the clip is maskedFill'ed with maskValue (:166); otherwise randomBlockMask (:114-129) paints `blocks`
    = [(tlx, tly)] (1-based, side blockSize = floor(h/6)) — its mask tensor is torch.Tensor(size) i.e. UNINITIALISED
    outside the blocks in the reference; the evident intent (0) is restated here.  hflip (:178-183) mirrors all three,
    then [0,1] -> [-1,1] (:185-187).  Returns (out, maskout, masked), each (predLen*nc) x fs x fs."""
    fs = fineSize
    out = np.ascontiguousarray(clip[:, h1:h1 + fs, w1:w1 + fs], np.float32).copy()
    maskout = np.broadcast_to(mask[:, h1:h1 + fs, w1:w1 + fs], out.shape).astype(np.uint8).copy()
    masked = out.copy()
    if maskout.max() > 0.5:
        masked[maskout != 0] = np.float32(maskValue)
    else:
        bs = blockSize if blockSize is not None else fs // 6
        maskout = np.zeros(out.shape, np.uint8)
        for tlx, tly in blocks:
            maskout[:, tly - 1:tly - 1 + bs, tlx - 1:tlx - 1 + bs] = 1
            masked[:, tly - 1:tly - 1 + bs, tlx - 1:tlx - 1 + bs] = np.float32(maskValue)
    if flip:
        out, masked, maskout = out[:, :, ::-1].copy(), masked[:, :, ::-1].copy(), maskout[:, :, ::-1].copy()
    out = out * np.float32(2) + np.float32(-1)
    masked = masked * np.float32(2) + np.float32(-1)
    return out, maskout, masked


# ------------------------------------------------------------------ inference (test_vid.lua / test_vid_wholeim.lua)
def whole_image_inpaint(net, fullImages, padmask, predLen, inputLen, fineSize=128, nc=3, netI=None, mid_mask=None):
    """test_vid_wholeim.lua:150-226, tile by tile exactly as the script walks them.

    net (and netI) must already be in evaluate() mode (:62,67).  fullImages: (predLen*nc) x outh x outw in [-1,1],
    padded bottom-right to multiples of fineSize (:137-141); padmask: nc x outh x outw Byte (:209-212).
    Returns (outImages, inpaintImages, fullImages) AFTER the final add(1):mul(0.5) (:222-224), shaped predLen x nc x H x W
    (fullImages stays (predLen*nc) x H x W)."""
    fs = fineSize
    C, outh, outw = fullImages.shape
    ncinput = nc * inputLen
    batch = predLen // inputLen
    assert C == nc * predLen and predLen % inputLen == 0
    outImages = np.zeros((predLen, nc, outh, outw), np.float32)
    for h in range(0, outh, fs):
        for w in range(0, outw, fs):
            flipped = h == 0 and w in (0, fs, 2 * fs)                       # :167 (1-based h==1, w in {1, fs+1, 2fs+1})
            patch = np.zeros((batch, ncinput, fs, fs), np.float32)
            for idx in range(batch):
                tmp = fullImages[idx * ncinput:(idx + 1) * ncinput, h:h + fs, w:w + fs]
                if flipped:
                    tmp = tmp[:, ::-1, :]                                    # image.vflip
                patch[idx] = tmp
            if netI is None:
                out_image = net.forward(patch.copy())
            else:                                                            # :181-190
                mid = netI.forward(patch.copy())
                tmask = mid_mask[:, h:h + fs, w:w + fs]
                filled = patch.copy()
                for b in range(batch):                                       # inpaint_utils.fillIn, 4-D dst / 3-D mask
                    filled[b][tmask != 0] = mid[b][tmask != 0]
                out_image = net.forward(filled)
            out_image = np.array(out_image, np.float32, copy=True)
            if flipped:                                                      # :191-197
                for idx in range(batch):
                    out_image[idx] = out_image[idx][:, ::-1, :]
            outImages[:, :, h:h + fs, w:w + fs] = out_image.reshape(predLen, nc, fs, fs)
    inpaint = np.array(fullImages, np.float32, copy=True).reshape(predLen, nc, outh, outw)
    for i in range(predLen):                                                 # :214-220
        inpaint[i][padmask != 0] = outImages[i][padmask != 0]
    half = np.float32(0.5)
    return (outImages + np.float32(1)) * half, (inpaint + np.float32(1)) * half, (fullImages + np.float32(1)) * half
