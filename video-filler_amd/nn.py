"""Host-side mirror of the Torch7 `nn` surface the reference drivers use, backed by the gfx950 C-ABI.

Same class names, constructor arguments, method names and field names as Torch7 (`forward`, `backward`,
`updateGradInput`, `accGradParameters`, `apply`, `getParameters`, `evaluate`, `.output`, `.gradInput`,
`.weight .bias .gradWeight .gradBias`, `.running_mean .running_var`, `.modules`, `.nInputPlane .nOutputPlane
.kW .kH .dW .dH .padW .padH`) — the protocol listed in SURVEY.md 8(b) — so a driver written against it reads
like train.lua / train_vid_weighted.lua.  Tensors are torch tensors used purely as device memory: logical
B x C x H x W exactly as the reference indexes them, physical channels-last (NHWC); parameters likewise
(see include/vf_hip.h).  All arithmetic happens in libvf_hip.so; there is no CPU path in this package.

Differences from Torch7 that are deliberate and documented in DESIGN.md:
  * `nn.Sequential` fuses an in-place LeakyReLU/ReLU (and a Tanh/Sigmoid that directly follows a convolution)
    into its producer's kernel.  The activated tensor is then the producer's `.output` — for in-place
    activations that is what Torch7 holds as well; for Tanh/Sigmoid the pre-activation tensor is not kept.
    `Sequential(fuse=False)` runs module by module like Torch7.
  * `zeroGradParameters()` is lazy by default: the next accGradParameters overwrites (beta = 0) instead of
    memset + accumulate.  `lazy_zero=False` restores the memset.
  * criteria return a lazy device scalar (`float(x)` synchronises) instead of a Lua number.
"""
import torch

from .backend import bump_param_version, get_backend, is_nhwc, param_version, to_nhwc


# ---------------------------------------------------------------------------------------------- scalars
class DeviceScalar:
    """A loss value that lives on the device until float() is called."""

    def __init__(self, fn):
        self._fn = fn

    @staticmethod
    def of(t):
        return DeviceScalar(lambda: float(t.item()))

    def __float__(self):
        return float(self._fn())

    def _bin(self, other, op):
        a, b = self, other
        return DeviceScalar(lambda: op(float(a), float(b)))

    def __add__(self, o):
        return self._bin(o, lambda x, y: x + y)

    __radd__ = __add__

    def __mul__(self, o):
        return self._bin(o, lambda x, y: x * y)

    __rmul__ = __mul__

    def __repr__(self):
        return "DeviceScalar(%r)" % float(self)


# ---------------------------------------------------------------------------------------------- base
class Module:
    _type = "nn.Module"

    def __init__(self):
        self.output = None
        self.gradInput = None
        self.train = True

    def type_name(self):            # torch.type(m)
        return self._type

    def forward(self, input, **kw):
        return self.updateOutput(input, **kw)

    def backward(self, input, gradOutput, scale=1):
        self.updateGradInput(input, gradOutput)
        self.accGradParameters(input, gradOutput, scale)
        return self.gradInput

    def updateGradInput(self, input, gradOutput):
        raise NotImplementedError

    def accGradParameters(self, input, gradOutput, scale=1):
        pass

    def parameters(self):
        return None

    def apply(self, fn):
        fn(self)
        return self

    def training(self):
        return self.apply(lambda m: setattr(m, "train", True))

    def evaluate(self):
        return self.apply(lambda m: setattr(m, "train", False))

    def cuda(self):
        return self

    def float(self):
        return self

    def _buf(self, name, B, Cc, H, W):
        t = getattr(self, name, None)
        if t is None or tuple(t.shape) != (B, Cc, H, W):
            t = get_backend().empty_act(B, Cc, H, W)
            setattr(self, name, t)
        return t


def _chlast_param(B, d0, d1, kH, kW):
    """logical [d0][d1][kH][kW], physical [d0][kH][kW][d1] (channels-last), zero-initialised."""
    return B.zeros(d0, kH, kW, d1).permute(0, 3, 1, 2)


# ---------------------------------------------------------------------------------------------- convolutions
class SpatialConvolution(Module):
    """nn.SpatialConvolution(nInputPlane, nOutputPlane, kW, kH, dW, dH, padW, padH) — train.lua:89."""
    _type = "nn.SpatialConvolution"
    _is_full = False

    def __init__(self, nInputPlane, nOutputPlane, kW, kH, dW=1, dH=1, padW=0, padH=0):
        super().__init__()
        assert kW == kH and dW == dH and padW == padH, "the reference only builds square kernels/strides"
        self.nInputPlane, self.nOutputPlane = nInputPlane, nOutputPlane
        self.kW, self.kH, self.dW, self.dH, self.padW, self.padH = kW, kH, dW, dH, padW, padH
        B = get_backend()
        d0, d1 = (nInputPlane, nOutputPlane) if self._is_full else (nOutputPlane, nInputPlane)
        self.weight = _chlast_param(B, d0, d1, kH, kW)
        self.gradWeight = _chlast_param(B, d0, d1, kH, kW)
        self.bias = B.zeros(nOutputPlane)
        self.gradBias = B.zeros(nOutputPlane)
        self._fresh = False           # lazy zeroGradParameters: next accumulate overwrites

    def out_hw(self, H, W):
        if self._is_full:
            return ((H - 1) * self.dH - 2 * self.padH + self.kH, (W - 1) * self.dW - 2 * self.padW + self.kW)
        return ((H + 2 * self.padH - self.kH) // self.dH + 1, (W + 2 * self.padW - self.kW) // self.dW + 1)

    # -- operands pre-split into bf16 planes (backend.pconv_*, csrc/vf_pgemm.hip): used in the default product mode wherever
    #    the shape allows; `in_planes` / `g_planes` are the planes of the activation operand when its producer (a BatchNorm)
    #    already wrote them, else the tensor is split here (one pass) — weights come from weight_planes()
    def _pconv_ok(self, Bn, Hgrid, Wgrid, Cgather, Cout, transposed):
        """Where the planes kernels pay.  Their GEMMs are 1.2-1.4x faster than the fp32-operand ones, but the path has costs
        that do not shrink with the batch: the weight planes (a pass over the weights per parameter update, 16 bytes per
        element moved), the planes written beside every activation / gradient that feeds one, and a few more launches per
        layer.  Measured on the same box: with 4.3 GFLOP per pass (train.lua's nets at batchSize 64) the
        iteration is 2.6 % faster with them, with 1.1 GFLOP per pass (the video nets at batchSize 16) 3.5 % slower, and at
        batchSize 4 (train_wholeim_input.lua) 2 % slower.  Threshold: 3 GFLOP per pass and 1024 GEMM rows."""
        B = get_backend()
        rows = Bn * Hgrid * Wgrid // (1 if transposed else 4)
        gflop = 2.0 * rows * 16 * Cgather * Cout * 1e-9
        return (not _NO_PCONV and getattr(B, "mfma_mode", None) in _PLANES_MODES and hasattr(B, "pconv_supported")
                and self.kH == 4 and self.dH == 2 and self.padH == 1 and rows >= _PCONV_MIN_ROWS and gflop >= _PCONV_MIN_GFLOP
                and B.pconv_supported(Bn, Hgrid, Wgrid, Cgather, Cout, 4, 2, 1, transposed))

    def pconv_layer(self):
        """could any pass of this layer use the planes kernels (4x4 stride 2, wide enough)?"""
        return (self.kH == 4 and self.dH == 2 and self.padH == 1 and min(self.nInputPlane, self.nOutputPlane) >= 32
                and self.nInputPlane % 4 == 0 and self.nOutputPlane % 4 == 0)

    def weight_planes(self, transposed, refresh=False):
        """(native, transposed) bf16 planes of the weight; Sequential.refresh_weight_planes() keeps them current in one
        launch per net — a module used on its own refreshes them on every call"""
        B = get_backend()
        if (getattr(self, "_wp", None) is None or self._wp_src != self.weight.data_ptr() or not getattr(self, "_wp_live", False)
                or getattr(self, "_wp_mode", None) != B.mfma_mode):      # (planes of another product mode have another format)
            # first use (or the weight moved, e.g. getParameters): split now; from here on the net's one-launch refresh
            # (Sequential.refresh_weight_planes) keeps this layer's planes current
            self._wp = B.weight_planes(self.weight, *(self._wp if getattr(self, "_wp", None) is not None and self._wp[0].shape[1] == self.weight.numel() else (None, None)))
            self._wp_src = self.weight.data_ptr()
            self._wp_live = True
            self._wp_mode = B.mfma_mode
        elif refresh:
            B.weight_planes(self.weight, *self._wp)
        return self._wp[1 if transposed else 0]

    def _planes_buf(self, name, numel):
        t = getattr(self, name, None)
        if t is None or t.shape[1] != numel:
            t = torch.empty((3, numel), dtype=torch.bfloat16, device=self.weight.device)
            setattr(self, name, t)
        return t

    def updateOutput(self, input, act="none", slope=0.0, in_planes=None, managed=False, want_planes=False):
        """want_planes: also leave the bf16 planes of the output in self.output_planes (for a planes-fed consumer) where this
        pass can write them itself (the 3-channel image-side layers); None otherwise — the consumer then splits."""
        input = to_nhwc(input)
        Bn, Cin, H, W = input.shape
        assert Cin == self.nInputPlane, "expected %d input planes, got %d" % (self.nInputPlane, Cin)
        Ho, Wo = self.out_hw(H, W)
        y = self._buf("output", Bn, self.nOutputPlane, Ho, Wo)
        B = get_backend()
        self.output_planes = None
        if (want_planes and not self._is_full and Cin == 3 and act in ("none", "lrelu", "relu") and hasattr(B, "conv2d_fwd_planes")
                and self.kH == 4 and self.dH == 2 and self.padH == 1 and self.nOutputPlane % 64 == 0 and H % 16 == 0 and W % 16 == 0):
            yp = self._planes_buf("_yp", y.numel())
            B.conv2d_fwd_planes(input, self.weight, self.bias, y, yp, self.kH, self.dH, self.padH, act, slope)
            self.output_planes = yp
            return y
        if act in ("none", "lrelu", "relu") and self._pconv_ok(Bn, H, W, Cin, self.nOutputPlane, self._is_full):
            xp = in_planes if in_planes is not None else B.planes_split(input, self._planes_buf("_xp", input.numel()))
            self._x_planes = (input.data_ptr(), tuple(input.shape), xp)      # accGradParameters reads the same operand
            wp = self.weight_planes(self._is_full, refresh=not managed)
            fn = B.pconv_scatter if self._is_full else B.pconv_gather
            fn(xp, wp, self.bias, y, Bn, H, W, Cin, self.nOutputPlane, act, slope)
            return y
        self._x_planes = None
        fn = B.deconv2d_fwd if self._is_full else B.conv2d_fwd
        fn(input, self.weight, self.bias, y, self.kH, self.dH, self.padH, act, slope)
        return y

    def updateGradInput(self, input, gradOutput, in_act=None, buf="gradInput", g_planes=None, managed=False):
        """in_act = (act, slope): `input` is the in-place activated output of the previous module; its
        updateGradInput (gx .* act'(input)) is applied in this pass's epilogue instead of in a pass of its own.
        buf: the attribute that holds the result (a pass over part of the batch keeps its own buffer)."""
        gx = self._buf(buf, *input.shape)
        B = get_backend()
        go = to_nhwc(gradOutput)
        Bn, Co, Ho, Wo = go.shape
        if self._pconv_ok(Bn, Ho, Wo, Co, self.nInputPlane, not self._is_full) and (in_act is None or not self._is_full):
            gp = g_planes if g_planes is not None else B.planes_split(go, self._planes_buf("_gp" + buf, go.numel()))
            self._g_planes = (go.data_ptr(), tuple(go.shape), gp)
            wp = self.weight_planes(not self._is_full, refresh=not managed)
            if self._is_full:      # full-conv data-gradient: an ordinary strided conv of gradOutput
                B.pconv_gather(gp, wp, None, gx, Bn, Ho, Wo, Co, self.nInputPlane)
            elif in_act is not None:
                B.pconv_scatter(gp, wp, None, gx, Bn, Ho, Wo, Co, self.nInputPlane, dmask=input, dact=in_act[0], dslope=in_act[1])
            else:
                B.pconv_scatter(gp, wp, None, gx, Bn, Ho, Wo, Co, self.nInputPlane)
            return gx
        self._g_planes = None
        if in_act is not None:
            B.conv2d_bwd_data_act(to_nhwc(gradOutput), self.weight, gx, input, in_act[0], in_act[1], self.kH, self.dH, self.padH)
            return gx
        fn = B.deconv2d_bwd_data if self._is_full else B.conv2d_bwd_data
        fn(to_nhwc(gradOutput), self.weight, gx, self.kH, self.dH, self.padH)
        return gx

    def accGradParameters(self, input, gradOutput, scale=1, defer_bias=None, use_planes=True):
        """defer_bias: a list the container flushes at the end of its backward walk — gradBias (the column sums of
        gradOutput) is then computed for all layers in two launches (backend.bias_grad_multi) instead of two each.
        use_planes=False: the caller runs this BEFORE this walk's updateGradInput (side-stream mode), so the remembered
        planes of gradOutput would be the previous iteration's."""
        assert scale == 1, "the reference always uses scale = 1"
        beta = 0.0 if self._fresh else 1.0
        self._fresh = False
        fn = get_backend().deconv2d_bwd_weight if self._is_full else get_backend().conv2d_bwd_weight
        go = to_nhwc(gradOutput)
        gb = self.gradBias
        if defer_bias is not None and self.nOutputPlane % 4 == 0 and go.data_ptr() % 16 == 0:
            defer_bias.append((go, self.gradBias, beta))
            gb = None
        x = to_nhwc(input)
        # the planes both GEMM passes of this layer were fed with (forward: the input's, data-gradient: gradOutput's) are the
        # operands of the weight gradient too: when both are still at hand — same tensors, same shapes — it takes them
        xs, gs = getattr(self, "_x_planes", None), getattr(self, "_g_planes", None)
        self._g_planes = None          # single use: only an updateGradInput of THIS walk may hand its planes over
        if (use_planes and _PWGRAD and not _NO_PCONV and xs is not None and gs is not None and xs[0] == x.data_ptr()
                and xs[1] == tuple(x.shape) and gs[0] == go.data_ptr() and gs[1] == tuple(go.shape)):
            fn(x, go, self.gradWeight, gb, self.kH, self.dH, self.padH, beta, xs[2], gs[2])
        else:
            fn(x, go, self.gradWeight, gb, self.kH, self.dH, self.padH, beta)

    def parameters(self):
        return [self.weight, self.bias], [self.gradWeight, self.gradBias]


class SpatialFullConvolution(SpatialConvolution):
    """nn.SpatialFullConvolution(nInputPlane, nOutputPlane, kW, kH, dW, dH, padW, padH) — train.lua:134."""
    _type = "nn.SpatialFullConvolution"
    _is_full = True


# ---------------------------------------------------------------------------------------------- batch norm
class SpatialBatchNormalization(Module):
    """nn.SpatialBatchNormalization(nOutput, eps=1e-5, momentum=0.1, affine=true) — train.lua:92."""
    _type = "nn.SpatialBatchNormalization"

    def __init__(self, nOutput, eps=1e-5, momentum=0.1, affine=True):
        super().__init__()
        assert affine, "the reference builds affine BN only"
        B = get_backend()
        self.nOutputPlane = nOutput
        self.eps, self.momentum = eps, momentum
        self.weight = B.zeros(nOutput) + 1.0
        self.bias = B.zeros(nOutput)
        self.gradWeight = B.zeros(nOutput)
        self.gradBias = B.zeros(nOutput)
        self.running_mean = B.zeros(nOutput)
        self.running_var = B.zeros(nOutput) + 1.0
        self.save_mean = B.zeros(nOutput)
        self.save_std = B.zeros(nOutput)          # holds 1/sqrt(var+eps), as THNN's save_std does
        self._sums = B.zeros(2 * nOutput, dtype=torch.float64)
        self._fresh = False
        self.sync_world = 1                       # >1: SyncBN — all-reduce the per-channel sums
        self.sync_group = None
        self.sync_force = False                   # take the SyncBN path at world 1 too (tests: collectives inside a capture)
        # groups > 1: the batch is the concatenation of `groups` independent batches (netD's real and fake passes run
        # as one batch of 2B): statistics, running-average updates and backward sums are per group, in group order —
        # exactly what the separate passes would do — while the convolutions around this module see one large batch
        self.groups = 1
        self._gsave = None

    # -- statistics summed by the convolution that produces / consumes the tensor (backend.bn_fuse_next_*): the container
    #    attaches the request to that convolution and tells this module how many partial rows it left (0: none, run the
    #    plain statistics pass)
    def _sync(self):
        return self.sync_world > 1 or self.sync_force

    def fusable(self):
        return self.train and not self._sync() and self.nOutputPlane % 4 == 0 and not _NO_BN_FUSE

    def part_buffer(self, npix):
        """[rows][2C] float64 partials: one row per 64-pixel output tile of the producing GEMM (or per block of its
        split-K combine, at most ~512)"""
        rows = max(npix // 64, 512) + 8 * self.groups
        need = rows * 2 * self.nOutputPlane
        if getattr(self, "_part", None) is None or self._part.numel() < need:
            self._part = get_backend().zeros(need, dtype=torch.float64)
        return self._part

    def _stat_bufs(self):
        if self.groups > 1:
            self._group_state()
            return self._gmean, self._gstd, self._gsums
        return self.save_mean, self.save_std, self._sums

    def updateOutput(self, input, act="none", slope=0.0, pre_rows=0, want_planes=False):
        """want_planes (with pre_rows): also write the three bf16 planes of the output (self.output_planes) for a planes-fed
        convolution behind this module"""
        B = get_backend()
        input = to_nhwc(input)
        Bn, Cc, H, W = input.shape
        assert Cc == self.nOutputPlane
        y = self._buf("output", Bn, Cc, H, W)
        self.output_planes = None
        if pre_rows > 0:
            assert self.fusable() and Bn % self.groups == 0
            sm, ss, su = self._stat_bufs()
            yp = None
            if want_planes:
                yp = getattr(self, "_yp", None)
                if yp is None or yp.shape[1] != y.numel():
                    yp = self._yp = torch.empty((3, y.numel()), dtype=torch.bfloat16, device=y.device)
            B.bn_train_fwd_pre(self._part, pre_rows, input, y, self.weight, self.bias, self.running_mean, self.running_var, sm, ss,
                               su, self.groups, self.momentum, self.eps, act, slope, yp)
            self.output_planes = yp
            return y
        if want_planes and self.train and not self._sync() and hasattr(B, "bn_train_fwd_groups") and Bn % self.groups == 0:
            # the separate statistics pass (no convolution in front could sum them), planes of the output all the same
            sm, ss, su = self._stat_bufs()
            yp = getattr(self, "_yp", None)
            if yp is None or yp.shape[1] != y.numel():
                yp = self._yp = torch.empty((3, y.numel()), dtype=torch.bfloat16, device=y.device)
            B.bn_train_fwd_groups(input, y, self.weight, self.bias, self.running_mean, self.running_var, sm, ss, su, self.groups,
                                  self.momentum, self.eps, act, slope, y_planes=yp)
            self.output_planes = yp
            return y
        if self.train and self.groups > 1:
            assert not self._sync() and Bn % self.groups == 0
            h = Bn // self.groups
            state = self._group_state()
            if hasattr(B, "bn_train_fwd_groups") and not _NO_BN_GROUPS:       # all groups: one launch per stage
                B.bn_train_fwd_groups(input, y, self.weight, self.bias, self.running_mean, self.running_var, self._gmean,
                                      self._gstd, self._gsums, self.groups, self.momentum, self.eps, act, slope)
                return y
            for g, (sm, ss, su) in enumerate(state):
                xg, yg = input[g * h:(g + 1) * h], y[g * h:(g + 1) * h]
                if hasattr(B, "bn_train_fwd"):
                    B.bn_train_fwd(xg, yg, self.weight, self.bias, self.running_mean, self.running_var, sm, ss, su,
                                   self.momentum, self.eps, act, slope)
                else:
                    B.bn_stats(xg, self.running_mean, su)
                    B.bn_finalize(su, self.running_mean, self.running_var, sm, ss, h * H * W, self.momentum, self.eps)
                    B.bn_apply(xg, yg, self.weight, self.bias, sm, ss, act, slope)
            return y
        if self.train and not self._sync() and hasattr(B, "bn_train_fwd"):
            B.bn_train_fwd(input, y, self.weight, self.bias, self.running_mean, self.running_var, self.save_mean,
                           self.save_std, self._sums, self.momentum, self.eps, act, slope)
        elif self.train:
            B.bn_stats(input, self.running_mean, self._sums)
            if self._sync():
                B.all_reduce(self._sums, self.sync_group)
            B.bn_finalize(self._sums, self.running_mean, self.running_var, self.save_mean, self.save_std,
                          Bn * H * W * self.sync_world, self.momentum, self.eps)
            B.bn_apply(input, y, self.weight, self.bias, self.save_mean, self.save_std, act, slope)
        else:
            B.bn_eval_fwd(input, y, self.weight, self.bias, self.running_mean, self.running_var, self.eps, act, slope)
        return y

    def _group_state(self):
        """per-group (save_mean, save_std, sums): rows of one [groups][C] / [groups][2C] allocation each, so that one
        kernel launch can serve every group (vf_bn_*_groups)"""
        if self._gsave is None or len(self._gsave) != self.groups:
            B = get_backend()
            n, G = self.nOutputPlane, self.groups
            self._gmean, self._gstd = B.zeros(G * n), B.zeros(G * n)
            self._gsums = B.zeros(G * 2 * n, dtype=torch.float64)
            self._gsave = [(self._gmean[g * n:(g + 1) * n], self._gstd[g * n:(g + 1) * n],
                            self._gsums[g * 2 * n:(g + 1) * 2 * n]) for g in range(G)]
        return self._gsave

    def _bwd(self, input, gradOutput, want_gx, want_gp, act="none", slope=0.0, y_act=None, group=None, buf="gradInput",
             pre_rows=0, want_planes=False):
        """group = g: `input` / `gradOutput` / `y_act` hold group g's samples only (a pass over one of the concatenated
        batches); group = None with groups > 1: all groups, one after the other.
        pre_rows > 0: gradOutput arrives ALREADY MASKED by the activation's derivative and its sums sit in the partial
        buffer (the data-gradient pass above left both: backend.bn_fuse_next_bwd)."""
        assert self.train, "the reference never back-propagates through BN in evaluate mode"
        B = get_backend()
        input, gradOutput = to_nhwc(input), to_nhwc(gradOutput)
        Bn, Cc, H, W = input.shape
        gx = self._buf(buf, Bn, Cc, H, W) if want_gx else None
        pbeta = 1.0
        if want_gp:
            pbeta = 0.0 if self._fresh else 1.0
            self._fresh = False
        self.grad_planes = None
        if pre_rows > 0:
            assert self.fusable()
            if group is not None:       # a pass over ONE of the concatenated batches: that group's saved statistics
                (sm, ss, su), G = self._group_state()[group], 1
            else:
                (sm, ss, su), G = self._stat_bufs(), self.groups
            gp = None
            if want_planes and gx is not None:
                gp = getattr(self, "_gp_" + buf, None)
                if gp is None or gp.shape[1] != gx.numel():
                    gp = torch.empty((3, gx.numel()), dtype=torch.bfloat16, device=gx.device)
                    setattr(self, "_gp_" + buf, gp)
            B.bn_bwd_pre(self._part, pre_rows, input, gradOutput, gx, self.gradWeight if want_gp else None,
                         self.gradBias if want_gp else None, self.weight, sm, ss, su, G, pbeta, gp)
            self.grad_planes = gp
            return gx
        if want_planes and gx is not None and not self._sync() and hasattr(B, "bn_bwd_groups"):
            # separate statistics pass, gradient planes all the same (the layers right under a bottleneck / under netD's head)
            if group is not None:
                (sm, ss, su), G = self._group_state()[group], 1
            else:
                (sm, ss, su), G = self._stat_bufs(), self.groups
            gp = getattr(self, "_gp_" + buf, None)
            if gp is None or gp.shape[1] != gx.numel():
                gp = torch.empty((3, gx.numel()), dtype=torch.bfloat16, device=gx.device)
                setattr(self, "_gp_" + buf, gp)
            B.bn_bwd_groups(input, y_act, gradOutput, gx, self.gradWeight if want_gp else None, self.gradBias if want_gp else None,
                            self.weight, sm, ss, su, G, act, slope, pbeta, gx_planes=gp)
            self.grad_planes = gp
            return gx
        if self.groups > 1:
            assert not self._sync()
            state = self._group_state()
            gw, gb = (self.gradWeight, self.gradBias) if want_gp else (None, None)

            def one(x, ya, gy, gxx, st, pb):
                sm, ss, su = st
                if hasattr(B, "bn_bwd"):
                    B.bn_bwd(x, ya, gy, gxx, gw, gb, self.weight, sm, ss, su, act, slope, pb)
                else:
                    B.bn_bwd_stats(x, ya, gy, sm, su, act, slope)
                    B.bn_bwd_apply(x, ya, gy, gxx, gw, gb, self.weight, sm, ss, su, x.shape[0] * H * W, act, slope, pb)

            if group is not None:
                one(input, y_act, gradOutput, gx, state[group], pbeta)
                return gx
            h = Bn // self.groups
            if hasattr(B, "bn_bwd_groups") and not _NO_BN_GROUPS:
                B.bn_bwd_groups(input, y_act, gradOutput, gx, gw, gb, self.weight, self._gmean, self._gstd, self._gsums,
                                self.groups, act, slope, pbeta)
                return gx
            for g, st in enumerate(state):
                sl = slice(g * h, (g + 1) * h)
                one(input[sl], None if y_act is None else y_act[sl], gradOutput[sl], None if gx is None else gx[sl], st,
                    pbeta if g == 0 else 1.0)
            return gx
        if not self._sync() and hasattr(B, "bn_bwd"):
            B.bn_bwd(input, y_act, gradOutput, gx, self.gradWeight if want_gp else None, self.gradBias if want_gp else None,
                     self.weight, self.save_mean, self.save_std, self._sums, act, slope, pbeta)
            return gx
        B.bn_bwd_stats(input, y_act, gradOutput, self.save_mean, self._sums, act, slope)
        n_total = Bn * H * W * self.sync_world
        if self._sync():
            # SyncBN: gamma/beta gradients come from THIS rank's sums (the flat-gradient all-reduce adds the other
            # ranks' shares later); gradInput needs the sums over the whole global batch.
            if want_gp:
                B.bn_bwd_apply(input, y_act, gradOutput, None, self.gradWeight, self.gradBias, self.weight,
                               self.save_mean, self.save_std, self._sums, n_total, act, slope, pbeta)
            B.all_reduce(self._sums, self.sync_group)
            if want_gx:
                B.bn_bwd_apply(input, y_act, gradOutput, gx, None, None, self.weight, self.save_mean, self.save_std,
                               self._sums, n_total, act, slope, 1.0)
            return gx
        B.bn_bwd_apply(input, y_act, gradOutput, gx, self.gradWeight if want_gp else None,
                       self.gradBias if want_gp else None, self.weight, self.save_mean, self.save_std, self._sums,
                       n_total, act, slope, pbeta)
        return gx

    def updateGradInput(self, input, gradOutput):
        return self._bwd(input, gradOutput, True, False)

    def accGradParameters(self, input, gradOutput, scale=1):
        assert scale == 1
        self._bwd(input, gradOutput, False, True)

    def backward(self, input, gradOutput, scale=1):
        assert scale == 1
        return self._bwd(input, gradOutput, True, True)

    def parameters(self):
        return [self.weight, self.bias], [self.gradWeight, self.gradBias]


# ---------------------------------------------------------------------------------------------- pointwise
class _Act(Module):
    act = "none"
    slope = 0.0
    inplace = False

    def updateOutput(self, input):
        if self.inplace:
            self.output = input
        elif self.output is None or self.output.shape != input.shape:
            self.output = torch.empty_like(input)
        get_backend().act_fwd(input, self.output, self.act, self.slope)
        return self.output

    def updateGradInput(self, input, gradOutput, y=None):
        # the derivative is evaluated from the ACTIVATED values; for in-place modules `input`
        # already holds them (the producer's .output was overwritten).
        # y: this module's output restricted to the samples of the pass (Sequential._walk, group passes)
        if y is not None and not self.inplace:
            gx = getattr(self, "gradInput_g", None)
            if gx is None or gx.shape != gradOutput.shape:
                gx = self.gradInput_g = torch.empty_like(gradOutput)
            get_backend().act_bwd(y, gradOutput, gx, self.act, self.slope)
            return gx
        y = input if self.inplace else self.output
        if self.inplace:
            self.gradInput = gradOutput
        else:
            self.gradInput = torch.empty_like(gradOutput) if (
                self.gradInput is None or self.gradInput.shape != gradOutput.shape) else self.gradInput
        get_backend().act_bwd(y, gradOutput, self.gradInput, self.act, self.slope)
        return self.gradInput


class LeakyReLU(_Act):
    _type = "nn.LeakyReLU"
    act = "lrelu"

    def __init__(self, negval=0.01, inplace=False):
        super().__init__()
        self.slope, self.inplace = negval, inplace
        self.negval = negval


class ReLU(_Act):
    _type = "nn.ReLU"
    act = "relu"

    def __init__(self, inplace=False):
        super().__init__()
        self.inplace = inplace


class Tanh(_Act):
    _type = "nn.Tanh"
    act = "tanh"


class Sigmoid(_Act):
    _type = "nn.Sigmoid"
    act = "sigmoid"


class View(Module):
    """nn.View(1):setNumInputDims(3) — B x 1 x 1 x 1 -> B x 1 (train.lua:198)."""
    _type = "nn.View"

    def __init__(self, *sizes):
        super().__init__()
        self.sizes = sizes
        self.numInputDims = None

    def setNumInputDims(self, n):
        self.numInputDims = n
        return self

    def updateOutput(self, input):
        self.output = input.reshape(input.shape[0], *self.sizes)
        return self.output

    def updateGradInput(self, input, gradOutput):
        self.gradInput = gradOutput.reshape(input.shape)
        return self.gradInput


# ---------------------------------------------------------------------------------------------- table modules
class JoinTable(Module):
    """nn.JoinTable(2): concatenate a table of [B, C_i, H, W] tensors along the channel dimension (train.lua:119,172).
    Channels are the innermost axis of the device layout, so this is a strided copy per input (vf_channel_copy)."""
    _type = "nn.JoinTable"

    def __init__(self, dimension):
        super().__init__()
        assert dimension == 2, "the reference joins along dimension 2 (channels) only"
        self.dimension = dimension

    def updateOutput(self, input):
        B = get_backend()
        xs = [to_nhwc(x) for x in input]
        Bn, _, H, W = xs[0].shape
        y = self._buf("output", Bn, sum(x.shape[1] for x in xs), H, W)
        off = 0
        for x in xs:
            assert tuple(x.shape[2:]) == (H, W) and x.shape[0] == Bn
            B.channel_copy(x, 0, y, off, x.shape[1])
            off += x.shape[1]
        return y

    def updateGradInput(self, input, gradOutput):
        B = get_backend()
        g = to_nhwc(gradOutput)
        if self.gradInput is None or len(self.gradInput) != len(input):
            self.gradInput = [None] * len(input)
        off = 0
        for i, x in enumerate(input):
            t = self.gradInput[i]
            if t is None or t.shape != x.shape:
                t = self.gradInput[i] = B.empty_act(*x.shape)
            B.channel_copy(g, off, t, 0, x.shape[1])
            off += x.shape[1]
        return self.gradInput


class ParallelTable(Module):
    """nn.ParallelTable: member i maps element i of the input table (train.lua:115-117,168-170).  Members are
    Sequentials; `skip_grad` lists members whose gradInput nobody reads (the context branch of a conditional netD)."""
    _type = "nn.ParallelTable"

    def __init__(self):
        super().__init__()
        self.modules = []
        self.skip_grad = ()

    def add(self, m):
        self.modules.append(m)
        return self

    def apply(self, fn):
        fn(self)
        for m in self.modules:
            m.apply(fn)
        return self

    def leaves(self):
        out = []
        for m in self.modules:
            out += m.leaves() if hasattr(m, "leaves") else [m]
        return out

    def parameters(self):
        ws, gs = [], []
        for m in self.leaves():
            p = m.parameters()
            if p:
                ws += p[0]
                gs += p[1]
        return (ws, gs) if ws else None

    def updateOutput(self, input):
        assert len(input) == len(self.modules)
        self.output = [m.forward(x) for m, x in zip(self.modules, input)]
        return self.output

    def walk(self, input, gradOutput, want_gx, want_gp):
        """updateGradInput and/or accGradParameters of every member in one pass each."""
        gi = []
        for i, (m, x, g) in enumerate(zip(self.modules, input, gradOutput)):
            need = want_gx and i not in self.skip_grad
            if not (need or want_gp):
                gi.append(None)
            elif isinstance(m, Sequential):
                gi.append(m._walk(x, g, want_gp, need))
            else:
                gi.append(m.updateGradInput(x, g) if need else None)
                if want_gp:
                    m.accGradParameters(x, g, 1)
        if want_gx:
            self.gradInput = gi
        return gi

    def updateGradInput(self, input, gradOutput):
        return self.walk(input, gradOutput, True, False)

    def accGradParameters(self, input, gradOutput, scale=1):
        assert scale == 1
        self.walk(input, gradOutput, False, True)

    def backward(self, input, gradOutput, scale=1):
        assert scale == 1
        return self.walk(input, gradOutput, True, True)


# ---------------------------------------------------------------------------------------------- container
_NO_DEFER_BIAS = bool(__import__("os").environ.get("VF_NO_DEFER_BIAS"))      # A/B switches (timing experiments)
_NO_WG_GROUP = bool(__import__("os").environ.get("VF_NO_WG_GROUP"))
_NO_BN_GROUPS = bool(__import__("os").environ.get("VF_NO_BN_GROUPS"))
_NO_BN_FUSE = bool(__import__("os").environ.get("VF_NO_BN_FUSE"))       # BatchNorm statistics from the neighbouring GEMMs
_NO_PCONV = bool(__import__("os").environ.get("VF_NO_PCONV"))           # convolutions from pre-split bf16 planes
_PCONV_MIN_ROWS = int(__import__("os").environ.get("VF_PCONV_MIN_ROWS", "1024"))
PCONV_MIN_GFLOP_SHIPPED = 3.0
_PCONV_MIN_GFLOP = float(__import__("os").environ.get("VF_PCONV_MIN_GFLOP", str(PCONV_MIN_GFLOP_SHIPPED)))
# Weight gradients from the planes too (k_pwgrad_group, single-stage form: 18-23 % faster than k_wgrad_group on its layers).  The
# bottleneck weight gradients of the same walk ride in the same launch in their fp32-fed form (vf_conv.hip's recorder), as
# they did in k_wgrad_group: write-bound tiles under MFMA-bound ones.  Same-box A/B of the iteration: +0.5 .. +1.2 %
# (DESIGN.md 4.7f).  VF_PWGRAD=0 keeps every weight gradient on the fp32-operand kernels.
_PWGRAD = __import__("os").environ.get("VF_PWGRAD", "1") == "1"
# product modes the planes kernels serve: three exact planes (fp32-grade) and ONE plane rounded to bf16 by the producer (the opt-in
# bf16-operand mode; the planes buffers are allocated for three and hold one)
_PLANES_MODES = ("f32_3xbf16", "bf16")


class Sequential(Module):
    _type = "nn.Sequential"
    _group_open = False       # a backward walk has a weight-gradient group open (single-threaded host code)
    # parity-test aid: act_hook(act_module, activated_tensor) runs after every (Leaky)ReLU of a forward pass, fused or not;
    # tests/test_gpu_trainers.py uses it to pin the derivative choice of pre-activations that sit within rounding
    # distance of the kink to the oracle's.  None (always, outside that test): no call, no cost.
    act_hook = None

    def __init__(self, fuse=True, lazy_zero=True):
        super().__init__()
        self.modules = []
        self.fuse = fuse
        self.lazy_zero = lazy_zero
        self._plan = None
        self._flat = None
        self.side = None      # optional side backend (backend.fork()): weight gradients overlap the data-grad chain
        self._act_done_at = -1
        self._bn_pre_at = None
        self._wp_managed = False  # True: the owner (a trainer) calls refresh_weight_planes() after every parameter update
        self._wp_plan = None

    def add(self, m):
        self.modules.append(m)
        self._plan = None
        return self

    def apply(self, fn):
        fn(self)
        for m in self.modules:
            m.apply(fn)
        return self

    def leaves(self):
        out = []
        for m in self.modules:
            out += m.leaves() if isinstance(m, (Sequential, ParallelTable)) else [m]
        return out

    def _plan_items(self):
        """nested Sequentials flattened; a ParallelTable stays one item (its members run their own plans)"""
        out = []
        for m in self.modules:
            out += m._plan_items() if isinstance(m, Sequential) else [m]
        return out

    # -- execution plan: [(main, act_or_None)] over the flattened item list
    def _build_plan(self):
        leaves = self._plan_items()
        plan, i = [], 0
        while i < len(leaves):
            m = leaves[i]
            nxt = leaves[i + 1] if i + 1 < len(leaves) else None
            fusable = False
            if self.fuse and nxt is not None and isinstance(nxt, _Act):
                if isinstance(m, (SpatialConvolution, SpatialBatchNormalization)) and nxt.inplace:
                    fusable = True
                elif isinstance(m, SpatialConvolution) and isinstance(nxt, (Tanh, Sigmoid)):
                    fusable = True
            if fusable:
                plan.append((m, nxt))
                i += 2
            else:
                plan.append((m, None))
                i += 1
        self._plan = plan
        return plan

    def updateOutput(self, input, before=None):
        """before = (plan index, fn): fn() runs right before that plan entry (a stream join in front of the first layer
        whose weights a side stream is still updating)."""
        plan = self._plan or self._build_plan()
        cur = input
        B = get_backend()
        fuse_bn = self.fuse and hasattr(B, "bn_fuse_next_fwd")
        pre_rows = 0
        cur_pl = None             # bf16 planes of `cur`, when its producer wrote them
        managed = self._wp_managed
        if not managed or self._wp_stale():
            self.refresh_weight_planes()
            managed = True
        for idx, (m, a) in enumerate(plan):
            if before is not None and idx == before[0]:
                before[1]()
            nxt = plan[idx + 1][0] if idx + 1 < len(plan) else None
            if (fuse_bn and a is None and isinstance(m, SpatialConvolution) and isinstance(nxt, SpatialBatchNormalization)
                    and nxt.fusable() and nxt.nOutputPlane == m.nOutputPlane):
                # the BatchNorm behind this convolution gets its statistics from the convolution's own epilogue
                Ho, Wo = m.out_hw(cur.shape[2], cur.shape[3])
                B.bn_fuse_next_fwd(nxt.running_mean, nxt.part_buffer(cur.shape[0] * Ho * Wo), nxt.groups)
                cur = m.updateOutput(cur, in_planes=cur_pl, managed=managed)
                cur_pl = None
                pre_rows = B.bn_fuse_result()
                continue
            if isinstance(m, SpatialBatchNormalization):
                rows, pre_rows = pre_rows, 0
                want_pl = (self.fuse and m.train and isinstance(nxt, SpatialConvolution) and nxt.pconv_layer()
                           and (a is None or a.act in ("lrelu", "relu")) and not _NO_PCONV
                           and nxt._pconv_ok(cur.shape[0], cur.shape[2], cur.shape[3], nxt.nInputPlane, nxt.nOutputPlane, nxt._is_full))
                cur = m.updateOutput(cur, *((a.act, a.slope) if a is not None else ("none", 0.0)), pre_rows=rows, want_planes=want_pl)
                cur_pl = m.output_planes
                if a is not None:
                    a.output = cur
                    self._run_act_hook(a, cur, cur_pl)
                continue
            if isinstance(m, SpatialConvolution):
                Ho, Wo = m.out_hw(cur.shape[2], cur.shape[3])
                want_pl = (self.fuse and isinstance(nxt, SpatialConvolution) and not _NO_PCONV and nxt.pconv_layer()
                           and nxt._pconv_ok(cur.shape[0], Ho, Wo, nxt.nInputPlane, nxt.nOutputPlane, nxt._is_full))
                cur = m.updateOutput(cur, *((a.act, a.slope) if a is not None else ("none", 0.0)), in_planes=cur_pl, managed=managed,
                                     want_planes=want_pl)
                cur_pl = m.output_planes
                if a is not None:
                    a.output = cur
                    self._run_act_hook(a, cur, cur_pl)
                continue
            cur_pl = None
            if a is None:
                cur = m.updateOutput(cur)
                if Sequential.act_hook is not None and isinstance(m, _Act) and m.act in ("lrelu", "relu"):
                    Sequential.act_hook(m, cur)
            else:
                cur = m.updateOutput(cur, a.act, a.slope)
                a.output = cur
                if Sequential.act_hook is not None and a.act in ("lrelu", "relu"):
                    Sequential.act_hook(a, cur)
        self.output = cur
        return cur

    @staticmethod
    def _run_act_hook(a, cur, cur_pl):
        """the parity tests' hook on a fused (Leaky)ReLU output.  The producer's planes of `cur` stay the consumer's operand
        (that hand-off is what ships); only if the hook reports that it EDITED the tensor are they re-split from it."""
        if Sequential.act_hook is None or a.act not in ("lrelu", "relu"):
            return
        edited = Sequential.act_hook(a, cur)
        if cur_pl is not None and edited is not False:
            get_backend().planes_split(cur, cur_pl)

    def _walk(self, input, gradOutput, want_gp, need_input_grad=True, hi=None, lo=0, group=None, defer_cut=None):
        """defer_cut = k: one uninterrupted walk whose weight / bias gradients leave in two parts, both at its END: those of the
        plan entries >= k now, the rest at backward_finish() (data parallel: the finished bucket's exchange starts in between).
        Backward over plan entries hi-1 .. lo (default: all of them).  A partial walk lets the caller cut the
        pass where a gradient bucket is complete (data parallel: that bucket's all-reduce then overlaps the rest).
        group = (g, G): the pass covers group g of the G batches the last forward ran concatenated (BatchNorm.groups):
        the saved activations are sliced to that group's samples; `input` / `gradOutput` hold that group only."""
        plan = self._plan or self._build_plan()
        B = get_backend()
        g = gradOutput
        used_side = False
        deferred = [] if (want_gp and self.fuse and hasattr(B, "bias_grad_multi") and not _NO_DEFER_BIAS) else None
        # one stream: the weight-gradient GEMMs of the walk are recorded and launched together at its end
        grouped = (want_gp and self.fuse and self.side is None and hasattr(B, "wgrad_group_begin") and not _NO_WG_GROUP
                   and not Sequential._group_open)       # a member of a ParallelTable records into the outer walk's group
        if grouped:
            B.wgrad_group_begin()
            Sequential._group_open = True
        hi = len(plan) if hi is None else hi
        act_done = self._act_done_at == hi if hi < len(plan) else False
        bn_pre, pre = 0, [0]      # partial rows the data-gradient pass above left for the BatchNorm about to be walked
        g_pl = None               # bf16 planes of `g`, when its producer (a BatchNorm backward) wrote them
        managed = self._wp_managed
        if not managed or self._wp_stale():
            self.refresh_weight_planes()
            managed = True
        if hi < len(plan) and self._bn_pre_at is not None and self._bn_pre_at[0] == hi:
            bn_pre = self._bn_pre_at[1]      # a walk cut between a convolution and the BatchNorm below it resumes here
        self._bn_pre_at = None
        cut_wg, cut_bias = None, 0
        assert not getattr(self, "_split_pending", None) or not want_gp, "a split backward is pending: call backward_finish() first"
        try:
            for idx in range(hi - 1, lo - 1, -1):
                if defer_cut is not None and want_gp and idx == defer_cut - 1:
                    cut_wg = B.wgrad_group_count() if grouped else 0       # everything recorded so far: the finished bucket
                    cut_bias = len(deferred) if deferred is not None else 0
                m, a = plan[idx]
                x = input if idx == 0 else plan[idx - 1][0].output
                mout = m.output
                gbuf = "gradInput"
                if group is not None:
                    gi, G = group
                    gbuf = "gradInput_g"
                    h = mout.shape[0] // G
                    mout = mout[gi * h:(gi + 1) * h]
                    if idx > 0:
                        x = x[gi * h:(gi + 1) * h]
                want_gx = need_input_grad or idx > 0
                if isinstance(m, ParallelTable):
                    assert group is None, "group passes are built for plain chains"
                    g = m.walk(x, g, want_gx, want_gp)
                    g_pl = None
                    act_done = False
                    continue
                if isinstance(m, SpatialBatchNormalization):
                    gsel = None if group is None else group[0]
                    rows, bn_pre = bn_pre, 0
                    # the convolution below consumes this gradient in its data-gradient pass: write its planes here
                    below = plan[idx - 1][0] if idx > 0 else None
                    want_pl = (self.fuse and want_gx and isinstance(below, SpatialConvolution) and below.pconv_layer()
                               and (idx - 1 > 0 or need_input_grad) and not _NO_PCONV
                               and below._pconv_ok(x.shape[0], x.shape[2], x.shape[3], below.nOutputPlane, below.nInputPlane,
                                                   not below._is_full))
                    if a is None:
                        g = m._bwd(x, g, want_gx, want_gp, group=gsel, buf=gbuf, pre_rows=rows, want_planes=want_pl)
                    else:
                        a.gradInput = g
                        g = m._bwd(x, g, want_gx, want_gp, a.act, a.slope, mout, group=gsel, buf=gbuf, pre_rows=rows,
                                   want_planes=want_pl)
                    g_pl = m.grad_planes
                else:
                    if a is not None:
                        if not act_done:
                            if a.inplace:
                                B.act_bwd(mout, g, g, a.act, a.slope)   # in place on the incoming gradient, as Torch7's
                            else:                                       # in-place modules do; Tanh / Sigmoid own theirs
                                bufs = a.__dict__.setdefault("_gfused", {})      # one per shape (full batch / one group)
                                gb_ = bufs.get(tuple(g.shape))
                                if gb_ is None:
                                    gb_ = bufs[tuple(g.shape)] = torch.empty_like(g)
                                B.act_bwd(mout, g, gb_, a.act, a.slope)
                                g = gb_
                            g_pl = None       # (planes a BatchNorm above wrote hold the UNMASKED gradient: conv + ReLU -> BN chains)
                        a.gradInput = g
                    # the module below is a bare conv + in-place (leaky) ReLU: its activation backward rides in this
                    # module's data-gradient epilogue (x IS that activated output)
                    in_act = None
                    if (self.fuse and want_gx and idx > 0 and type(m) is SpatialConvolution and m.dH == 2 and m.kH == 4 and m.padH == 1 and hasattr(B, "conv2d_bwd_data_act")):
                        pm, pa = plan[idx - 1]
                        if pa is not None and pa.act in ("lrelu", "relu") and not isinstance(pm, SpatialBatchNormalization):
                            in_act = (pa.act, pa.slope)
                    # the module below is a BatchNorm (+ activation): this module's data-gradient pass also sums what that
                    # BatchNorm's backward needs, and stores its output masked by the activation's derivative
                    fuse_below = None
                    if (self.fuse and want_gx and idx > 1 and isinstance(m, SpatialConvolution) and hasattr(B, "bn_fuse_next_bwd")):
                        pm, pa = plan[idx - 1]
                        if (isinstance(pm, SpatialBatchNormalization) and pm.fusable() and (pa is None or pa.act in ("lrelu", "relu"))
                                and pm.nOutputPlane == m.nInputPlane and (group is None or pm.groups == group[1])):
                            fuse_below = (pm, pa)
                    if fuse_below is not None:
                        def upd(m=m, x=x, g=g, fb=fuse_below, idx=idx, gpl=g_pl):
                            pm, pa = fb
                            xbn = plan[idx - 2][0].output            # the BatchNorm's input: the convolution below it
                            yact = pm.output if pa is not None else None
                            if group is not None:                    # one of the concatenated batches: its rows, its statistics
                                gi, G = group
                                hb = xbn.shape[0] // G
                                xbn = xbn[gi * hb:(gi + 1) * hb]
                                yact = None if yact is None else yact[gi * hb:(gi + 1) * hb]
                                sm, ng = pm._group_state()[gi][0], 1
                            else:
                                sm, ng = pm._stat_bufs()[0], pm.groups
                            B.bn_fuse_next_bwd(xbn, yact, pa.act if pa is not None else "none", pa.slope if pa is not None else 0.0,
                                               sm, pm.part_buffer(xbn.shape[0] * xbn.shape[2] * xbn.shape[3]), ng)
                            out = m.updateGradInput(x, g, None, gbuf, g_planes=gpl, managed=managed)
                            pre[0] = B.bn_fuse_result()
                            return out
                    elif isinstance(m, SpatialConvolution):
                        upd = lambda: m.updateGradInput(x, g, in_act, gbuf, g_planes=g_pl, managed=managed)
                    elif group is not None and isinstance(m, _Act):
                        upd = lambda: m.updateGradInput(x, g, mout)
                    else:
                        upd = lambda: m.updateGradInput(x, g)
                    if want_gp and self.side is not None and m.parameters():
                        # dW/db only read x and g; nothing on the main stream writes either before the join below
                        with self.side.on():
                            self._acc(m, x, g, deferred, use_planes=False)      # (runs before this walk's upd())
                        used_side = True
                        gin = upd() if want_gx else None
                    else:
                        gin = upd() if want_gx else None
                        if want_gp:
                            self._acc(m, x, g, deferred)
                    g = gin
                    g_pl = None
                    act_done = in_act is not None
                    bn_pre, pre[0] = pre[0], 0
                    continue
                act_done = False
        except BaseException:
            # a raise in mid-walk (shape assert, VF_REQUIRE through _lib.check) must not leave the group open: later walks
            # would skip begin/end while the library kept recording, and no weight gradient would ever be launched again
            if grouped:
                Sequential._group_open = False
                B.wgrad_group_abort()
            raise
        if cut_wg is not None:
            # the finished bucket's gradients now, the rest stay recorded (the group stays open) until backward_finish()
            if grouped:
                B.wgrad_group_end_partial(cut_wg)
            if deferred:
                if deferred[:cut_bias]:
                    B.bias_grad_multi(deferred[:cut_bias])
            self._split_pending = (grouped, deferred[cut_bias:] if deferred else [])
        else:
            if grouped:
                Sequential._group_open = False
                B.wgrad_group_end()
            if deferred:
                B.bias_grad_multi(deferred)      # every deferred gradBias of this walk: two launches
        if used_side:
            self.side.join()
        self._act_done_at = lo if act_done else -1       # a partial walk resumes at `lo` (backward_range)
        self._bn_pre_at = (lo, bn_pre) if (bn_pre and lo > 0) else None
        if lo == 0:
            self.gradInput = g
        return g

    @staticmethod
    def _acc(m, x, g, deferred, use_planes=True):
        if isinstance(m, SpatialConvolution):
            m.accGradParameters(x, g, 1, deferred, use_planes)
        else:
            m.accGradParameters(x, g, 1)

    def refresh_weight_planes(self):
        """The bf16 planes (native + transposed) of every convolution weight the planes kernels can use, in ONE launch.
        A trainer calls this after each optim.adam (set_weight_planes_managed); on its own a Sequential refreshes at the
        start of every forward and backward call, so edits of the weights by any means are always seen."""
        B = get_backend()
        if self._flat is not None:
            self._wp_version = param_version(self._flat[0])
        if _NO_PCONV or getattr(B, "mfma_mode", None) not in _PLANES_MODES or not hasattr(B, "weight_planes_multi"):
            return
        # the layers that have taken the planes path at least once (SpatialConvolution.weight_planes marks them)
        mods = [m for m in self.leaves() if isinstance(m, SpatialConvolution) and getattr(m, "_wp_live", False)
                and m._wp_src == m.weight.data_ptr() and getattr(m, "_wp_mode", None) == B.mfma_mode]
        if not mods:
            return
        key = tuple(m.weight.data_ptr() for m in mods)
        if self._wp_plan is None or self._wp_plan[3] != key:
            self._wp_plan = B.weight_planes_multi([(m.weight, m._wp[0], m._wp[1]) for m in mods])
        B.weight_planes_run(self._wp_plan)

    def _wp_stale(self):
        """managed mode: has anyone (optim.adam_update, load_reference_flat) written the flat parameters since the planes were
        last split?  The owner's explicit refresh points make this False on the hot path; a forward after the last Adam step
        of an iteration (train.lua:437-439's display pass, evaluation, a checkpointed net) finds it True and refreshes."""
        return self._flat is not None and param_version(self._flat[0]) != getattr(self, "_wp_version", -1)

    def set_weight_planes_managed(self, on=True):
        self._wp_managed = bool(on)
        return self

    def redirect_last_output(self, cur, buf):
        """Let the last convolution of this net write its output into `buf` from the next forward on, if `cur` — what a consumer
        was just handed — IS that convolution's output buffer (same storage, same shape): saves the consumer's copy per iteration
        (the generator's image straight into the fake half of netD's [real; fake] batch).  Returns whether it took effect."""
        last = [m for m in self.leaves() if isinstance(m, SpatialConvolution)][-1]
        lo = getattr(last, "output", None)
        if lo is not None and lo.data_ptr() == cur.data_ptr() and tuple(lo.shape) == tuple(buf.shape):
            last.output = buf
            return True
        return False

    def bucket_split(self, frac=0.9):
        """(plan index k, flat offset): the shortest tail plan[k:] of the backward order's head that owns at least
        `frac` of the parameters.  After a walk over plan[k:] the flat gradient [offset, end) is final."""
        plan = self._plan or self._build_plan()
        segs = self._flat[2]
        total = float(sum(n for *_, n in segs))
        first_off = {}
        count = {}
        for m, name, gname, o, n in segs:
            first_off.setdefault(id(m), o)
            count[id(m)] = count.get(id(m), 0) + n
        acc, best = 0, (0, 0)
        for idx in range(len(plan) - 1, -1, -1):
            m = plan[idx][0]
            if id(m) in count:
                acc += count[id(m)]
                best = (idx, first_off[id(m)])
                if acc >= frac * total:
                    break
        return best

    def updateGradInput(self, input, gradOutput, group=None):
        """group = (g, G): see _walk — the data-gradient pass over one of the G concatenated batches."""
        return self._walk(input, gradOutput, False, group=group)

    def setBatchGroups(self, G):
        """The next forwards carry G concatenated, independent batches (BatchNorm statistics per group)."""
        for m in self.leaves():
            if isinstance(m, SpatialBatchNormalization):
                m.groups = G
        return self

    def backward(self, input, gradOutput, scale=1, need_input_grad=True):
        """need_input_grad=False skips the first layer's gradInput — Torch7 always computes it, the reference
        drivers never read it for netD's two full backward passes nor for netG (train.lua:318,348,403)."""
        assert scale == 1
        return self._walk(input, gradOutput, True, need_input_grad)

    def backward_split(self, input, gradOutput, k, need_input_grad=True):
        """backward() whose parameter gradients of the plan entries >= k are complete on return and those below k after
        backward_finish(): the walk itself is not interrupted (see _walk)"""
        return self._walk(input, gradOutput, True, need_input_grad, defer_cut=k)

    def backward_finish(self):
        pend = getattr(self, "_split_pending", None)
        if not pend:
            return
        self._split_pending = None
        grouped, rest = pend
        B = get_backend()
        if grouped:
            Sequential._group_open = False
            B.wgrad_group_end()
        if rest:
            B.bias_grad_multi(rest)

    def backward_range(self, input, gradOutput, hi, lo, need_input_grad=True):
        """backward() restricted to plan entries hi-1 .. lo; gradOutput is what the entry above `hi` returned."""
        return self._walk(input, gradOutput, True, need_input_grad, hi, lo)

    def parameters(self):
        ws, gs = [], []
        for m in self.leaves():
            p = m.parameters()
            if p:
                ws += p[0]
                gs += p[1]
        return ws, gs

    # -- net:getParameters()  
    def getParameters(self, align=64):
        """Flatten {weight, bias} of every module, depth first, into ONE fp32 storage (+ one for grads) and make
        the module tensors views of it.  Inside a tensor the order is the physical channels-last order; segment
        starts are padded to `align` floats so every tensor stays 16-byte aligned.  `reference_flat()` gives
        the exact A.11 vector."""
        B = get_backend()
        owners = [m for m in self.leaves() if m.parameters()]
        segs, off = [], 0
        for m in owners:
            for name, gname in (("weight", "gradWeight"), ("bias", "gradBias")):
                t = getattr(m, name)
                off = (off + align - 1) // align * align
                segs.append((m, name, gname, off, t.numel()))
                off += t.numel()
        total = (off + align - 1) // align * align
        flat, gflat = B.zeros(total), B.zeros(total)
        bias_offs, bias_lens = [], []
        for m, name, gname, o, n in segs:
            t, g = getattr(m, name), getattr(m, gname)
            if t.dim() == 4:
                d0, d1, kH, kW = t.shape
                pv = flat[o:o + n].view(d0, kH, kW, d1)
                pv.copy_(t.permute(0, 2, 3, 1))
                gv = gflat[o:o + n].view(d0, kH, kW, d1)
                gv.copy_(g.permute(0, 2, 3, 1))
                setattr(m, name, pv.permute(0, 3, 1, 2))
                setattr(m, gname, gv.permute(0, 3, 1, 2))
            else:
                flat[o:o + n].copy_(t)
                gflat[o:o + n].copy_(g)
                setattr(m, name, flat[o:o + n])
                setattr(m, gname, gflat[o:o + n])
            if name == "bias" and "Convolution" in m.type_name():
                bias_offs.append(o)
                bias_lens.append(n)
        self._flat = (flat, gflat, segs)
        self._bias_offs = B.from_host(torch.tensor(bias_offs, dtype=torch.int64))
        self._bias_lens = B.from_host(torch.tensor(bias_lens, dtype=torch.int64))
        return flat, gflat

    def reference_flat(self, grads=False):
        """The flat vector exactly as Torch7's getParameters() lays it out (A.11): logical row-major tensors,
        no padding."""
        parts = []
        for m, name, gname, o, n in self._flat[2]:
            t = getattr(m, gname if grads else name)
            parts.append(t.contiguous().reshape(-1))
        return torch.cat(parts)

    def load_reference_flat(self, vec):
        off = 0
        for m, name, gname, o, n in self._flat[2]:
            t = getattr(m, name)
            t.copy_(vec[off:off + n].reshape(t.shape))
            off += n
        assert off == vec.numel()
        bump_param_version(self._flat[0])

    def n_parameters(self):
        return sum(n for *_, n in self._flat[2])

    def zeroConvBiases(self):
        """netX:apply(function(m) if torch.type(m):find('Convolution') then m.bias:zero() end end) — train.lua:279."""
        if self._flat is not None:
            get_backend().zero_segments(self._flat[0], self._bias_offs, self._bias_lens)
        else:
            for m in self.leaves():
                if "Convolution" in m.type_name():
                    get_backend().zero(m.bias)

    def zeroConvBiasesWith(self, other):
        """this net's sweep and `other`'s (both closures zero the conv biases of BOTH nets, train.lua:279-280) in one launch:
        the segment table of the two flat buffers, offsets taken from this net's base"""
        if self._flat is None or other._flat is None or getattr(get_backend(), "name", "") != "hip-gfx950":      # (raw addresses: the HIP backend only)
            self.zeroConvBiases()
            other.zeroConvBiases()
            return
        key = (self._flat[0].data_ptr(), other._flat[0].data_ptr())
        if getattr(self, "_both_key", None) != key:
            delta = (other._flat[0].data_ptr() - self._flat[0].data_ptr()) // 4
            self._both_offs = torch.cat([self._bias_offs, other._bias_offs + delta]).contiguous()
            self._both_lens = torch.cat([self._bias_lens, other._bias_lens]).contiguous()
            self._both_key = key
        get_backend().zero_segments(self._flat[0], self._both_offs, self._both_lens)

    def zeroGradParameters(self):
        """gradParameters:zero() — train.lua:282."""
        if self.lazy_zero:
            for m in self.leaves():
                if m.parameters():
                    m._fresh = True
        elif self._flat is not None:
            get_backend().zero(self._flat[1])
        else:
            for m in self.leaves():
                p = m.parameters()
                if p:
                    for g in p[1]:
                        get_backend().zero(g)


# ---------------------------------------------------------------------------------------------- criteria
class _Criterion:
    def __init__(self):
        # ring of device loss slots: each forward() writes the next one, so a value returned earlier in the
        # iteration (errD_real) is still readable after later evaluations (errD_fake, errG); a captured graph
        # replays into the slots it was captured with.
        self._slots = get_backend().zeros(16, dtype=torch.float64)
        self._i = 0
        self.gradInput = None
        self.output = None

    def next_slot(self):
        t = self._slots[self._i:self._i + 1]
        self._i = (self._i + 1) % self._slots.numel()
        return t

    def cuda(self):
        return self

    def _g(self, like):
        if self.gradInput is None or self.gradInput.shape != like.shape:
            self.gradInput = torch.empty_like(to_nhwc(like) if like.dim() == 4 else like)
        return self.gradInput


class BCECriterion(_Criterion):
    """nn.BCECriterion() — train.lua:204.  `target` is the constant the reference fills `label` with."""

    def forward(self, input, target):
        slot = self.next_slot()
        get_backend().bce_fwd(input, float(target), slot)
        self.output = DeviceScalar.of(slot)
        return self.output

    def backward(self, input, target):
        g = self._g(input)
        get_backend().bce_bwd(input, float(target), g)
        return g

    def forward_backward(self, input, target):
        """criterion:forward(input, target) and criterion:backward(input, target) — the closures always call the pair back to
        back (train.lua:364-366) — as one launch where the backend has it; returns (loss, gradInput)"""
        B = get_backend()
        if not hasattr(B, "bce_fwd_bwd") or not input.is_contiguous():
            return self.forward(input, target), self.backward(input, target)
        slot, g = self.next_slot(), self._g(input)
        B.bce_fwd_bwd(input, float(target), float(target), input.numel(), 1, slot, None, g)
        self.output = DeviceScalar.of(slot)
        return self.output, g


class MSECriterion(_Criterion):
    """nn.MSECriterion() — train.lua:207."""

    def forward(self, input, target):
        slot = self.next_slot()
        get_backend().mse_fwd(to_nhwc(input), to_nhwc(target), slot)
        self.output = DeviceScalar.of(slot)
        return self.output

    def backward(self, input, target):
        g = self._g(input)
        get_backend().mse_bwd(to_nhwc(input), to_nhwc(target), g)
        return g


class GDLCriterion(_Criterion):
    """nn.GDLCriterion(alpha) — gdl_criterion.lua:6; the drivers consume only :forward (train_vid_weighted.lua:524);
    :backward (gdl_criterion.lua:47-53) completes the nn.Criterion protocol."""

    def __init__(self, alpha=1):
        super().__init__()
        assert alpha == 1  # gdl_criterion.lua:9

    def forward(self, input, target):
        slot = self.next_slot()
        get_backend().gdl_fwd(to_nhwc(input), to_nhwc(target), slot)
        self.output = DeviceScalar.of(slot)
        return self.output

    def backward(self, input, target):
        B = get_backend()
        x = to_nhwc(input)
        if self.gradInput is None or tuple(self.gradInput.shape) != tuple(x.shape):
            self.gradInput = B.empty_act(*x.shape)
        B.gdl_bwd(x, to_nhwc(target), self.gradInput)
        return self.gradInput

    updateGradInput = backward


class MaskedMSECriterion(_Criterion):
    """nn.MaskedMSECriterion(mWeight) — MaskedMSECriterion.lua:7."""

    def __init__(self, mWeight=1.0):
        super().__init__()
        self.mWeight = mWeight
        self.mask = None

    def setMask(self, m):
        assert m.dtype == torch.uint8, "setMask wants a ByteTensor (MaskedMSECriterion.lua:25)"
        self.mask = m if m.dim() != 4 else to_nhwc(m)

    def forward(self, input, target):
        slot = self.next_slot()
        get_backend().masked_mse_fwd(to_nhwc(input), to_nhwc(target), self.mask, self.mWeight, slot)
        self.output = DeviceScalar.of(slot)
        return self.output

    def backward(self, input, target):
        g = self._g(input)
        get_backend().masked_mse_bwd(to_nhwc(input), to_nhwc(target), self.mask, self.mWeight, g)
        return g
