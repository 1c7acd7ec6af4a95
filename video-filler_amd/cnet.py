"""nn.Sequential hosted by the C-ABI's own net object (include/vf_hip.h `vf_net_*`, csrc/vf_net.hip).

`util.cudnn(net)` is where the reference puts a net on its GPU backend (util.lua:108-131; train.lua:245-258); after that the
drivers only call net:forward / :backward / :updateGradInput / :getParameters and a few sweeps.  `CNet` is the thin host of the
library object that implements exactly that surface with the whole fast path inside the library (activation fusion, BatchNorm
statistics out of the GEMMs, operand planes handed from producer to consumer, grouped weight / bias gradients, netD's 2B batching,
cut backward walks): every method below is one `vf_net_*` call plus pointer plumbing.  It subclasses the module-by-module
mirror `nn.Sequential` only to share its protocol surface (module list, `getParameters()` layout, `reference_flat`, `apply`, ...):
none of the mirror's execution code runs.  `lua/hipnn.lua`'s `hipnn.Net` is the same host written for Torch7.

What stays host-owned, as in Torch7: the flat parameter / gradient storage (torch tensors, bound with vf_net_bind_parameters)
and the BatchNorm running statistics (vf_net_bind_bn_running).  Activations, planes and scratch belong to the library object.
"""
import ctypes as C

import torch

from . import _lib, nn
from .backend import ACT, get_backend, param_version, tensor_from_ptr, to_nhwc

VF_L_CONV, VF_L_FULLCONV, VF_L_BN, VF_L_ACT, VF_L_VIEW = 1, 2, 3, 4, 5


class LayerDesc(C.Structure):       # vf_layer_desc
    _fields_ = [("kind", C.c_int), ("nin", C.c_int), ("nout", C.c_int), ("k", C.c_int), ("stride", C.c_int), ("pad", C.c_int),
                ("act", C.c_int), ("slope", C.c_float), ("eps", C.c_float), ("momentum", C.c_float)]


_OBSERVER = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int64)


def layer_descs(mods):
    """the vf_layer_desc array of a flat module list (what netG:add(...) / netD:add(...) built, train.lua:87-199)"""
    arr = (LayerDesc * len(mods))()
    for i, m in enumerate(mods):
        if isinstance(m, nn.SpatialConvolution):
            arr[i] = LayerDesc(VF_L_FULLCONV if m._is_full else VF_L_CONV, m.nInputPlane, m.nOutputPlane, m.kH, m.dH, m.padH, 0, 0.0, 0.0, 0.0)
        elif isinstance(m, nn.SpatialBatchNormalization):
            arr[i] = LayerDesc(VF_L_BN, 0, m.nOutputPlane, 0, 0, 0, 0, 0.0, m.eps, m.momentum)
        elif isinstance(m, nn._Act):
            arr[i] = LayerDesc(VF_L_ACT, 0, 0, 0, 0, 0, ACT[m.act], float(m.slope), 0.0, 0.0)
        elif isinstance(m, nn.View):
            arr[i] = LayerDesc(VF_L_VIEW, 0, 0, 0, 0, 0, 0, 0.0, 0.0, 0.0)
        else:
            raise TypeError("vf_net hosts chains of convolutions, BatchNorms, activations and views; got %s" % m.type_name())
    return arr


def is_chain(seq):
    """can vf_net host this net?  (the table modules of train.lua's option branches cannot; neither can un-fused or
    memset-zeroing containers, which exist for the mirror's own A/B tests)"""
    ok = (nn.SpatialConvolution, nn.SpatialBatchNormalization, nn._Act, nn.View)
    return isinstance(seq, nn.Sequential) and seq.fuse and all(isinstance(m, ok) for m in seq._plan_items())


class CNet(nn.Sequential):
    _type = "nn.Sequential"

    def __init__(self, fuse=True, lazy_zero=True):
        assert fuse, "vf_net always runs the fused plan"
        super().__init__(True, lazy_zero)
        self._net = None
        self._shape = None
        self._layers = None           # flat module list: index = vf_net layer index
        self._groups = 1
        self._pushed = {}
        self._cb = None
        self._cb_error = None

    @classmethod
    def adopt(cls, seq):
        """a Sequential as build_netG / build_netD made it -> the same module objects hosted by vf_net (before getParameters())"""
        assert is_chain(seq) and seq._flat is None
        net = cls(True, seq.lazy_zero)
        net.modules = seq.modules
        net.train = seq.train
        return net

    def __del__(self):
        try:
            if self._net is not None:
                _lib.load().vf_net_destroy(self._net)
        except Exception:      # noqa: BLE001 — interpreter shutdown
            pass

    # ---- the library object
    def _lib_ctx(self):
        B = get_backend()
        return B.lib, B.ctx

    def _ensure(self, x):
        lib, ctx = self._lib_ctx()
        shape = tuple(x.shape)
        if self._net is None:
            if self._flat is None:
                self.getParameters()
            self._layers = self._plan_items()
            self._index = {id(m): i for i, m in enumerate(self._layers)}
            arr = layer_descs(self._layers)
            h = C.c_void_p()
            Bn, Cc, H, W = shape
            _lib.check(lib.vf_net_create(ctx, C.byref(h), arr, len(arr), Bn, Cc, H, W))
            self._net = h
            flat, gflat, segs = self._flat
            for m, name, gname, o, n in segs:       # one layout on both sides (nn.Sequential.getParameters)
                ln = C.c_int64()
                off = lib.vf_net_param_offset(h, self._index[id(m)], 0 if name == "weight" else 1, C.byref(ln))
                assert off == o and ln.value == n, ("flat layout mismatch", m.type_name(), name, off, o, ln.value, n)
            _lib.check(lib.vf_net_bind_parameters(h, C.c_void_p(flat.data_ptr()), C.c_void_p(gflat.data_ptr()), flat.numel()))
            for i, m in enumerate(self._layers):
                if isinstance(m, nn.SpatialBatchNormalization):
                    _lib.check(lib.vf_net_bind_bn_running(h, i, C.c_void_p(m.running_mean.data_ptr()), C.c_void_p(m.running_var.data_ptr())))
            plan = self._plan or self._build_plan()
            assert lib.vf_net_plan_size(h) == len(plan), "the library's execution plan differs from the mirror's"
            self._shape = shape
            self._pushed = {}
        elif shape != self._shape:
            Bn, Cc, H, W = shape
            _lib.check(lib.vf_net_reshape(self._net, Bn, Cc, H, W))
            self._shape = shape
        self._push("groups", self._groups, lambda v: lib.vf_net_set_batch_groups(self._net, v))
        self._push("train", bool(self.train), lambda v: lib.vf_net_training(self._net, 1 if v else 0))
        self._push("managed", bool(self._wp_managed), lambda v: lib.vf_net_set_weight_planes_managed(self._net, 1 if v else 0))
        _lib.check(lib.vf_net_set_planes_gate(float(nn._PCONV_MIN_GFLOP), int(nn._PCONV_MIN_ROWS)))      # (process-wide on both sides)
        sync = self._sync_state()
        comm = get_backend().comm
        if sync[0] > 1 and comm is None:
            raise RuntimeError("SyncBN over vf_net needs the C-ABI communicator (backend.comm); without it the trainers keep such "
                               "nets on the module-by-module host (trainers._host_nets)")
        # (keyed by the communicator too: one attached after the first forward must reach the net)
        # a forced SyncBN at world 1 (tests, `--force-dist`) takes the communicator only when that communicator IS one rank: the library
        # all-reduces the sums over all of its ranks and divides by npix * world (vf_net_set_sync_bn refuses a mismatch)
        def _sync_comm(v):
            if v[0] > 1:
                return comm
            return comm if (v[1] and comm is not None and lib.vf_comm_world(comm) == v[0]) else None
        self._push("sync", sync + (getattr(comm, "value", None),),
                   lambda v: lib.vf_net_set_sync_bn(self._net, _sync_comm(v), v[0], 1 if v[1] else 0))
        hook = nn.Sequential.act_hook
        if (hook is not None) != (self._cb is not None):
            if hook is not None:
                self._cb = _OBSERVER(self._observe)
                _lib.check(lib.vf_net_set_act_observer(self._net, C.cast(self._cb, C.c_void_p), None))
            else:
                _lib.check(lib.vf_net_set_act_observer(self._net, None, None))
                self._cb = None
        return lib

    def _push(self, key, value, fn):
        if self._pushed.get(key, None) != value:
            _lib.check(fn(value))
            self._pushed[key] = value

    def _sync_state(self):
        bns = [m for m in self.leaves() if isinstance(m, nn.SpatialBatchNormalization)]
        world = max([m.sync_world for m in bns] + [1])
        force = any(m.sync_force for m in bns)
        return (world, force)

    def _observe(self, user, layer, ptr, numel):
        """vf_net_act_observer: hand the (Leaky)ReLU output to nn.Sequential.act_hook as the mirror does"""
        try:
            a = self._layers[layer]
            lib = _lib.load()
            dims = [C.c_int() for _ in range(7)]
            prod = layer - 1 if (layer > 0 and not isinstance(self._layers[layer - 1], (nn._Act, nn.View))) else layer
            _lib.check(lib.vf_net_layer_shape(self._net, prod, *[C.byref(d) for d in dims]))
            Bn, _, _, _, Co, Ho, Wo = [d.value for d in dims]
            assert Bn * Co * Ho * Wo == numel
            y = tensor_from_ptr(ptr, (Bn, Ho, Wo, Co), get_backend().device).permute(0, 3, 1, 2)
            edited = nn.Sequential.act_hook(a, y)
            return 0 if edited is False else 1
        except BaseException as e:     # noqa: BLE001 — must not unwind through the C frames; re-raised by the caller
            self._cb_error = e
            return -1

    def _call(self, fn, *args):
        rc = fn(*args)
        if self._cb_error is not None:
            e, self._cb_error = self._cb_error, None
            raise e
        _lib.check(rc)

    # ---- nn.Module protocol
    def _final_shape(self):
        lib = _lib.load()
        dims = [C.c_int() for _ in range(7)]
        last = max(i for i, m in enumerate(self._layers) if not isinstance(m, (nn.View,)))
        _lib.check(lib.vf_net_layer_shape(self._net, last, *[C.byref(d) for d in dims]))
        Bn, _, _, _, Co, Ho, Wo = [d.value for d in dims]
        return Bn, Co, Ho, Wo

    def updateOutput(self, input, before=None):
        assert before is None, "stream joins in mid-forward belong to the module-by-module mirror"
        assert torch.is_tensor(input), "vf_net hosts chain nets (tensor in, tensor out)"
        x = to_nhwc(input)
        lib = self._ensure(x)
        if self._wp_managed and self._wp_stale():
            self.refresh_weight_planes()
        yp = C.c_void_p()
        self._call(lib.vf_net_forward, self._net, C.c_void_p(x.data_ptr()), C.byref(yp))
        Bn, Co, Ho, Wo = self._final_shape()
        y = tensor_from_ptr(yp.value, (Bn, Ho, Wo, Co), x.device).permute(0, 3, 1, 2)
        tail = self._layers[-1]
        if isinstance(tail, nn.View):
            y = y.reshape(Bn, *tail.sizes)
        self.output = y
        return y

    def _entry_input_shape(self, lo):
        """logical shape of what plan entry `lo` reads (= the gradInput a walk down to `lo` returns)"""
        plan = self._plan or self._build_plan()
        lib = _lib.load()
        dims = [C.c_int() for _ in range(7)]
        _lib.check(lib.vf_net_layer_shape(self._net, self._index[id(plan[lo][0])], *[C.byref(d) for d in dims]))
        Bn, Cc, H, W = [d.value for d in dims[:4]]
        return Bn, Cc, H, W

    # ---- optim.adam inside the bottleneck pair's weight gradients (vf_net_set_fused_adam / vf_net_adam_fused)
    def set_fused_adam(self, on=True):
        """mark the layers whose weight gradient vf_wgrad_adam_outer takes; returns their [(lo, hi)] slices of the flat vectors
        (empty: nothing to fuse).  Takes effect from the next backward pass; needs the library net (a forward has run)."""
        assert self._net is not None, "set_fused_adam after the first forward"
        lib = _lib.load()
        cnt = C.c_int()
        _lib.check(lib.vf_net_set_fused_adam(self._net, 1 if on else 0, C.byref(cnt)))
        self._fused_ranges = []
        for i in range(cnt.value):
            o, n = C.c_int64(), C.c_int64()
            _lib.check(lib.vf_net_fused_adam_range(self._net, i, C.byref(o), C.byref(n)))
            self._fused_ranges.append((o.value, o.value + n.value))
        return list(self._fused_ranges)

    def fused_adam_ranges(self):
        return list(getattr(self, "_fused_ranges", []))

    def adam_fused(self, m, v, beta1, beta2, eps, t_dev, keep_grad=False):
        _lib.check(_lib.load().vf_net_adam_fused(self._net, C.c_void_p(m.data_ptr()), C.c_void_p(v.data_ptr()), beta1, beta2, eps,
                                                 C.c_void_p(t_dev.data_ptr()), 1 if keep_grad else 0))

    def fused_adam_pack_size(self):
        n = C.c_int64()
        _lib.check(_lib.load().vf_net_fused_adam_pack_size(self._net, C.byref(n)))
        return n.value

    def fused_adam_pack(self, segment):
        """this rank's operands of the marked layers' pending weight gradients -> its segment of the gather buffer"""
        assert segment.is_contiguous() and segment.numel() >= self.fused_adam_pack_size()
        _lib.check(_lib.load().vf_net_fused_adam_pack(self._net, C.c_void_p(segment.data_ptr())))

    def adam_fused_gathered(self, all_segments, world, m, v, beta1, beta2, eps, t_dev, keep_grad=False, rows=None):
        """rows = (rank, ranks): this rank forms and applies ITS 1 / ranks of every fused tensor's rows only (the caller all-gathers
        the updated rows: fused_adam_ranges() slices, equal contiguous row blocks)"""
        seg = all_segments.numel() // world
        assert seg * world == all_segments.numel() and seg % 4 == 0
        args = (self._net, C.c_void_p(all_segments.data_ptr()), world, seg, C.c_void_p(m.data_ptr()), C.c_void_p(v.data_ptr()), beta1,
                beta2, eps, C.c_void_p(t_dev.data_ptr()), 1 if keep_grad else 0)
        if rows is None:
            _lib.check(_lib.load().vf_net_adam_fused_gathered(*args))
        else:
            _lib.check(_lib.load().vf_net_adam_fused_gathered_rows(*args, int(rows[0]), int(rows[1])))

    def fused_adam_rows_ok(self, ranks):
        """can the fused tensors' rows be dealt to `ranks` ranks in blocks of at least 64 (even blocks; a shorter last one where the
        row count does not split)?"""
        return self._net is not None and bool(_lib.load().vf_net_fused_adam_rows_ok(self._net, int(ranks)))

    def fused_adam_row_ranges(self, ranks):
        """per marked layer (set_fused_adam(True) first): [(lo, hi)] of the flat vectors, rank by rank — the row block each rank updates
        under adam_fused_gathered(rows=...) and everybody gathers afterwards"""
        lib = _lib.load()
        out = []
        for i in range(len(self.fused_adam_ranges())):
            per = []
            for r in range(ranks):
                o, n = C.c_int64(), C.c_int64()
                _lib.check(lib.vf_net_fused_adam_row_range(self._net, i, r, int(ranks), C.byref(o), C.byref(n)))
                per.append((o.value, o.value + n.value))
            out.append(per)
        return out

    def forward_wait_fused(self, comm, tickets):
        """the next forward waits for these collectives of `comm` (vf_comm_* tickets) in front of its bottleneck conv instead of at its
        start: the layers before it read none of the rows the collectives deliver"""
        for t in tickets:
            _lib.check(_lib.load().vf_net_forward_wait_fused(self._net, comm, int(t)))

    def backward_finish(self):
        if self._net is not None:
            _lib.check(_lib.load().vf_net_backward_finish(self._net))

    def _walk(self, input, gradOutput, want_gp, need_input_grad=True, hi=None, lo=0, group=None, defer_cut=None):
        assert self._net is not None, "backward before forward"
        lib = _lib.load()
        x = to_nhwc(input)
        g = to_nhwc(gradOutput) if gradOutput.dim() == 4 else gradOutput.contiguous()
        if self._wp_managed and self._wp_stale():
            self.refresh_weight_planes()
        gxp = C.c_void_p()
        xp, gp = C.c_void_p(x.data_ptr()), C.c_void_p(g.data_ptr())
        if group is not None:
            assert not want_gp and hi is None and lo == 0
            self._call(lib.vf_net_update_grad_input_group, self._net, xp, gp, group[0], group[1], C.byref(gxp))
        elif want_gp and defer_cut is not None:
            assert hi is None and lo == 0
            self._call(lib.vf_net_backward_split, self._net, xp, gp, int(defer_cut), 1 if need_input_grad else 0, C.byref(gxp))
        elif want_gp:
            self._call(lib.vf_net_backward_range, self._net, xp, gp, -1 if hi is None else hi, lo, 1 if need_input_grad else 0, C.byref(gxp))
        else:
            assert hi is None and lo == 0, "updateGradInput walks the whole net"
            self._call(lib.vf_net_update_grad_input, self._net, xp, gp, C.byref(gxp))
        out = None
        if gxp.value:
            Bn, Cc, H, W = self._entry_input_shape(lo)
            if group is not None:
                Bn //= group[1]
            out = tensor_from_ptr(gxp.value, (Bn, H, W, Cc), x.device).permute(0, 3, 1, 2)
        if lo == 0:
            self.gradInput = out
        return out

    # ---- sweeps and state
    def refresh_weight_planes(self):
        if self._flat is not None:
            self._wp_version = param_version(self._flat[0])
        if self._net is not None:
            _lib.check(_lib.load().vf_net_refresh_weight_planes(self._net))

    def setBatchGroups(self, G):
        super().setBatchGroups(G)
        self._groups = int(G)
        return self

    def zeroConvBiases(self):
        if self._net is None:
            return super().zeroConvBiases()
        _lib.check(_lib.load().vf_net_zero_conv_biases(self._net, None))

    def zeroConvBiasesWith(self, other):
        if self._net is None or not isinstance(other, CNet) or other._net is None:
            self.zeroConvBiases()
            other.zeroConvBiases()
            return
        _lib.check(_lib.load().vf_net_zero_conv_biases(self._net, other._net))

    def zeroGradParameters(self):
        if self.lazy_zero:
            if self._net is not None:
                _lib.check(_lib.load().vf_net_zero_grad(self._net))
        elif self._flat is not None:
            get_backend().zero(self._flat[1])

    def redirect_last_output(self, cur, buf):
        """see nn.Sequential.redirect_last_output: the last convolution writes into `buf` from the next forward on"""
        if self._net is None or self.output is None or cur.data_ptr() != self.output.data_ptr() or tuple(cur.shape) != tuple(buf.shape):
            return False
        last = max(i for i, m in enumerate(self._layers) if isinstance(m, nn.SpatialConvolution))
        _lib.check(_lib.load().vf_net_bind_output(self._net, last, C.c_void_p(buf.data_ptr())))
        return True

    def layer_output(self, i):
        """net.modules[i].output (flat module index), as a logical B x C x H x W view"""
        lib = _lib.load()
        p = C.c_void_p()
        _lib.check(lib.vf_net_layer_output(self._net, i, C.byref(p)))
        dims = [C.c_int() for _ in range(7)]
        j = i
        while j > 0 and isinstance(self._layers[j], (nn._Act, nn.View)):
            j -= 1
        _lib.check(lib.vf_net_layer_shape(self._net, j, *[C.byref(d) for d in dims]))
        Bn, _, _, _, Co, Ho, Wo = [d.value for d in dims]
        return tensor_from_ptr(p.value, (Bn, Ho, Wo, Co), get_backend().device).permute(0, 3, 1, 2)


def adopt_if_chain(seq):
    """util.hip(net) for the C-ABI host: the net object if the library can host this net, else the mirror unchanged"""
    return CNet.adopt(seq) if is_chain(seq) else seq
