"""A CPU stand-in for video_filler_amd.backend.HipBackend, built on the oracle — TESTS ONLY.

It lets the host logic of the package (nn mirror, Sequential fusion plan, flat-parameter layout, trainers'
closures, data-parallel gradient averaging and SyncBN hooks) run without a GPU, e.g. under gloo with
world_size 2.  The product never imports this; `set_backend` is the only way it gets installed.
Tensors are CPU torch tensors with the same logical-NCHW / physical-NHWC convention as on the device.
"""
import ctypes as C

import numpy as np
import torch

from oracle import oracle as O


def _np(t):
    """logical-order contiguous float32 numpy copy"""
    return np.ascontiguousarray(t.detach().contiguous().numpy())


def _put(t, arr):
    t.copy_(torch.from_numpy(np.ascontiguousarray(arr)).reshape(t.shape))


_ACT = {"none": lambda v, s: v, "lrelu": lambda v, s: np.where(v > 0, v, v * np.float32(s)),
        "relu": lambda v, s: np.maximum(v, 0), "tanh": lambda v, s: np.tanh(v),
        "sigmoid": lambda v, s: 1 / (1 + np.exp(-v))}


def _act_grad(y, g, act, s):
    if act == "lrelu":
        return np.where(y > 0, g, g * np.float32(s))
    if act == "relu":
        return np.where(y > 0, g, 0)
    if act == "tanh":
        return g * (1 - y * y)
    if act == "sigmoid":
        return g * (1 - y) * y
    return g


class OracleBackend:
    name = "oracle-cpu (tests only)"

    def __init__(self):
        self.device = torch.device("cpu")
        self.lib = O.lib()

    # plumbing
    def use_current_stream(self):
        pass

    def synchronize(self):
        pass

    def empty(self, *shape, dtype=torch.float32):
        return torch.empty(*shape, dtype=dtype)

    def zeros(self, *shape, dtype=torch.float32):
        return torch.zeros(*shape, dtype=dtype)

    def empty_act(self, B, Cc, H, W):
        return torch.empty((B, H, W, Cc)).permute(0, 3, 1, 2)

    def from_host(self, t):
        return t

    def all_reduce(self, t, group=None):
        import torch.distributed as dist
        dist.all_reduce(t, group=group)

    def all_reduce_avg(self, t, world, group=None, async_op=False):
        """Mean over ranks of a flat gradient bucket.  RCCL averages inside the collective (no extra pass over the
        bucket); other backends sum, then scale.  async_op: returns a handle whose wait() orders the current stream
        after the collective, so kernels launched in between overlap it."""
        import torch.distributed as dist
        if dist.get_backend(group) == "nccl":
            h = dist.all_reduce(t, op=dist.ReduceOp.AVG, group=group, async_op=async_op)
            return h if async_op else None
        dist.all_reduce(t, group=group)
        self.scale_shift(t, 1.0 / world, 0.0)
        return None

    def reduce_scatter_avg(self, t, world, rank, group=None):
        import torch.distributed as dist
        n = t.numel() // world
        assert n * world == t.numel()
        dist.all_reduce(t, group=group)          # (gloo has no reduce-scatter: average everything, hand out the shard)
        self.scale_shift(t, 1.0 / world, 0.0)
        return t[rank * n:(rank + 1) * n]

    def all_gather_shards(self, t, world, rank, group=None, async_op=False):
        import torch.distributed as dist
        n = t.numel() // world
        parts = [t[r * n:(r + 1) * n] for r in range(world)]
        dist.all_gather(parts, parts[rank].clone(), group=group)
        return None

    def copy(self, dst, src):
        dst.copy_(src)

    def zero(self, t):
        t.zero_()

    def zero_segments(self, base, offs, lens):
        for o, n in zip(offs.tolist(), lens.tolist()):
            base[o:o + n] = 0

    # convolutions
    def _conv_mod(self, cls, w, k, s, p, full):
        nIn, nOut = (w.shape[0], w.shape[1]) if full else (w.shape[1], w.shape[0])
        m = cls(nIn, nOut, k, k, s, s, p, p)
        m.weight = _np(w)
        return m

    def conv2d_fwd(self, x, w, bias, y, k, stride, pad, act="none", slope=0.0, full=False):
        m = self._conv_mod(O.SpatialFullConvolution if full else O.SpatialConvolution, w, k, stride, pad, full)
        m.bias = _np(bias) if bias is not None else np.zeros(m.nOutputPlane, np.float32)
        out = m.forward(_np(x))
        _put(y, _ACT[act](out, slope).astype(np.float32))

    def conv2d_bwd_data(self, gy, w, gx, k, stride, pad, full=False):
        m = self._conv_mod(O.SpatialFullConvolution if full else O.SpatialConvolution, w, k, stride, pad, full)
        _put(gx, m.updateGradInput(np.empty(tuple(gx.shape), np.float32), _np(gy)))

    def channel_copy(self, src, c_src, dst, c_dst, ncopy):
        d = _np(dst).copy()
        d[:, c_dst:c_dst + ncopy] = _np(src)[:, c_src:c_src + ncopy]
        _put(dst, d)

    def noise_fill(self, out, seed, counter=0, normal=True, counter_dev=None):
        ctr = int(counter_dev[0]) if counter_dev is not None else counter
        _put(out, O.noise_fill(tuple(out.shape), seed, ctr, normal))

    def bias_grad_multi(self, items):
        for g, gb, beta in items:
            _put(gb, np.float32(beta) * _np(gb) + _np(g).sum(axis=(0, 2, 3), dtype=np.float64).astype(np.float32))

    def conv2d_bwd_data_act(self, gy, w, gx, x_act, act, slope, k, stride, pad):
        self.conv2d_bwd_data(gy, w, gx, k, stride, pad)
        _put(gx, _act_grad(_np(x_act), _np(gx), act, slope))

    def conv2d_bwd_weight(self, x, gy, gw, gb, k, stride, pad, beta, full=False):
        m = self._conv_mod(O.SpatialFullConvolution if full else O.SpatialConvolution, gw, k, stride, pad, full)
        m.gradWeight = np.zeros(tuple(gw.shape), np.float32)
        m.gradBias = np.zeros(m.nOutputPlane, np.float32)
        m.accGradParameters(_np(x), _np(gy))
        _put(gw, np.float32(beta) * _np(gw) + m.gradWeight)
        if gb is not None:
            _put(gb, np.float32(beta) * _np(gb) + m.gradBias)

    def deconv2d_fwd(self, x, w, bias, y, k, stride, pad, act="none", slope=0.0):
        self.conv2d_fwd(x, w, bias, y, k, stride, pad, act, slope, full=True)

    def deconv2d_bwd_data(self, gy, w, gx, k, stride, pad):
        self.conv2d_bwd_data(gy, w, gx, k, stride, pad, full=True)

    def deconv2d_bwd_weight(self, x, gy, gw, gb, k, stride, pad, beta):
        self.conv2d_bwd_weight(x, gy, gw, gb, k, stride, pad, beta, full=True)

    # batch norm, in the HIP backend's two-phase algebra (double accumulators)
    def bn_stats(self, x, shift, sums):
        xn = _np(x).astype(np.float64)
        C_ = xn.shape[1]
        d = xn - _np(shift).astype(np.float64).reshape(1, C_, 1, 1)
        sums[:C_] = torch.from_numpy(d.sum(axis=(0, 2, 3)))
        sums[C_:] = torch.from_numpy((d * d).sum(axis=(0, 2, 3)))

    def bn_finalize(self, sums, rm, rv, save_mean, save_invstd, n_total, momentum, eps):
        C_ = rm.numel()
        s1, s2 = sums[:C_].numpy(), sums[C_:].numpy()
        n = float(n_total)
        shift = rm.numpy().astype(np.float64)
        mean = shift + s1 / n
        m2 = np.maximum(s2 - s1 * s1 / n, 0)
        invstd = 1.0 / np.sqrt(m2 / n + eps)
        save_mean.copy_(torch.from_numpy(mean.astype(np.float32)))
        save_invstd.copy_(torch.from_numpy(invstd.astype(np.float32)))
        rm.copy_(torch.from_numpy((momentum * mean + (1 - momentum) * rm.numpy()).astype(np.float32)))
        with np.errstate(divide="ignore", invalid="ignore"):
            unb = m2 / (n - 1.0)
        rv.copy_(torch.from_numpy((momentum * unb + (1 - momentum) * rv.numpy()).astype(np.float32)))

    def bn_apply(self, x, y, gamma, beta, mean, invstd, act="none", slope=0.0):
        C_ = x.shape[1]
        r = lambda t: _np(t).reshape(1, C_, 1, 1)
        v = ((_np(x) - r(mean)) * r(invstd)) * r(gamma) + r(beta)
        _put(y, _ACT[act](v.astype(np.float32), slope).astype(np.float32))

    def bn_eval_fwd(self, x, y, gamma, beta, rm, rv, eps, act="none", slope=0.0):
        invstd = torch.from_numpy((1.0 / np.sqrt(rv.numpy().astype(np.float64) + eps)).astype(np.float32))
        self.bn_apply(x, y, gamma, beta, rm, invstd, act, slope)

    def bn_bwd_stats(self, x, y_act, gy, save_mean, sums, act="none", slope=0.0):
        C_ = x.shape[1]
        g = _act_grad(_np(y_act), _np(gy), act, slope) if act != "none" else _np(gy)
        g = g.astype(np.float64)
        xc = _np(x).astype(np.float64) - _np(save_mean).astype(np.float64).reshape(1, C_, 1, 1)
        sums[:C_] = torch.from_numpy(g.sum(axis=(0, 2, 3)))
        sums[C_:] = torch.from_numpy((g * xc).sum(axis=(0, 2, 3)))

    def bn_bwd_apply(self, x, y_act, gy, gx, ggamma, gbeta, gamma, save_mean, save_invstd, sums, n_total, act="none",
                     slope=0.0, pbeta=1.0):
        C_ = x.shape[1]
        r = lambda a: np.asarray(a, np.float64).reshape(1, C_, 1, 1)
        s, dp = sums[:C_].numpy(), sums[C_:].numpy()
        inv = _np(save_invstd).astype(np.float64)
        n = float(n_total)
        if gx is not None:
            g = _act_grad(_np(y_act), _np(gy), act, slope) if act != "none" else _np(gy)
            xc = _np(x).astype(np.float64) - r(_np(save_mean))
            k = (dp * inv * inv / n).astype(np.float32)
            out = (g - r(s / n) - xc * r(k)) * r(inv) * r(_np(gamma))
            _put(gx, out.astype(np.float32))
        if ggamma is not None:
            _put(ggamma, np.float32(pbeta) * _np(ggamma) + (dp * inv).astype(np.float32))
        if gbeta is not None:
            _put(gbeta, np.float32(pbeta) * _np(gbeta) + s.astype(np.float32))

    # pointwise
    def act_fwd(self, x, y, act, slope=0.0):
        _put(y, _ACT[act](_np(x), slope).astype(np.float32))

    def act_bwd(self, y, gy, gx, act, slope=0.0):
        _put(gx, _act_grad(_np(y), _np(gy), act, slope).astype(np.float32))

    def axpby(self, a, x, b, y):
        _put(y, np.float32(a) * _np(x) + np.float32(b) * _np(y))

    def cmul(self, x, y):
        _put(y, _np(y) * _np(x))

    def scale_shift(self, y, a, b):
        _put(y, _np(y) * np.float32(a) + np.float32(b))

    # batch preparation / inference tile loop: numpy restatements of the same index maps
    def center_prepare(self, batch_nchw, ctx_out, center_out, fill, overlapPred):
        x = _np(batch_nchw)
        fs = x.shape[-1]
        lo, hi, ov = fs // 4, fs // 2 + fs // 4, overlapPred
        _put(center_out, x[:, :, lo:hi, lo:hi])
        c = x.copy()
        c[:, :, lo + ov:hi - ov, lo + ov:hi - ov] = _np(fill)[None, :, None, None]
        _put(ctx_out, c)

    def clip_prepare(self, clip, mask, full, masked, maskout, w1, h1, flip, mask_value, blocks=None, block_size=0):
        fs = full.shape[-1]
        m = (np.zeros((1,) + tuple(clip.shape[1:]), np.uint8) if mask is None else (_np(mask) != 0).astype(np.uint8)[None])
        o, mo, ma = O.clip_train_hook(_np(clip), m, fs, w1, h1, flip, mask_value, blocks, block_size)
        _put(full, o[None])
        _put(masked, ma[None])
        _put(maskout, mo[None].astype(np.float32))

    def _tile_index(self, Ct, H, W, fs, groups, vflip):
        TX = W // fs
        flips = np.zeros((H // fs) * TX, np.uint8) if vflip is None else _np(vflip.float()).astype(np.uint8)
        nc = Ct // groups
        for t in range((H // fs) * TX):
            ty, tx = divmod(t, TX)
            for g in range(groups):
                yield t * groups + g, slice(g * nc, (g + 1) * nc), slice(ty * fs, (ty + 1) * fs), slice(tx * fs, (tx + 1) * fs), bool(flips[t])

    def tiles_gather(self, full, tiles, groups, vflip=None):
        Ct, H, W = full.shape
        fs = tiles.shape[-1]
        f, out = _np(full), np.zeros(tuple(tiles.shape), np.float32)
        for b, cs, ys, xs, fl in self._tile_index(Ct, H, W, fs, groups, vflip):
            out[b] = f[cs, ys, xs][:, ::-1] if fl else f[cs, ys, xs]
        _put(tiles, out)

    def tiles_scatter(self, tiles, out, groups, vflip=None):
        Ct, H, W = out.shape
        fs = tiles.shape[-1]
        t, o = _np(tiles), np.zeros((Ct, H, W), np.float32)
        for b, cs, ys, xs, fl in self._tile_index(Ct, H, W, fs, groups, vflip):
            o[cs, ys, xs] = t[b][:, ::-1] if fl else t[b]
        _put(out, o)

    def masked_compose(self, out, real, fake, mask):
        _put(out, np.where(_np(mask) != 0, _np(fake), _np(real)))

    # criteria
    def bce_fwd(self, x, label, loss):
        xv = _np(x).reshape(-1)
        t = np.full(xv.shape, label, np.float32)
        loss[0] = self.lib.vfo_bce_fwd(O._p(xv), O._p(t), xv.size)

    def bce_bwd(self, x, label, gx):
        xv = _np(x)
        _put(gx, O.BCECriterion().backward(xv, np.full(xv.reshape(-1).shape, label, np.float32)))

    def mse_fwd(self, x, t, loss):
        loss[0] = O.MSECriterion().forward(_np(x), _np(t))

    def mse_bwd(self, x, t, gx):
        _put(gx, O.MSECriterion().backward(_np(x), _np(t)))

    def recon_grad_mix(self, df_dg, x, t, mask, alpha, c0, c1, band, loss):
        xv, tv = _np(x), _np(t)
        loss[0] = O.MSECriterion().forward(xv, tv)
        g = np.float32(2.0 / xv.size) * (xv - tv)
        if mask is not None:
            w = np.float32(c0) + np.float32(c1) * _np(mask)
        elif band > 0:
            HW = xv.shape[2]
            w = np.full(xv.shape, np.float32(c0 + c1), np.float32)
            w[:, :, band:HW - band, band:HW - band] = np.float32(c0)
        else:
            w = np.float32(c0)
        _put(df_dg, np.float32(alpha) * _np(df_dg) + g * w)

    def gdl_fwd(self, yhat, y, loss):
        loss[0] = O.GDLCriterion(1).forward(_np(yhat), _np(y))

    def gdl_bwd(self, yhat, y, gyhat):
        _put(gyhat, O.GDLCriterion(1).backward(_np(yhat), _np(y)))

    def masked_mse_fwd(self, x, xhat, mask_u8, w, loss):
        c = O.MaskedMSECriterion(w)
        c.setMask(np.ascontiguousarray(mask_u8.contiguous().numpy()))
        loss[0] = c.forward(_np(x), _np(xhat))

    def masked_mse_bwd(self, x, xhat, mask_u8, w, gx):
        c = O.MaskedMSECriterion(w)
        c.setMask(np.ascontiguousarray(mask_u8.contiguous().numpy()))
        _put(gx, c.backward(_np(x), _np(xhat)))

    # the two halves of adam_step, as the HIP backend splits them (optim.adam_update_fused / adam_update_split)
    def adam_prep(self, lr, beta1, beta2, t_dev):
        t_dev[0] += 1
        self._adam_lr = float(lr)

    def adam_apply(self, x, g, m, v, beta1, beta2, eps, t_dev):
        xa, ga, ma, va = x.numpy(), np.ascontiguousarray(g.numpy()), m.numpy(), v.numpy()
        assert xa.flags["C_CONTIGUOUS"] and ma.flags["C_CONTIGUOUS"] and va.flags["C_CONTIGUOUS"]
        self.lib.vfo_adam_step(O._p(xa), O._p(ga), O._p(ma), O._p(va), None, C.c_size_t(xa.size), C.c_double(self._adam_lr),
                               C.c_double(beta1), C.c_double(beta2), C.c_double(eps), int(t_dev[0]))

    def all_gather_ranges(self, t, ranges, rank, group=None, async_op=False):
        """HipBackend.all_gather_ranges over a process group: equal adjacent blocks as one all-gather, a ragged split as one broadcast
        per rank; async_op returns the handles still in flight"""
        import torch.distributed as dist
        world = len(ranges)
        lens = [hi - lo for lo, hi in ranges]
        if len(set(lens)) == 1 and all(ranges[r + 1][0] == ranges[r][1] for r in range(world - 1)):
            self.all_gather_shards(t[ranges[0][0]:ranges[-1][1]], world, rank, group)
            return []
        hs = [dist.broadcast(t[lo:hi], src=r, group=group, async_op=True) for r, (lo, hi) in enumerate(ranges) if hi > lo]
        if async_op:
            return hs
        for h in hs:
            h.wait()
        return []

    # adam: element-wise, so the physical order of x does not matter
    def adam_step(self, x, g, m, v, lr, beta1, beta2, eps, t_dev):
        t_dev[0] += 1
        xa, ga, ma, va = x.numpy(), g.numpy(), m.numpy(), v.numpy()
        self.lib.vfo_adam_step(O._p(xa), O._p(ga), O._p(ma), O._p(va), None, C.c_size_t(xa.size), C.c_double(lr),
                               C.c_double(beta1), C.c_double(beta2), C.c_double(eps), int(t_dev[0]))


# ------------------------------------------------------------------------------------------------ fused-Adam protocol, test double
def install_fused_adam_emulation(min_rows=2):
    """A TEST DOUBLE of cnet.CNet's fused-Adam protocol (set_fused_adam / fused_adam_pack / adam_fused[_gathered] / row ranges) on the
    module-by-module host, so that the trainers' data-parallel HOST LOGIC for it — which slices stay out of the gradient exchange, what
    is gathered instead, the row blocks of the sharded update (ragged ones included), the deferred exchange of the updated rows, the
    marks on the optimiser state — runs under gloo on the CPU.  What travels in a rank's segment here is its LOCAL weight gradient of
    the two bottleneck tensors (the mean over the segments is the global-batch gradient, as the mean of the ranks' operand products is);
    the kernel that forms that gradient from the gathered operands is covered on the GPU (tests/test_gpu_fused_adam.py,
    tests/test_gpu_dp_rehearsal.py).  min_rows: the library deals row blocks of at least 64; the CPU suite's nets have 32 rows."""
    from video_filler_amd import nn
    from video_filler_amd.backend import get_backend
    S = nn.Sequential
    pad4 = lambda n: (n + 3) & ~3

    def _layers(self):
        out = []
        for m, name, gname, o, n in self._flat[2]:
            if name == "weight" and isinstance(m, nn.SpatialConvolution) and m.kH == 4 and m.dH == 1 and m.padH == 0:
                out.append((o, n, m.nInputPlane if m._is_full else m.nOutputPlane))
        return out

    def set_fused_adam(self, on=True):
        self._fa_layers = _layers(self) if on else []
        self._fused_ranges = [(o, o + n) for o, n, _ in self._fa_layers]
        return list(self._fused_ranges)

    def fused_adam_ranges(self):
        return list(getattr(self, "_fused_ranges", []))

    def fused_adam_pack_size(self):
        return sum(pad4(n) for _, n, _ in self._fa_layers)

    def fused_adam_pack(self, segment):
        g, pos = self._flat[1], 0
        for o, n, _ in self._fa_layers:
            segment[pos:pos + n].copy_(g[o:o + n])
            g[o:o + n].zero_()              # (the real kernel never writes these slices)
            pos += pad4(n)

    def _block(Nu, r, world):
        bs = 2 * ((Nu + 2 * world - 1) // (2 * world))
        r0 = min(Nu, r * bs)
        return r0, min(Nu, r0 + bs) - r0

    def fused_adam_rows_ok(self, ranks):
        ls = _layers(self)
        return bool(ls) and all(_block(Nu, ranks - 1, ranks)[1] >= min_rows for _, _, Nu in ls)

    def fused_adam_row_ranges(self, ranks):
        out = []
        for o, n, Nu in self._fa_layers:
            nc = n // Nu
            out.append([(o + _block(Nu, r, ranks)[0] * nc, o + (_block(Nu, r, ranks)[0] + _block(Nu, r, ranks)[1]) * nc) for r in range(ranks)])
        return out

    def _apply(self, grads, m, v, b1, b2, eps, t_dev, keep, rows):
        B = get_backend()
        x, gflat = self._flat[0], self._flat[1]
        for (o, n, Nu), g in zip(self._fa_layers, grads):
            lo, hi = o, o + n
            if rows is not None:
                r0, nr = _block(Nu, rows[0], rows[1])
                nc = n // Nu
                lo, hi = o + r0 * nc, o + (r0 + nr) * nc
            B.adam_apply(x[lo:hi], g[lo - o:hi - o], m[lo:hi], v[lo:hi], b1, b2, eps, t_dev)
            if keep:
                gflat[lo:hi].copy_(g[lo - o:hi - o])

    def adam_fused(self, m, v, b1, b2, eps, t_dev, keep_grad=False):
        g = self._flat[1]
        grads = [g[o:o + n].clone() for o, n, _ in self._fa_layers]
        if not keep_grad:
            for o, n, _ in self._fa_layers:
                g[o:o + n].zero_()
        _apply(self, grads, m, v, b1, b2, eps, t_dev, keep_grad, None)

    def adam_fused_gathered(self, all_segments, world, m, v, b1, b2, eps, t_dev, keep_grad=False, rows=None):
        seg = all_segments.numel() // world
        grads, pos = [], 0
        for o, n, _ in self._fa_layers:
            acc = all_segments[pos:pos + n].clone()
            for r in range(1, world):
                acc += all_segments[r * seg + pos:r * seg + pos + n]
            grads.append(acc * (1.0 / world))
            pos += pad4(n)
        _apply(self, grads, m, v, b1, b2, eps, t_dev, keep_grad, rows)

    for name, fn in dict(set_fused_adam=set_fused_adam, fused_adam_ranges=fused_adam_ranges, fused_adam_pack_size=fused_adam_pack_size,
                         fused_adam_pack=fused_adam_pack, fused_adam_rows_ok=fused_adam_rows_ok, fused_adam_row_ranges=fused_adam_row_ranges,
                         adam_fused=adam_fused, adam_fused_gathered=adam_fused_gathered).items():
        setattr(S, name, fn)
    S.fused_adam_emulated = True
