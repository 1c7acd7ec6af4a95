"""Tensor-level backend used by the nn mirror: torch tensors in, raw device pointers out to the C-ABI.

PyTorch is plumbing here (device memory, streams, torch.distributed); every FLOP of the hot path runs in
libvf_hip.so.  `HipBackend` is the only backend the package ships.  Tests may install another object with
the same methods through `set_backend` to exercise host logic on CPU (tests/oracle_backend.py); the package
itself never constructs anything else and never falls back.
"""
import ctypes as C
import os

import torch

from . import _lib

ACT = {"none": 0, "lrelu": 1, "relu": 2, "tanh": 3, "sigmoid": 4}
# struct VfColsumDesc of csrc/vf_bn.hip (64 bytes)
COLSUM_DESC = [("g", "<u8"), ("gb", "<u8"), ("part", "<u8"), ("P", "<i8"), ("C", "<i4"), ("cq", "<i4"), ("rows_per_block", "<i4"),
               ("gx", "<i4"), ("gy", "<i4"), ("blk1_off", "<i4"), ("blk2_off", "<i4"), ("beta", "<f4")]
# struct VfWpDesc of csrc/vf_pgemm.hip (48 bytes)
WPLANES_DESC = [("w", "<u8"), ("nat", "<u8"), ("tr", "<u8"), ("d0", "<i4"), ("d1", "<i4"), ("gx", "<i4"), ("gz", "<i4"),
                ("blk_off", "<i4"), ("pad", "<i4")]
MFMA_MODES = {"f32": 0, "bf16": 1, "f32_3xbf16": 3}
DEFAULT_MFMA_MODE = "f32_3xbf16"


_COMM_DTYPE = {torch.float32: 0, torch.float64: 1}
_COMM_OP = {"sum": 0, "avg": 1, "max": 2, "min": 3}


class _Range:
    def __init__(self, lib, name):
        self.lib, self.name = lib, name.encode()

    def __enter__(self):
        self.lib.vf_range_push(self.name)
        return self

    def __exit__(self, *exc):
        self.lib.vf_range_pop()
        return False


class _CommHandle:
    """one collective in flight on the communicator's stream (vf_comm_allreduce_async's ticket)"""

    def __init__(self, backend, ticket):
        self.backend, self.ticket = backend, ticket

    def wait(self):
        b = self.backend
        _lib.check(b.lib.vf_comm_wait(b.comm, b.ctx, self.ticket))


def exchange_comm_id(backend, world, rank, addr=None, port=None, timeout_s=300):
    """rank 0's vf_comm_unique_id to every rank through a TCP key-value store at MASTER_ADDR:MASTER_PORT (what
    torch.distributed.run exports).  The store is the only piece of torch.distributed involved: no process group."""
    import datetime
    from torch.distributed import TCPStore
    addr = addr or os.environ.get("MASTER_ADDR", "127.0.0.1")
    port = int(port or os.environ["MASTER_PORT"])
    # under torch.distributed.run the launcher's agent already serves a store on that port (TORCHELASTIC_USE_AGENT_STORE):
    # every rank is then a client of it; otherwise rank 0 hosts the store
    agent = os.environ.get("TORCHELASTIC_USE_AGENT_STORE", "") == "True"
    store = TCPStore(addr, port, world, (rank == 0) and not agent, timeout=datetime.timedelta(seconds=timeout_s))
    if agent:
        from torch.distributed import PrefixStore
        store = PrefixStore("vf_comm/%s" % os.environ.get("TORCHELASTIC_RESTART_COUNT", "0"), store)
    if rank == 0:
        store.set("vf_comm_id", backend.comm_unique_id())
    return bytes(store.get("vf_comm_id")), store


def bring_up_comm(backend, world, rank, addr=None, port=None, verify=True):
    """Collective bring-up of the C-ABI exchange (vf_comm_*): EVERY rank ends up on the same path.

    Returns (ok, store).  ok = True: `backend.comm` is a verified communicator on all ranks.  ok = False: no rank holds
    one (any that had been created is destroyed) and the host runs the exchange through torch.distributed instead.
    The ranks agree through the TCP store the launcher serves, never through a collective that a failed rank would miss:
      1. every rank checks locally that RCCL can be bound (vf_comm_available) and publishes the result; all read all;
      2. only if every rank can: rank 0's id goes round, vf_comm_init (collective);
      3. verify: an all-reduce (sum, average), a reduce-scatter + all-gather and a broadcast of known vectors are compared
         with their closed-form results on every rank; the verdicts go round the same way."""
    import datetime
    from torch.distributed import TCPStore
    addr = addr or os.environ.get("MASTER_ADDR", "127.0.0.1")
    port = int(port or os.environ["MASTER_PORT"])
    agent = os.environ.get("TORCHELASTIC_USE_AGENT_STORE", "") == "True"
    store = TCPStore(addr, port, world, (rank == 0) and not agent, timeout=datetime.timedelta(seconds=300))
    if agent:
        from torch.distributed import PrefixStore
        store = PrefixStore("vf_comm/%s" % os.environ.get("TORCHELASTIC_RESTART_COUNT", "0"), store)

    def agree(tag, ok):
        store.set("%s/%d" % (tag, rank), "1" if ok else "0")
        return all(bytes(store.get("%s/%d" % (tag, r))) == b"1" for r in range(world))

    local_ok = backend.lib.vf_comm_available() == 0
    if not agree("vf_comm_avail", local_ok):
        return False, store
    if rank == 0:
        store.set("vf_comm_id", backend.comm_unique_id())
    cid = bytes(store.get("vf_comm_id"))
    try:
        backend.init_comm(world, rank, cid)
        init_ok = True
    except Exception as e:      # noqa: BLE001 — reported, and decided on by all ranks together below
        import sys
        sys.stderr.write("video-filler_amd: vf_comm_init failed on rank %d (%s: %s)\n" % (rank, type(e).__name__, e))
        init_ok = False
    if not agree("vf_comm_init", init_ok):
        backend.destroy_comm()
        return False, store
    ver_ok = True
    if verify:
        try:
            ver_ok = verify_comm(backend, world, rank)
        except Exception as e:  # noqa: BLE001
            import sys
            sys.stderr.write("video-filler_amd: vf_comm self-check raised on rank %d (%s: %s)\n" % (rank, type(e).__name__, e))
            ver_ok = False
    if not agree("vf_comm_verify", ver_ok):
        backend.destroy_comm()
        return False, store
    return True, store


def verify_comm(backend, world, rank, n=4096):
    """closed-form checks of every collective the iteration uses, on this rank's communicator (bring_up_comm, tests)"""
    dev = backend.device
    base = torch.arange(n, dtype=torch.float32, device=dev) * 0.25 + 1.0
    # all-reduce average of (rank + 1) * base  ->  base * (world + 1) / 2 ; async form with a wait
    t = base * float(rank + 1)
    backend.all_reduce_avg(t, world)
    ok = bool(torch.allclose(t, base * ((world + 1) / 2.0), rtol=1e-6, atol=0))
    # inline sum in double (SyncBN's sums)
    d = torch.full((64,), float(rank + 1), dtype=torch.float64, device=dev)
    backend.all_reduce(d)
    ok = ok and bool((d == world * (world + 1) / 2.0).all())
    # reduce-scatter (mean) + all-gather: every shard of the result is the mean over ranks
    m = (n // world) * world
    t = (base[:m] * float(rank + 1)).contiguous()
    shard = backend.reduce_scatter_avg(t, world, rank)
    k = m // world
    ok = ok and bool(torch.allclose(shard, base[rank * k:(rank + 1) * k] * ((world + 1) / 2.0), rtol=1e-6, atol=0))
    t[rank * k:(rank + 1) * k] = float(rank)            # every rank marks its own shard, all-gather spreads the marks
    backend.all_gather_shards(t, world, rank)
    want = torch.arange(world, dtype=torch.float32, device=dev).repeat_interleave(k)
    ok = ok and bool((t == want).all())
    # broadcast from rank 0
    b = torch.full((256,), float(rank + 7), dtype=torch.float32, device=dev)
    backend.comm_broadcast(b, 0)
    ok = ok and bool((b == 7.0).all())
    torch.cuda.synchronize(dev)
    return ok


# ---- parameter versions: who moved a flat parameter vector last.  optim.adam_update / load_reference_flat bump the
#      version of the storage they wrote; a net whose weight planes (bf16 shadows of its conv weights, vf_pgemm.hip) were
#      split from an older version refreshes them before its next forward / backward, whoever calls it.
_PARAM_VERSION = {}


def _storage_key(t):
    return t.untyped_storage().data_ptr()


def bump_param_version(t):
    k = _storage_key(t)
    _PARAM_VERSION[k] = _PARAM_VERSION.get(k, 0) + 1


def param_version(t):
    return _PARAM_VERSION.get(_storage_key(t), 0)


class _DevPtr:
    """a device allocation of the library as a __cuda_array_interface__ object (what torch.as_tensor adopts without a copy)"""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": tuple(int(v) for v in shape), "typestr": typestr, "data": (int(ptr), False),
                                         "version": 2, "strides": None}


def tensor_from_ptr(ptr, shape, device, typestr="<f4"):
    """a torch view (no copy, no ownership) of library-owned device memory — plumbing for hosts of vf_net_*"""
    return torch.as_tensor(_DevPtr(ptr, shape, typestr), device=device)


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def is_nhwc(t):
    """True if a logical B x C x H x W tensor is physically [B][H][W][C] and dense."""
    return t.dim() == 4 and t.permute(0, 2, 3, 1).is_contiguous()


def nhwc_empty(B, Cc, H, W, device, dtype=torch.float32):
    return torch.empty((B, H, W, Cc), device=device, dtype=dtype).permute(0, 3, 1, 2)


def to_nhwc(t):
    return t if is_nhwc(t) else t.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)


class HipBackend:
    name = "hip-gfx950"

    def __init__(self, device=None, workspace_bytes=None):
        if not torch.cuda.is_available():
            raise RuntimeError("video-filler_amd: no MI355X visible (torch.cuda.is_available() is False); "
                               "the HIP backend has no CPU fallback")
        self.lib = _lib.load()
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None else device)
        ctx = C.c_void_p()
        _lib.check(self.lib.vf_ctx_create(C.byref(ctx), self.device.index, None))
        self.ctx = ctx
        nbytes = workspace_bytes or self.lib.vf_workspace_bytes_hint()
        self.workspace = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        _lib.check(self.lib.vf_ctx_set_workspace(self.ctx, _ptr(self.workspace), nbytes))
        self.use_current_stream()
        self._forks = []
        self.mfma_mode = DEFAULT_MFMA_MODE       # what vf_ctx_create starts in
        mode = os.environ.get("VF_MFMA_MODE", DEFAULT_MFMA_MODE)
        if mode != self.mfma_mode:
            self.set_mfma_mode(mode)

    def set_mfma_mode(self, mode):
        """How the conv / full-conv products are formed (vf_ctx_set_mfma_mode):
          'f32_3xbf16' (default) fp32 operands split EXACTLY into three bf16 planes, six cross terms on the bf16 matrix
                       pipe, fp32 accumulation — fp32-grade results (same tolerances as 'f32'), 1.1-1.4x faster;
          'f32'        native v_mfma_f32_32x32x2_f32;
          'bf16'       operands ROUNDED to bf16 (opt-in, 1e-2 tolerance).
        Applies to this backend and the side backends forked from it."""
        code = MFMA_MODES[mode]
        for b in [self] + list(self._forks):
            _lib.check(self.lib.vf_ctx_set_mfma_mode(b.ctx, code))
            b.mfma_mode = mode

    # ---- plumbing
    def use_current_stream(self):
        s = torch.cuda.current_stream(self.device).cuda_stream
        _lib.check(self.lib.vf_ctx_set_stream(self.ctx, C.c_void_p(s)))

    def fork(self, workspace_bytes=128 << 20):
        """A second context on its OWN stream with its OWN workspace, for work that may overlap the main stream
        (weight gradients beside the data-gradient chain; netG forward beside netD's real pass).  Use it through
        `with side.on(): ...` which orders it after everything issued so far on the current stream; the caller
        joins with `side.join()`."""
        side = HipBackend.__new__(HipBackend)
        side.lib, side.device = self.lib, self.device
        ctx = C.c_void_p()
        _lib.check(self.lib.vf_ctx_create(C.byref(ctx), self.device.index, None))
        side.ctx = ctx
        side.workspace = torch.empty(workspace_bytes, dtype=torch.uint8, device=self.device)
        _lib.check(self.lib.vf_ctx_set_workspace(side.ctx, _ptr(side.workspace), workspace_bytes))
        side.stream = torch.cuda.Stream(device=self.device)
        _lib.check(self.lib.vf_ctx_set_stream(side.ctx, C.c_void_p(side.stream.cuda_stream)))
        side.parent = self
        side.comm = self.comm
        side._forks = []
        side.mfma_mode = self.mfma_mode
        _lib.check(self.lib.vf_ctx_set_mfma_mode(side.ctx, MFMA_MODES[self.mfma_mode]))
        self._forks.append(side)
        return side

    def on(self):
        """context manager: make this (side) backend current, on its stream, after the current stream's work"""
        return _SideScope(self)

    def join(self):
        """the current stream waits for everything issued on this side stream"""
        torch.cuda.current_stream(self.device).wait_stream(self.stream)

    def empty(self, *shape, dtype=torch.float32):
        return torch.empty(*shape, dtype=dtype, device=self.device)

    def zeros(self, *shape, dtype=torch.float32):
        return torch.zeros(*shape, dtype=dtype, device=self.device)

    def empty_act(self, B, Cc, H, W):
        return nhwc_empty(B, Cc, H, W, self.device)

    def from_host(self, t):
        return t.to(self.device)

    # ---- the data-parallel exchange.  With a communicator attached (init_comm: vf_comm_* of the C-ABI, RCCL underneath)
    #      nothing here touches torch.distributed; without one — tests that rehearse the control flow over gloo, or a host
    #      that already owns a torch process group — the same calls go through torch.distributed.
    comm = None

    def init_comm(self, world, rank, id_bytes):
        """vf_comm_init on this backend's device (collective over all ranks).  id_bytes: rank 0's `comm_unique_id()`,
        moved to every rank by the host (`exchange_comm_id` does it through a TCP store under torchrun's MASTER_ADDR/PORT)."""
        assert self.comm is None and getattr(self, "parent", None) is None, "one communicator per process, on the main backend"
        assert len(id_bytes) == 128
        torch.cuda.set_device(self.device)
        h = C.c_void_p()
        buf = C.create_string_buffer(bytes(id_bytes), 128)
        _lib.check(self.lib.vf_comm_init(C.byref(h), buf, world, rank))
        self.comm = h
        for b in self._forks:
            b.comm = h
        return self

    def comm_unique_id(self):
        buf = C.create_string_buffer(128)
        _lib.check(self.lib.vf_comm_unique_id(buf))
        return buf.raw

    def destroy_comm(self):
        if self.comm is not None:
            _lib.check(self.lib.vf_comm_destroy(self.comm))
            self.comm = None
            for b in self._forks:
                b.comm = None

    def comm_barrier(self):
        _lib.check(self.lib.vf_comm_barrier(self.comm, self.ctx))

    def comm_broadcast(self, t, root=0):
        assert t.is_contiguous() and t.dtype in _COMM_DTYPE
        _lib.check(self.lib.vf_comm_broadcast(self.comm, self.ctx, _ptr(t), t.numel(), _COMM_DTYPE[t.dtype], root))

    def all_reduce(self, t, group=None, op="sum"):
        """in place, on this backend's stream (SyncBN's sums: the next kernel reads them)"""
        if self.comm is not None:
            assert t.is_contiguous() and t.dtype in _COMM_DTYPE
            _lib.check(self.lib.vf_comm_allreduce_inline(self.comm, self.ctx, _ptr(t), t.numel(), _COMM_DTYPE[t.dtype],
                                                         _COMM_OP[op]))
            return
        import torch.distributed as dist
        rop = {"sum": dist.ReduceOp.SUM, "max": dist.ReduceOp.MAX, "min": dist.ReduceOp.MIN}[op]
        dist.all_reduce(t, op=rop, group=group)

    def all_reduce_avg(self, t, world, group=None, async_op=False):
        """Mean over ranks of a flat gradient bucket.  RCCL averages inside the collective (no extra pass over the
        bucket); other backends sum, then scale.  async_op: returns a handle whose wait() orders this backend's stream
        after the collective, so kernels launched in between overlap it."""
        if self.comm is not None:
            assert t.is_contiguous() and t.dtype == torch.float32
            ticket = C.c_int32(-1)
            _lib.check(self.lib.vf_comm_allreduce_avg_async(self.comm, self.ctx, _ptr(t), t.numel(), C.byref(ticket)))
            h = _CommHandle(self, ticket.value)
            if async_op:
                return h
            h.wait()
            return None
        import torch.distributed as dist
        if dist.get_backend(group) == "nccl":
            h = dist.all_reduce(t, op=dist.ReduceOp.AVG, group=group, async_op=async_op)
            return h if async_op else None
        # (gloo, the CPU tests and the one-GPU rehearsal: it stages device tensors itself, ordered with the current stream, which
        #  is this backend's.  Round 3 wrapped these calls in a host copy after "intermittently a stale bucket" in the rehearsal;
        #  that was the packed-FMA fault of the then row-dot kernel showing under GPU sharing — DESIGN.md 4.9 — not gloo.)
        dist.all_reduce(t, group=group)
        self.scale_shift(t, 1.0 / world, 0.0)
        return None

    def reduce_scatter_avg(self, t, world, rank, group=None):
        """mean over ranks of shard `rank` of the flat vector t (numel % world == 0), in place at t[rank * n : (rank + 1) * n];
        returns that shard (a view).  The other shards of t are scratch afterwards."""
        n = t.numel() // world
        assert n * world == t.numel() and t.is_contiguous() and t.dtype == torch.float32
        if self.comm is not None:
            ticket = C.c_int32(-1)
            _lib.check(self.lib.vf_comm_reduce_scatter_avg_async(self.comm, self.ctx, _ptr(t), n, C.byref(ticket)))
            _CommHandle(self, ticket.value).wait()
        else:
            import torch.distributed as dist
            if dist.get_backend(group) == "nccl":      # RCCL through torch.distributed: on the device, in place (recv = send + rank * n)
                dist.reduce_scatter_tensor(t[rank * n:(rank + 1) * n], t, op=dist.ReduceOp.AVG, group=group)
            else:       # (gloo has no reduce-scatter: the whole vector is averaged, the shard read out of it)
                dist.all_reduce(t, group=group)
                self.scale_shift(t, 1.0 / world, 0.0)
        return t[rank * n:(rank + 1) * n]

    def all_gather_shards(self, t, world, rank, group=None, async_op=False):
        """every rank's shard t[r * n : (r + 1) * n] to all ranks, in place.  async_op: a handle whose wait() orders this backend's
        stream after the collective (the C-ABI exchange; other backends complete here and return None)"""
        n = t.numel() // world
        assert n * world == t.numel() and t.is_contiguous() and t.dtype == torch.float32
        if self.comm is not None:
            ticket = C.c_int32(-1)
            _lib.check(self.lib.vf_comm_allgather_async(self.comm, self.ctx, _ptr(t), n, C.byref(ticket)))
            h = _CommHandle(self, ticket.value)
            if async_op:
                return h
            h.wait()
            return None
        import torch.distributed as dist
        if dist.get_backend(group) == "nccl":          # RCCL through torch.distributed: on the device, in place (send = recv + rank * n)
            h = dist.all_gather_into_tensor(t, t[rank * n:(rank + 1) * n], group=group, async_op=async_op)
            return h if async_op else None
        parts = [t[r * n:(r + 1) * n] for r in range(world)]
        dist.all_gather(parts, parts[rank].clone(), group=group)
        return None

    def all_gather_ranges(self, t, ranges, rank, group=None, async_op=False):
        """rank r's block t[ranges[r][0] : ranges[r][1]] of the flat fp32 tensor t to every rank, in place — the row blocks of a weight
        tensor after an update sharded by rows (vf_net_fused_adam_row_range).  Equal, adjacent blocks travel as ONE all-gather; a ragged
        split (4000 rows over 3 ranks) as one broadcast per rank.  -> handles in flight (async_op; wait() orders this backend's stream
        behind them), [] when done."""
        world = len(ranges)
        lens = [hi - lo for lo, hi in ranges]
        if len(set(lens)) == 1 and all(ranges[r + 1][0] == ranges[r][1] for r in range(world - 1)):
            h = self.all_gather_shards(t[ranges[0][0]:ranges[-1][1]], world, rank, group, async_op=async_op)
            return [h] if h is not None else []
        hs = []
        for r, (lo, hi) in enumerate(ranges):
            if hi <= lo:
                continue
            blk = t[lo:hi]
            assert blk.is_contiguous() and blk.dtype == torch.float32
            if self.comm is not None:
                ticket = C.c_int32(-1)
                _lib.check(self.lib.vf_comm_broadcast_async(self.comm, self.ctx, _ptr(blk), hi - lo, r, C.byref(ticket)))
                hs.append(_CommHandle(self, ticket.value))
            else:
                import torch.distributed as dist
                src = r if group is None else dist.get_global_rank(group, r)
                h = dist.broadcast(blk, src=src, group=group, async_op=async_op)
                if async_op and h is not None:
                    hs.append(h)
        if not async_op:
            for h in hs:
                h.wait()
            return []
        return hs

    def _c(self, name, *args):
        _lib.check(getattr(self.lib, name)(self.ctx, *args))

    # ---- roctx ranges (vf_trace.hip): `with B.range("fDx"): ...` shows up in `rocprofv3 --marker-trace`; no-ops without roctx
    def range(self, name):
        return _Range(self.lib, name)

    def trace_enable(self, on=True):
        """one range per library launch as well (named like bench.py's kernel table)"""
        _lib.check(self.lib.vf_trace_enable(1 if on else 0))

    # ---- convolution family.  x/y logical BxCxHxW (NHWC physical); w logical as the reference (channels-last)
    def conv2d_fwd(self, x, w, bias, y, k, stride, pad, act="none", slope=0.0):
        B, Cin, H, W = x.shape
        self._c("vf_conv2d_fwd", _ptr(x), _ptr(w), _ptr(bias), _ptr(y), B, H, W, Cin, w.shape[0], k, stride, pad,
                ACT[act], slope)

    def conv2d_fwd_planes(self, x, w, bias, y, y_planes, k, stride, pad, act="none", slope=0.0):
        B, Cin, H, W = x.shape
        self._c("vf_conv2d_fwd_planes", _ptr(x), _ptr(w), _ptr(bias), _ptr(y), _ptr(y_planes), B, H, W, Cin, w.shape[0], k, stride,
                pad, ACT[act], slope)

    def conv2d_bwd_data(self, gy, w, gx, k, stride, pad):
        B, Cin, H, W = gx.shape
        self._c("vf_conv2d_bwd_data", _ptr(gy), _ptr(w), _ptr(gx), B, H, W, Cin, w.shape[0], k, stride, pad)

    def conv2d_bwd_data_act(self, gy, w, gx, x_act, act, slope, k, stride, pad):
        """conv data-gradient with the backward of the in-place activation that produced this conv's input."""
        B, Cin, H, W = gx.shape
        self._c("vf_conv2d_bwd_data_act", _ptr(gy), _ptr(w), _ptr(gx), _ptr(x_act), ACT[act], slope, B, H, W, Cin,
                w.shape[0], k, stride, pad)

    # ---- all weight gradients of one backward walk in one launch (vf_wgrad_group_begin / _end)
    def wgrad_group_begin(self):
        self._c("vf_wgrad_group_begin")

    def wgrad_group_end(self):
        self._c("vf_wgrad_group_end")

    def wgrad_group_abort(self):
        self._c("vf_wgrad_group_abort")

    def wgrad_group_count(self):
        n = C.c_int(0)
        self._c("vf_wgrad_group_count", C.byref(n))
        return n.value

    def wgrad_group_end_partial(self, count):
        self._c("vf_wgrad_group_end_partial", int(count))

    # ---- all conv bias gradients of one backward walk in two launches (vf_bias_grad_multi)
    def bias_grad_multi(self, items):
        """items: [(gradOutput B x C x H x W channels-last, gradBias [C], beta)], C % 4 == 0.  The descriptor table is built
        once per combination of buffers and betas and kept on the device (the same walk recurs every iteration)."""
        import numpy as np
        if not items:
            return
        key = tuple((g.data_ptr(), gb.data_ptr(), g.numel(), gb.numel(), float(beta)) for g, gb, beta in items)
        cache = self.__dict__.setdefault("_colsum_plans", {})
        plan = cache.get(key)
        if plan is None:
            desc = np.zeros(len(items), dtype=COLSUM_DESC)
            ws = self.workspace.data_ptr()
            off, b1, b2 = 0, 0, 0
            for i, (g, gb, beta) in enumerate(items):
                Cc = gb.numel()
                P = g.numel() // Cc
                assert Cc % 4 == 0 and g.data_ptr() % 16 == 0 and is_nhwc(g)
                geo = [C.c_int() for _ in range(4)]
                _lib.check(self.lib.vf_bias_grad_plan(P, Cc, *[C.byref(v) for v in geo]))
                cq, rpb, gx, gy = [v.value for v in geo]
                desc[i] = (g.data_ptr(), gb.data_ptr(), ws + off, P, Cc, cq, rpb, gx, gy, b1, b2, beta)
                off += (gx * Cc * 8 + 255) // 256 * 256
                b1 += gx * gy
                b2 += (Cc + 3) // 4
            assert off <= self.workspace.numel(), "workspace too small for the bias-gradient partials"
            dev = torch.from_numpy(desc.view(np.uint8).copy()).to(self.device)
            plan = (dev, len(items), b1, b2)
            cache[key] = plan
        dev, n, b1, b2 = plan
        self._c("vf_bias_grad_multi", _ptr(dev), n, b1, b2)

    def conv2d_bwd_weight(self, x, gy, gw, gb, k, stride, pad, beta, x_planes=None, gy_planes=None):
        B, Cin, H, W = x.shape
        if x_planes is not None and gy_planes is not None:
            self._c("vf_conv2d_bwd_weight_planes", _ptr(x), _ptr(gy), _ptr(x_planes), _ptr(gy_planes), _ptr(gw), _ptr(gb), B, H, W,
                    Cin, gw.shape[0], k, stride, pad, beta)
            return
        self._c("vf_conv2d_bwd_weight", _ptr(x), _ptr(gy), _ptr(gw), _ptr(gb), B, H, W, Cin, gw.shape[0], k, stride,
                pad, beta)

    def deconv2d_fwd(self, x, w, bias, y, k, stride, pad, act="none", slope=0.0):
        B, Cin, H, W = x.shape
        self._c("vf_deconv2d_fwd", _ptr(x), _ptr(w), _ptr(bias), _ptr(y), B, H, W, Cin, w.shape[1], k, stride, pad,
                ACT[act], slope)

    def deconv2d_bwd_data(self, gy, w, gx, k, stride, pad):
        B, Cin, H, W = gx.shape
        self._c("vf_deconv2d_bwd_data", _ptr(gy), _ptr(w), _ptr(gx), B, H, W, Cin, w.shape[1], k, stride, pad)

    def deconv2d_bwd_weight(self, x, gy, gw, gb, k, stride, pad, beta, x_planes=None, gy_planes=None):
        B, Cin, H, W = x.shape
        if x_planes is not None and gy_planes is not None:
            self._c("vf_deconv2d_bwd_weight_planes", _ptr(x), _ptr(gy), _ptr(x_planes), _ptr(gy_planes), _ptr(gw), _ptr(gb), B, H,
                    W, Cin, gw.shape[1], k, stride, pad, beta)
            return
        self._c("vf_deconv2d_bwd_weight", _ptr(x), _ptr(gy), _ptr(gw), _ptr(gb), B, H, W, Cin, gw.shape[1], k, stride,
                pad, beta)

    # ---- operands pre-split into three bf16 planes (vf_pgemm.hip).  `planes`: torch.bfloat16 [3, n]
    def planes_split(self, x, planes=None):
        n = x.numel()
        if planes is None:
            planes = torch.empty((3, n), dtype=torch.bfloat16, device=self.device)
        self._c("vf_planes_split", _ptr(x), _ptr(planes), n)
        return planes

    def weight_planes(self, w, native=None, transposed=None, want_transposed=True):
        """w: a conv / full-conv weight (logical [d0][d1][4][4], physical [d0][4][4][d1]) -> (native, transposed) planes"""
        d0, d1 = w.shape[0], w.shape[1]
        n = w.numel()
        if native is None:
            native = torch.empty((3, n), dtype=torch.bfloat16, device=self.device)
        if transposed is None and want_transposed:
            transposed = torch.empty((3, n), dtype=torch.bfloat16, device=self.device)
        self._c("vf_weight_planes", _ptr(w), _ptr(native), _ptr(transposed), d0, d1)
        return native, transposed

    def weight_planes_multi(self, items):
        """items: [(w, native planes, transposed planes)] -> one launch; returns the plan (kept by the caller: the table lives on
        the device and is reused every iteration)"""
        import numpy as np
        desc = np.zeros(len(items), dtype=WPLANES_DESC)
        blocks = 0
        for i, (w, nat, tr) in enumerate(items):
            d0, d1 = w.shape[0], w.shape[1]
            gx, gz = (d0 + 31) // 32, (d1 + 31) // 32
            desc[i] = (w.data_ptr(), nat.data_ptr(), tr.data_ptr(), d0, d1, gx, gz, blocks, 0)
            blocks += gx * 16 * gz
        dev = torch.from_numpy(desc.view(np.uint8).copy()).to(self.device)
        return (dev, len(items), blocks, tuple(w.data_ptr() for w, _, _ in items))

    def weight_planes_run(self, plan):
        dev, n, blocks, _ = plan
        self._c("vf_weight_planes_multi", _ptr(dev), n, blocks)

    def pconv_supported(self, B, H, W, Cin, Cout, k, stride, pad, transposed):
        """can the planes kernels serve this pass in the backend's CURRENT product mode (3: three exact planes; 1: one rounded
        plane, whole 64-wide tiles only; 0: never)?"""
        return bool(self.lib.vf_pconv_supported_in_mode(MFMA_MODES[self.mfma_mode], B, H, W, Cin, Cout, k, stride, pad, 1 if transposed else 0))

    def pconv_set_routing(self, gather_patch=-1, scatter_patch=-1):
        """process-wide: which kernels serve the planes passes (vf_pconv_set_routing; -1 leaves a setting alone)"""
        _lib.check(self.lib.vf_pconv_set_routing(int(gather_patch), int(scatter_patch)))

    def pconv_gather(self, ap, wp, bias, y, B, H, W, Cin, Cout, act="none", slope=0.0):
        self._c("vf_pconv_gather", _ptr(ap), _ptr(wp), _ptr(bias), _ptr(y), B, H, W, Cin, Cout, ACT[act], slope)

    def pconv_scatter(self, ap, wp, bias, y, B, H, W, Cin, Cout, act="none", slope=0.0, dmask=None, dact="none", dslope=0.0):
        self._c("vf_pconv_scatter", _ptr(ap), _ptr(wp), _ptr(bias), _ptr(y), B, H, W, Cin, Cout, ACT[act], slope, _ptr(dmask),
                ACT[dact], dslope)

    # ---- batch norm
    def bn_stats(self, x, shift, sums):
        B, Cc, H, W = x.shape
        self._c("vf_bn_stats", _ptr(x), _ptr(shift), _ptr(sums), B * H * W, Cc)

    def bn_finalize(self, sums, rm, rv, save_mean, save_invstd, n_total, momentum, eps):
        self._c("vf_bn_finalize", _ptr(sums), _ptr(rm), _ptr(rv), _ptr(save_mean), _ptr(save_invstd), n_total,
                rm.numel(), momentum, eps)

    def bn_apply(self, x, y, gamma, beta, mean, invstd, act="none", slope=0.0):
        B, Cc, H, W = x.shape
        self._c("vf_bn_apply", _ptr(x), _ptr(y), _ptr(gamma), _ptr(beta), _ptr(mean), _ptr(invstd), B * H * W, Cc,
                ACT[act], slope)

    def bn_train_fwd(self, x, y, gamma, beta, rm, rv, save_mean, save_invstd, sums, momentum, eps, act="none", slope=0.0):
        B, Cc, H, W = x.shape
        self._c("vf_bn_train_fwd", _ptr(x), _ptr(y), _ptr(gamma), _ptr(beta), _ptr(rm), _ptr(rv), _ptr(save_mean),
                _ptr(save_invstd), _ptr(sums), B * H * W, Cc, momentum, eps, ACT[act], slope)

    def bn_train_fwd_groups(self, x, y, gamma, beta, rm, rv, save_mean, save_invstd, sums, groups, momentum, eps,
                            act="none", slope=0.0, y_planes=None):
        """x = `groups` concatenated batches; save_mean / save_invstd [groups][C], sums [groups][2C] (vf_hip.h)."""
        B, Cc, H, W = x.shape
        assert B % groups == 0 and save_mean.numel() == groups * Cc and sums.numel() == groups * 2 * Cc
        if y_planes is not None:
            self._c("vf_bn_train_fwd_planes", _ptr(x), _ptr(y), _ptr(gamma), _ptr(beta), _ptr(rm), _ptr(rv), _ptr(save_mean),
                    _ptr(save_invstd), _ptr(sums), (B // groups) * H * W, Cc, groups, momentum, eps, ACT[act], slope, _ptr(y_planes))
            return
        self._c("vf_bn_train_fwd_groups", _ptr(x), _ptr(y), _ptr(gamma), _ptr(beta), _ptr(rm), _ptr(rv), _ptr(save_mean),
                _ptr(save_invstd), _ptr(sums), (B // groups) * H * W, Cc, groups, momentum, eps, ACT[act], slope)

    def bn_bwd_groups(self, x, y_act, gy, gx, ggamma, gbeta, gamma, save_mean, save_invstd, sums, groups, act="none",
                      slope=0.0, pbeta=1.0, gx_planes=None):
        B, Cc, H, W = x.shape
        assert B % groups == 0 and save_mean.numel() == groups * Cc and sums.numel() == groups * 2 * Cc
        if gx_planes is not None:
            self._c("vf_bn_bwd_planes", _ptr(x), _ptr(y_act), _ptr(gy), _ptr(gx), _ptr(ggamma), _ptr(gbeta), _ptr(gamma),
                    _ptr(save_mean), _ptr(save_invstd), _ptr(sums), (B // groups) * H * W, Cc, groups, ACT[act], slope, pbeta,
                    _ptr(gx_planes))
            return
        self._c("vf_bn_bwd_groups", _ptr(x), _ptr(y_act), _ptr(gy), _ptr(gx), _ptr(ggamma), _ptr(gbeta), _ptr(gamma),
                _ptr(save_mean), _ptr(save_invstd), _ptr(sums), (B // groups) * H * W, Cc, groups, ACT[act], slope, pbeta)

    # ---- BatchNorm statistics as a by-product of the producing convolution (vf_bn_fuse_next_* / vf_bn_*_pre)
    def bn_fuse_next_fwd(self, shift, part, groups=1):
        """the next conv / full-conv forward on this backend also sums (v - shift), (v - shift)^2 per channel into `part`
        (float64, rows x 2C); bn_fuse_result() afterwards: rows per group, 0 = that launch could not"""
        self._c("vf_bn_fuse_next_fwd", _ptr(shift), _ptr(part), part.numel() // (2 * shift.numel()), groups)

    def bn_fuse_next_bwd(self, x, y_act, act, slope, save_mean, part, groups=1):
        Cc = x.shape[1]
        self._c("vf_bn_fuse_next_bwd", _ptr(x), _ptr(y_act), ACT[act], slope, _ptr(save_mean), _ptr(part),
                part.numel() // (2 * Cc), groups)

    def bn_fuse_result(self):
        rows = C.c_int(0)
        self._c("vf_bn_fuse_result", C.byref(rows))
        return rows.value

    def bn_train_fwd_pre(self, part, rows, x, y, gamma, beta, rm, rv, save_mean, save_invstd, sums, groups, momentum, eps,
                         act="none", slope=0.0, y_planes=None):
        B, Cc, H, W = x.shape
        self._c("vf_bn_train_fwd_pre", _ptr(part), rows, _ptr(x), _ptr(y), _ptr(gamma), _ptr(beta), _ptr(rm), _ptr(rv),
                _ptr(save_mean), _ptr(save_invstd), _ptr(sums), (B // groups) * H * W, Cc, groups, momentum, eps, ACT[act], slope,
                _ptr(y_planes))

    def bn_bwd_pre(self, part, rows, x, g_masked, gx, ggamma, gbeta, gamma, save_mean, save_invstd, sums, groups, pbeta=1.0,
                   gx_planes=None):
        B, Cc, H, W = x.shape
        self._c("vf_bn_bwd_pre", _ptr(part), rows, _ptr(x), _ptr(g_masked), _ptr(gx), _ptr(ggamma), _ptr(gbeta), _ptr(gamma),
                _ptr(save_mean), _ptr(save_invstd), _ptr(sums), (B // groups) * H * W, Cc, groups, pbeta, _ptr(gx_planes))

    def bn_eval_fwd(self, x, y, gamma, beta, rm, rv, eps, act="none", slope=0.0):
        B, Cc, H, W = x.shape
        self._c("vf_bn_eval_fwd", _ptr(x), _ptr(y), _ptr(gamma), _ptr(beta), _ptr(rm), _ptr(rv), B * H * W, Cc, eps,
                ACT[act], slope)

    def bn_bwd_stats(self, x, y_act, gy, save_mean, sums, act="none", slope=0.0):
        B, Cc, H, W = x.shape
        self._c("vf_bn_bwd_stats", _ptr(x), _ptr(y_act), _ptr(gy), _ptr(save_mean), _ptr(sums), B * H * W, Cc,
                ACT[act], slope)

    def bn_bwd_apply(self, x, y_act, gy, gx, ggamma, gbeta, gamma, save_mean, save_invstd, sums, n_total, act="none",
                     slope=0.0, pbeta=1.0):
        B, Cc, H, W = x.shape
        self._c("vf_bn_bwd_apply", _ptr(x), _ptr(y_act), _ptr(gy), _ptr(gx), _ptr(ggamma), _ptr(gbeta), _ptr(gamma),
                _ptr(save_mean), _ptr(save_invstd), _ptr(sums), B * H * W, n_total, Cc, ACT[act], slope, pbeta)

    def bn_bwd(self, x, y_act, gy, gx, ggamma, gbeta, gamma, save_mean, save_invstd, sums, act="none", slope=0.0, pbeta=1.0):
        B, Cc, H, W = x.shape
        self._c("vf_bn_bwd", _ptr(x), _ptr(y_act), _ptr(gy), _ptr(gx), _ptr(ggamma), _ptr(gbeta), _ptr(gamma),
                _ptr(save_mean), _ptr(save_invstd), _ptr(sums), B * H * W, Cc, ACT[act], slope, pbeta)

    # ---- pointwise
    def act_fwd(self, x, y, act, slope=0.0):
        self._c("vf_act_fwd", _ptr(x), _ptr(y), x.numel(), ACT[act], slope)

    def act_bwd(self, y, gy, gx, act, slope=0.0):
        self._c("vf_act_bwd", _ptr(y), _ptr(gy), _ptr(gx), y.numel(), ACT[act], slope)

    def axpby(self, a, x, b, y):
        self._c("vf_axpby", a, _ptr(x), b, _ptr(y), y.numel())

    def cmul(self, x, y):
        self._c("vf_cmul", _ptr(x), _ptr(y), y.numel())

    def scale_shift(self, y, a, b):
        self._c("vf_scale_shift", _ptr(y), a, b, y.numel())

    # ---- batch preparation / inference tile loop (vf_pipeline.hip)
    def center_prepare(self, batch_nchw, ctx_out, center_out, fill, overlapPred):
        """train.lua:284-298.  batch_nchw: contiguous B x C x fs x fs; outputs are channels-last tensors."""
        B, Cc, fs, _ = batch_nchw.shape
        assert batch_nchw.is_contiguous()
        self._c("vf_center_prepare", _ptr(batch_nchw), _ptr(ctx_out), _ptr(center_out), _ptr(fill), B, Cc, fs, overlapPred)

    def clip_prepare(self, clip, mask, full, masked, maskout, w1, h1, flip, mask_value, blocks=None, block_size=0):
        """datavid/donkey_folder.lua:135-189 for one sample.  clip: C x iH x iW planar in [0,1]; mask: iH x iW float or
        None (then `blocks` = [(tlx, tly)] 1-based, randomBlockMask); outputs: 1 x C x fs x fs channels-last."""
        import ctypes
        Cc, iH, iW = clip.shape
        fs = full.shape[-1]
        blocks = blocks or []
        n = len(blocks)
        tlx = (ctypes.c_int * 10)(*([b[0] for b in blocks] + [0] * (10 - n)))
        tly = (ctypes.c_int * 10)(*([b[1] for b in blocks] + [0] * (10 - n)))
        self._c("vf_clip_prepare", _ptr(clip), _ptr(mask) if mask is not None else None, _ptr(full), _ptr(masked),
                _ptr(maskout), Cc, iH, iW, fs, w1, h1, int(bool(flip)), mask_value, n, block_size,
                ctypes.cast(tlx, ctypes.c_void_p), ctypes.cast(tly, ctypes.c_void_p))

    def tiles_gather(self, full, tiles, groups, vflip=None):
        """test_vid_wholeim.lua:159-175.  full: (groups*nc) x H x W planar; tiles: (T*groups) x nc x fs x fs channels-last."""
        Ct, H, W = full.shape
        fs = tiles.shape[-1]
        self._c("vf_tiles_gather", _ptr(full), _ptr(tiles), groups, Ct // groups, H, W, fs,
                _ptr(vflip) if vflip is not None else None)

    def tiles_scatter(self, tiles, out, groups, vflip=None):
        Ct, H, W = out.shape
        fs = tiles.shape[-1]
        self._c("vf_tiles_scatter", _ptr(tiles), _ptr(out), groups, Ct // groups, H, W, fs,
                _ptr(vflip) if vflip is not None else None)

    def channel_copy(self, src, c_src, dst, c_dst, ncopy):
        """dst[:, c_dst:c_dst+ncopy] = src[:, c_src:c_src+ncopy] on NHWC tensors of equal B, H, W (nn.JoinTable(2))."""
        Bn, Cs, H, W = src.shape
        assert tuple(dst.shape[0:1] + dst.shape[2:]) == (Bn, H, W)
        self._c("vf_channel_copy", _ptr(src), Cs, c_src, _ptr(dst), dst.shape[1], c_dst, ncopy, Bn * H * W)

    def noise_fill(self, out, seed, counter=0, normal=True, counter_dev=None):
        self._c("vf_noise_fill", _ptr(out), out.numel(), int(seed), _ptr(counter_dev) if counter_dev is not None else None,
                int(counter), 1 if normal else 0)

    def masked_compose(self, out, real, fake, mask):
        self._c("vf_masked_compose", _ptr(out), _ptr(real), _ptr(fake), _ptr(mask), out.numel())

    def zero(self, t):
        self._c("vf_zero", _ptr(t), t.numel() * t.element_size())

    def zero_segments(self, base, offs, lens):
        self._c("vf_zero_segments", _ptr(base), _ptr(offs), _ptr(lens), offs.numel())

    def copy(self, dst, src):
        dst.copy_(src)  # cudaMemcpyAsync D2D on the current stream — plumbing

    # ---- criteria (loss: 1-element float64 device tensor)
    def bce_fwd_bwd(self, x, label0, label1, n_per_group, groups, loss0, loss1, gx):
        self._c("vf_bce_fwd_bwd", _ptr(x), float(label0), float(label1), n_per_group, groups, _ptr(loss0), _ptr(loss1), _ptr(gx))

    def bce_fwd(self, x, label, loss):
        self._c("vf_bce_fwd", _ptr(x), float(label), x.numel(), _ptr(loss))

    def bce_bwd(self, x, label, gx):
        self._c("vf_bce_bwd", _ptr(x), float(label), _ptr(gx), x.numel())

    def mse_fwd(self, x, t, loss):
        self._c("vf_mse_fwd", _ptr(x), _ptr(t), x.numel(), _ptr(loss))

    def mse_bwd(self, x, t, gx):
        self._c("vf_mse_bwd", _ptr(x), _ptr(t), _ptr(gx), x.numel())

    def recon_grad_mix(self, df_dg, x, t, mask, alpha, c0, c1, band, loss):
        B, Cc, H, W = x.shape
        self._c("vf_recon_grad_mix", _ptr(df_dg), _ptr(x), _ptr(t), _ptr(mask), alpha, c0, c1, band, H, Cc, x.numel(),
                _ptr(loss))

    def gdl_fwd(self, yhat, y, loss):
        B, Cc, H, W = y.shape
        self._c("vf_gdl_fwd", _ptr(yhat), _ptr(y), B, H, W, Cc, _ptr(loss))

    def gdl_bwd(self, yhat, y, gyhat):
        B, Cc, H, W = y.shape
        self._c("vf_gdl_bwd", _ptr(yhat), _ptr(y), _ptr(gyhat), B, H, W, Cc)

    def masked_mse_fwd(self, x, xhat, mask_u8, w, loss):
        self._c("vf_masked_mse_fwd", _ptr(x), _ptr(xhat), _ptr(mask_u8), w, x.numel(), _ptr(loss))

    def masked_mse_bwd(self, x, xhat, mask_u8, w, gx):
        self._c("vf_masked_mse_bwd", _ptr(x), _ptr(xhat), _ptr(mask_u8), w, _ptr(gx), x.numel())

    # ---- adam
    def adam_step(self, x, g, m, v, lr, beta1, beta2, eps, t_dev):
        self._c("vf_adam_step", _ptr(x), _ptr(g), _ptr(m), _ptr(v), x.numel(), lr, beta1, beta2, eps, _ptr(t_dev))

    def adam_prep(self, lr, beta1, beta2, t_dev):
        self._c("vf_adam_prep", lr, beta1, beta2, _ptr(t_dev))

    def adam_apply(self, x, g, m, v, beta1, beta2, eps, t_dev):
        self._c("vf_adam_apply", _ptr(x), _ptr(g), _ptr(m), _ptr(v), x.numel(), beta1, beta2, eps, _ptr(t_dev))

    def adam_apply_ranges(self, x, g, m, v, ranges, beta1, beta2, eps, t_dev):
        """vf_adam_apply_ranges: the update of the element ranges [(lo, hi)] of the flat vectors in one launch"""
        ranges = [(lo, hi) for lo, hi in ranges if hi > lo]
        if not ranges:
            return
        n = len(ranges)
        offs = (C.c_int64 * n)(*[lo for lo, _ in ranges])
        lens = (C.c_int64 * n)(*[hi - lo for lo, hi in ranges])
        self._c("vf_adam_apply_ranges", _ptr(x), _ptr(g), _ptr(m), _ptr(v), offs, lens, n, beta1, beta2, eps, _ptr(t_dev))

    def wgrad_adam_outer(self, U, V, x, m, v, g, beta1, beta2, eps, t_dev):
        """vf_wgrad_adam_outer: the bottleneck weight gradient U^T V (U [K][Nu], V [K][Ncols]) consumed by optim.adam in the
        kernel that forms it; x, m, v (and g, or None) = [Nu][Ncols] slices, t_dev after adam_prep."""
        K, Nu = U.shape
        Ncols = V.shape[1]
        assert V.shape[0] == K and x.numel() == Nu * Ncols
        self._c("vf_wgrad_adam_outer", _ptr(U), _ptr(V), K, Nu, Ncols, _ptr(x), _ptr(m), _ptr(v), _ptr(g) if g is not None else None,
                beta1, beta2, eps, _ptr(t_dev))

    def wgrad_adam_outer_gathered(self, buf, u_off, v_off, world, K, seg, Nu, Ncols, x, m, v, g, beta1, beta2, eps, t_dev):
        """vf_wgrad_adam_outer_gathered: `buf` holds `world` segments of `seg` floats; segment r has rank r's U [K][Nu] at u_off and
        V [K][Ncols] at v_off.  g = the MEAN over ranks of U_r^T V_r (the data-parallel gradient), consumed by optim.adam in place."""
        base = buf.data_ptr()
        self._c("vf_wgrad_adam_outer_gathered", C.c_void_p(base + 4 * u_off), C.c_void_p(base + 4 * v_off), world * K, K, seg, Nu, Ncols,
                _ptr(x), _ptr(m), _ptr(v), _ptr(g) if g is not None else None, 1.0 / world, beta1, beta2, eps, _ptr(t_dev))

    def wgrad_adam_outer_rows(self, buf, u_off, v_off, world, K, seg, Nu, Ncols, row0, nrows, x, m, v, g, beta1, beta2, eps, t_dev):
        """vf_wgrad_adam_outer_rows: wgrad_adam_outer_gathered for rows [row0, row0 + nrows) of the tensors only (x, m, v, g are the
        whole [Nu][Ncols] tensors) — one rank's share of the update sharded by weight rows"""
        base = buf.data_ptr()
        self._c("vf_wgrad_adam_outer_rows", C.c_void_p(base + 4 * u_off), C.c_void_p(base + 4 * v_off), world * K, K, seg, Nu, Ncols,
                row0, nrows, _ptr(x), _ptr(m), _ptr(v), _ptr(g) if g is not None else None, 1.0 / world, beta1, beta2, eps, _ptr(t_dev))

    # ---- per-kernel timers
    def prof_begin(self):
        self._c("vf_prof_begin")

    def prof_end(self):
        """-> {kernel name: dict(launches, ms, flops, bytes)}"""
        self._c("vf_prof_end")
        out = {}
        for i in range(self.lib.vf_prof_count()):
            name = C.create_string_buffer(96)
            n, ms, fl, by = C.c_int64(), C.c_double(), C.c_double(), C.c_double()
            _lib.check(self.lib.vf_prof_get(i, name, 96, C.byref(n), C.byref(ms), C.byref(fl), C.byref(by)))
            out[name.value.decode()] = dict(launches=n.value, ms=ms.value, flops=fl.value, bytes=by.value)
        return out

    def synchronize(self):
        _lib.check(self.lib.vf_stream_synchronize(self.ctx))


class _SideScope:
    def __init__(self, side):
        self.side = side

    def __enter__(self):
        global _BACKEND
        self.prev = _BACKEND
        self.side.stream.wait_stream(torch.cuda.current_stream(self.side.device))
        self.cm = torch.cuda.stream(self.side.stream)
        self.cm.__enter__()
        _BACKEND = self.side
        return self.side

    def __exit__(self, *exc):
        global _BACKEND
        _BACKEND = self.prev
        self.cm.__exit__(*exc)
        return False


_BACKEND = None


def get_backend():
    """The process-wide backend; created on first use.  Raises without a GPU — no CPU fallback exists."""
    global _BACKEND
    if _BACKEND is None:
        _BACKEND = HipBackend()
    return _BACKEND


def set_backend(b):
    """Install a backend object (tests only: host-logic checks on CPU)."""
    global _BACKEND
    _BACKEND = b
    return b
