"""The reference drivers' hot loop — net construction, the fDx / fGx closures and the two optim.adam calls —
restated over the nn mirror so it reads like the Lua it replaces.

  CenterTrainer  <- train.lua                  (net: :87-199, closures: :278-410, loop body: :421-424)
  VidTrainer     <- train_vid_weighted.lua     (net: :112-236, closures: :373-537, loop body: :548-551)
                    train_wholeim_input.lua    (same closures; nc_in/nc_out/widths differ, :116-119,137-260)

Option names are the reference's (`opt` table / environment variables).  Data loading, display and checkpoint
cadence are outside the hot path; `set_batch` takes tensors with the loaders' contract.

Data parallelism (new capability; the reference is single-device): one process per GPU, each with a
full replica and an equal shard of the batch; flat gradients are all-reduce-averaged over RCCL at the end of
each closure, which reproduces the single-device big-batch gradient because every criterion is a batch mean.
`sync_bn=True` additionally all-reduces the BatchNorm sums so statistics equal the big batch's.
"""
import os

import torch

from . import data, nn, optim
from .backend import bump_param_version, get_backend, to_nhwc

DEFAULT_OPT_TRAIN = dict(batchSize=64, fineSize=128, nBottleneck=100, nef=64, ngf=64, ndf=64, nc=3, wtl2=0.0,
                         overlapPred=0, lr=0.0002, beta1=0.5, nz=100, conditionAdv=False, noiseGen=False,
                         noisetype="normal")
DEFAULT_OPT_VID = dict(batchSize=16, fineSize=128, nBottleneck=4000, nef=64, ngf=64, ndf=64, nc=3, predLen=4,
                       wtl2=0.999, weight_nomask=0.05, wtgdl=0.0, overlapPred=0, lr=0.0002, beta1=0.5,
                       nc_in=None, nc_out=None)


# Which host drives netG / netD: "mirror" = the module-by-module nn.Sequential of nn.py (every layer call crosses the C-ABI on its
# own), "cabi" = the library's own net object (vf_net_*, cnet.CNet: one C-ABI call per Torch7 method, the whole fast path inside
# libvf_hip.so — what a Lua host gets through hipnn.Net).  Same kernels, same plan; nets with table modules (train.lua's option
# branches) stay on the mirror under either setting.
DEFAULT_HOST = os.environ.get("VF_HOST", "cabi")


def _host_nets(host, netG, netD, sync_world=1):
    host = host or DEFAULT_HOST
    assert host in ("mirror", "cabi"), host
    if getattr(get_backend(), "name", "") != "hip-gfx950":
        host = "mirror"        # (a test backend on the CPU: host logic only — vf_net lives in libvf_hip.so)
    if host == "cabi" and sync_world > 1 and getattr(get_backend(), "comm", None) is None:
        # SyncBN inside vf_net exchanges its sums through vf_comm_*; without that communicator (bench.py --comm torch, or the
        # --comm auto fallback) the module-by-module mirror carries the exchange over torch.distributed instead
        host = "mirror"
    if host == "cabi":
        from .cnet import adopt_if_chain
        return adopt_if_chain(netG), adopt_if_chain(netD), host
    return netG, netD, host


def opt_from_env(defaults):
    """`for k,v in pairs(opt) do opt[k] = tonumber(os.getenv(k)) or os.getenv(k) or opt[k] end` (train.lua:36)."""
    opt = dict(defaults)
    for k in opt:
        v = os.getenv(k)
        if v is not None:
            try:
                opt[k] = float(v) if ("." in v or "e" in v.lower()) else int(v)
            except ValueError:
                opt[k] = v
    return opt


def _conv(nIn, nOut, s2=True):
    return nn.SpatialConvolution(nIn, nOut, 4, 4, 2, 2, 1, 1) if s2 else nn.SpatialConvolution(nIn, nOut, 4, 4)


def _full(nIn, nOut, s2=True):
    return nn.SpatialFullConvolution(nIn, nOut, 4, 4, 2, 2, 1, 1) if s2 else nn.SpatialFullConvolution(nIn, nOut, 4, 4)


def _acts(smooth):
    """smooth=True is a parity-test aid: every LeakyReLU(0.2)/ReLU becomes LeakyReLU(1.0) — same graph, same
    kernels, no derivative discontinuity (tests/test_gpu_trainers.py explains why)."""
    if not smooth:
        return nn.LeakyReLU, nn.ReLU
    return (lambda negval, inplace: nn.LeakyReLU(1.0, inplace)), (lambda inplace: nn.LeakyReLU(1.0, inplace))


def build_netG(nc_in, nc_out, nef, ngf, nBottleneck, extra_decoder_layer, fuse=True, lazy_zero=True, smooth=False, noise_nz=0,
               half_last=False, extra_bottleneck_stage=False):
    """train.lua:87-148 (64x64 output) / train_vid_weighted.lua:112-176 (extra ngf->ngf layer, 128x128 output).
    noise_nz > 0: the noiseGen generator (train.lua:109-124), input {context, noise [B, nz, 1, 1]};
    half_last: train_logo_withmask.lua:95-98, the extra decoder layer is ngf -> ngf/2.
    extra_bottleneck_stage: NOT in the reference — the labelled 256x256 extension (opt.ext256).  As written the generator
    "works" at fineSize 256 with a 5x5 bottleneck map, i.e. a 4x4 valid conv 8x8 -> 5x5 and its full-conv twin:
    non-power-of-two maps that only the thin-layer generic kernels serve.  The extension keeps the bottleneck at 1x1
    instead: one more stride-2 stage nef*8 -> nef*8 (8x8 -> 4x4) in front of it and ngf*8 -> ngf*8 (4x4 -> 8x8) behind it,
    so every layer stays on the matrix-core path — the same kind of addition netD needs at that size."""
    BN = nn.SpatialBatchNormalization
    LReLU, ReLU = _acts(smooth)
    netE = nn.Sequential(fuse, lazy_zero)
    netE.add(_conv(nc_in, nef)).add(LReLU(0.2, True))
    netE.add(_conv(nef, nef)).add(BN(nef)).add(LReLU(0.2, True))
    netE.add(_conv(nef, nef * 2)).add(BN(nef * 2)).add(LReLU(0.2, True))
    netE.add(_conv(nef * 2, nef * 4)).add(BN(nef * 4)).add(LReLU(0.2, True))
    netE.add(_conv(nef * 4, nef * 8)).add(BN(nef * 8)).add(LReLU(0.2, True))
    if extra_bottleneck_stage:
        netE.add(_conv(nef * 8, nef * 8)).add(BN(nef * 8)).add(LReLU(0.2, True))
    netE.add(_conv(nef * 8, nBottleneck, s2=False))
    netG = nn.Sequential(fuse, lazy_zero)
    nz_size = nBottleneck
    if noise_nz:
        netG_noise = nn.Sequential(fuse, lazy_zero).add(nn.SpatialConvolution(noise_nz, noise_nz, 1, 1, 1, 1, 0, 0))
        netG.add(nn.ParallelTable().add(netE).add(netG_noise))
        netG.add(nn.JoinTable(2))
        nz_size = nBottleneck + noise_nz
    else:
        netG.add(netE)
    netG.add(BN(nz_size)).add(LReLU(0.2, True))
    netG.add(_full(nz_size, ngf * 8, s2=False)).add(BN(ngf * 8)).add(ReLU(True))
    if extra_bottleneck_stage:
        netG.add(_full(ngf * 8, ngf * 8)).add(BN(ngf * 8)).add(ReLU(True))
    netG.add(_full(ngf * 8, ngf * 4)).add(BN(ngf * 4)).add(ReLU(True))
    netG.add(_full(ngf * 4, ngf * 2)).add(BN(ngf * 2)).add(ReLU(True))
    netG.add(_full(ngf * 2, ngf)).add(BN(ngf)).add(ReLU(True))
    last = ngf
    if extra_decoder_layer:
        last = ngf // 2 if half_last else ngf
        netG.add(_full(ngf, last)).add(BN(last)).add(ReLU(True))
    netG.add(_full(last, nc_out)).add(nn.Tanh())
    return netG


def build_netD(nc, ndf, extra_first_layer, fuse=True, lazy_zero=True, smooth=False, conditionAdv=False, extra_last_layer=False):
    """train.lua:157-199 (64x64 input) / train_vid_weighted.lua:213-236 (extra floor(ndf/2) layer, 128x128 input).
    conditionAdv (train.lua:158-180): input {context 128x128, prediction 64x64}, two 5x5 stride-2 branches joined.
    extra_last_layer: NOT in the reference — the labelled 256x256 extension (opt.ext256): one more stride-2 block
    ndf*8 -> ndf*8 in front of the final 4x4 conv, without which the reference's own netD yields 5x5 scores per sample at
    fineSize 256 and its BCECriterion fails on the label's size."""
    BN = nn.SpatialBatchNormalization
    LReLU, _ = _acts(smooth)
    netD = nn.Sequential(fuse, lazy_zero)
    if conditionAdv:
        assert not extra_first_layer
        netD_ctx = nn.Sequential(fuse, lazy_zero).add(nn.SpatialConvolution(nc, ndf, 5, 5, 2, 2, 2, 2))
        # 32: to keep the scaling of the features the same as the context's (train.lua:166)
        netD_pred = nn.Sequential(fuse, lazy_zero).add(nn.SpatialConvolution(nc, ndf, 5, 5, 2, 2, 2 + 32, 2 + 32))
        netD.add(nn.ParallelTable().add(netD_ctx).add(netD_pred))
        netD.add(nn.JoinTable(2))
        netD.add(LReLU(0.2, True))
        netD.add(_conv(ndf * 2, ndf)).add(BN(ndf)).add(LReLU(0.2, True))
    elif extra_first_layer:
        mylayer = ndf // 2
        netD.add(_conv(nc, mylayer)).add(LReLU(0.2, True))
        netD.add(_conv(mylayer, ndf)).add(LReLU(0.2, True))
    else:
        netD.add(_conv(nc, ndf)).add(LReLU(0.2, True))
    netD.add(_conv(ndf, ndf * 2)).add(BN(ndf * 2)).add(LReLU(0.2, True))
    netD.add(_conv(ndf * 2, ndf * 4)).add(BN(ndf * 4)).add(LReLU(0.2, True))
    netD.add(_conv(ndf * 4, ndf * 8)).add(BN(ndf * 8)).add(LReLU(0.2, True))
    if extra_last_layer:
        netD.add(_conv(ndf * 8, ndf * 8)).add(BN(ndf * 8)).add(LReLU(0.2, True))
    netD.add(_conv(ndf * 8, 1, s2=False)).add(nn.Sigmoid())
    netD.add(nn.View(1).setNumInputDims(3))
    return netD


def _ext256(o):
    """fineSize 256 (BASELINE configs[4] as quoted): the reference's netD does not work there and its netG only
    with a 5x5 bottleneck map — so that size runs only as the labelled extension `ext256=True`: one more stride-2 stage in
    netD (build_netD extra_last_layer) and around netG's bottleneck (build_netG extra_bottleneck_stage)."""
    fs = o.get("fineSize", 128)
    if fs != 256:
        assert not o.get("ext256"), "ext256 is the fineSize-256 extension"
        return False
    assert o.get("ext256"), ("fineSize 256: the reference's netD yields 5x5 scores per sample there and its criterion fails "
                             "(SURVEY D5); this size runs only as the labelled non-parity extension opt.ext256 (one more netD block)")
    return True


def weights_init(net, gen):
    """train.lua:58-67.  `gen` is a torch.Generator on CPU (Torch7's MT19937 stream cannot be reproduced;
    parity tests load explicit weights instead)."""
    B = get_backend()

    def init(m):
        name = m.type_name()
        if "Convolution" in name:
            w = torch.empty(m.weight.shape).normal_(0.0, 0.02, generator=gen)
            m.weight.copy_(B.from_host(w))
            m.bias.zero_()
        elif "BatchNormalization" in name:
            m.weight.copy_(B.from_host(torch.empty(m.weight.shape).normal_(1.0, 0.02, generator=gen)))
            m.bias.zero_()

    net.apply(init)


def _solver(opt):
    wt = opt["wtl2"]
    lrG = opt["lr"] * 10 if (wt > 0 and wt < 1) else opt["lr"]      # train.lua:219-226
    return ({"learningRate": lrG, "beta1": opt["beta1"]}, {"learningRate": opt["lr"], "beta1": opt["beta1"]})


import contextlib


def _no_range(name):
    return contextlib.nullcontext()


def _eager_sync_every():
    """Bound on un-synchronised eager iterations (~150-270 launches each).  Under a counter-collecting profiler every dispatch is
    serialised behind the tool's own packets and ~12 k queued dispatches ended in a fault inside the runtime/profiler dispatch
    path; so with a rocprofiler tool library in the process (or VF_EAGER_SYNC_EVERY set) the eager loop drains
    the device every N iterations.  0 = never (the default without a profiler: nothing is gained by waiting)."""
    v = os.environ.get("VF_EAGER_SYNC_EVERY")
    if v is not None:
        return max(int(v), 0)
    tools = " ".join(os.environ.get(k, "") for k in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "HSA_TOOLS_LIB"))
    return 64 if "rocprof" in tools else 0


class _TrainerBase:
    def _finish_init(self, seed, world, rank, group, sync_bn, skip_dead_grads, overlap=True, shard_adam=False):
        # stream-level overlap: weight gradients beside the data-gradient chain (both nets) and netG's forward
        # beside netD's real pass.  Each side stream has its own context and workspace (backend.fork()).
        self.side_g = None
        B0 = get_backend()
        from .cnet import CNet
        if any(isinstance(net, CNet) for net in (self.netG, self.netD)):
            overlap = False        # a vf_net lives on ONE context / stream; the side-stream experiments belong to the mirror
        if overlap and hasattr(B0, "fork"):
            self.netD.side = B0.fork()
            self.netG.side = self.netD.side
            self.side_g = B0.fork()
        gen = torch.Generator().manual_seed(seed)
        weights_init(self.netG, gen)
        weights_init(self.netD, gen)
        self.criterion = nn.BCECriterion()
        self.optimStateG, self.optimStateD = _solver(self.opt)
        self.parametersD, self.gradParametersD = self.netD.getParameters()
        self.parametersG, self.gradParametersG = self.netG.getParameters()
        self.world, self.rank, self.group = world, rank, group
        self.skip_dead_grads = skip_dead_grads
        # shard_adam (data parallel, un-pipelined step): the generator's gradient is REDUCE-SCATTERED instead of all-reduced,
        # every rank applies Adam to its 1 / world of the parameters (its m and v are that size too) and the updated shards
        # are all-gathered: the bytes on the wire of one all-reduce, 1 / world of the 28-bytes-per-parameter update per rank
        self.shard_adam = bool(shard_adam) and world > 1
        if self.shard_adam:
            assert self.parametersG.numel() % (4 * world) == 0, "shard_adam: the flat vector must split into 16-byte-aligned shards"
            n = self.parametersG.numel() // world
            self._shard = (rank * n, (rank + 1) * n)
            self.optimStateG_shard = dict(self.optimStateG)
        if world > 1 and sync_bn:
            for net in (self.netG, self.netD):
                for m in net.leaves():
                    if isinstance(m, nn.SpatialBatchNormalization):
                        m.sync_world, m.sync_group = world, group
        # weight planes of the planes-fed convolutions (nn.Sequential.refresh_weight_planes): refreshed by the closures,
        # once per net and parameter update, instead of on every forward / backward call
        self.netG.set_weight_planes_managed(True)
        self.netD.set_weight_planes_managed(True)
        self.errD = self.errG = self.errG_l2 = self.errG_gdl = None
        self._graph = None
        self._graphs = None
        self._defer_comm = False
        self._split_g = None
        self._g_mid = None
        self._pipelined = False      # data-parallel step with G's exchange + Adam deferred into the next iteration
        self._inflight = []          # async all-reduce handles of G's gradient buckets
        self._gen = None
        self._force_comm = False     # run the exchange even at world == 1 (exercises the DP path on one GPU)
        # optim.adam(fGx) with the bottleneck pair's weight gradients consumed inside the kernel that forms them
        # (optim.adam_update_fused; single device, the C-ABI host): "on" (default; gradParametersG does not receive those two
        # slices), "keep" (it does: 28 B per weight instead of 24), "off" (accGradParameters + the plain one-pass update, 32 B)
        self.fuse_adam = os.environ.get("VF_FUSE_ADAM", "on")
        assert self.fuse_adam in ("on", "keep", "off"), "VF_FUSE_ADAM: on | keep | off"
        # data parallel, what becomes of the bottleneck pair's gradient (92 % of G's bytes) — a recorded choice:
        #   "gathered" (default)  every rank all-gathers the OPERANDS (6 MB per rank) and forms the global-batch gradient of ALL rows
        #                         inside the fused update: 72 MB on the wire at 8 ranks, N-fold redundant matrix-core work
        #   "rows"                the same gather, but rank r forms and applies rows [r R / N, (r + 1) R / N) only (2 B flops per weight
        #                         again, Adam traffic / N), then the updated rows are all-gathered (131 MB per tensor)
        #   "reduced"             the pair's gradients are all-reduced like everything else (fuse_adam off under data parallelism)
        #   Default since round 5: "rows" — it does strictly less work than "gathered", and its row exchange is issued at the top of the NEXT
        #   iteration, on the communicator's stream, with the wait in front of the generator's bottleneck conv (E1 ... E5 read none of
        #   those rows); where a tensor's rows cannot be dealt in blocks of >= 64 the trainer keeps the gathered form for that step.
        self.dp_fused = os.environ.get("VF_DP_FUSED", "rows")
        assert self.dp_fused in ("gathered", "rows", "reduced"), "VF_DP_FUSED: gathered | rows | reduced"
        self._dpf = []               # phased data-parallel step: the fused slices of this iteration (set in phase B)
        self._opbuf = None           # ... and the gather buffer of their operands: world segments
        self._row_ranges = None      # rows mode: per fused slice, the (lo, hi) block of every rank (cnet.fused_adam_row_ranges)
        self._rows_stale = False     # rows mode: phase C updated this rank's rows only and the exchange of the row blocks is still owed
        self.defer_adam_g = False
        self.adam_overlap = False    # enable_adam_overlap(): Adam(G)'s two big weight tensors beside the next encoder forward
        self.side_a = None
        self._g_big = None
        self._pending_g = False
        self._graph_stale = False
        self.batch_d = False         # set_batch_d(): netD's real and fake passes as one batch of 2B
        self._cat = self._cat_df = None
        self._cat_real = None        # (buffer, batch version) the real half of _cat was filled from
        self._batch_ver = 0
        if not (world > 1 and sync_bn) and os.environ.get("VF_NO_BATCH_D") != "1":
            # measured +5 % on train.lua's nets; data parallel too, except with SyncBN (whose statistics are
            # all-reduced per layer and pass, one group at a time) and in the pipelined step (step_pipelined switches it off:
            # there netD's separate real pass is the window behind which G's exchange completes)
            self.set_batch_d(True)

    @property
    def force_comm(self):
        return self._force_comm

    @force_comm.setter
    def force_comm(self, v):
        self._force_comm = bool(v)

    def _comm_on(self):
        return self.world > 1 or self.force_comm

    def _g_in(self):
        """what netG:forward / netG:backward receive (a table when the generator takes noise, train.lua:326,404)"""
        return self.input_ctx

    # -- netD's two passes of fDx as ONE batch [real; fake] (single device).  The reference runs netD twice per
    #    closure on B samples each (train.lua:331-349); the convolutions are linear in the batch and gradParameters
    #    accumulates over the two backward calls, so one pass over 2B samples is the same arithmetic as long as every
    #    BatchNorm keeps the two halves apart (nn.SpatialBatchNormalization.groups = 2: statistics, running averages
    #    and backward sums per half, real first).  Twice the rows per GEMM launch and half the launches for netD.
    def set_batch_d(self, on=True):
        assert not (on and self._pipelined), "the pipelined data-parallel step keeps netD's real pass separate"
        assert not (on and any(getattr(m, "sync_world", 1) > 1 for m in self.netD.leaves())), "batch_d and SyncBN do not combine"
        assert self._graph is None and self._graphs is None, "set_batch_d before capture()"
        self.batch_d = bool(on)
        self.netD.setBatchGroups(2 if on else 1)
        return self

    def _netD_both(self, real_in, fake_in):
        """NOTE: the real half of the [real; fake] tensor is refreshed by set_batch() (keyed by buffer and batch version); a batch
        edited IN PLACE without set_batch() is not seen here (nor by a captured graph) — batches change through set_batch() only."""
        B = get_backend()
        n = real_in.shape[0]
        shp = (2 * n,) + tuple(real_in.shape[1:])
        if self._cat is None or tuple(self._cat.shape) != shp:
            self._cat = B.empty_act(*shp)
            self._cat_real = None
        # real half: the batch only changes in set_batch(), which refreshes this half itself (_refresh_cat_real): no copy per
        # iteration.  fake half: the generator's last convolution is pointed at this half, so from the next forward on its output
        # IS the fake half (no copy either); whatever else arrives (the masked composite, a side-stream buffer) is copied.
        key = (real_in.data_ptr(), self._batch_ver)
        if self._cat_real != key:
            B.copy(self._cat[:n], real_in)
            self._cat_real = key
        if fake_in.data_ptr() != self._cat[n:].data_ptr():
            B.copy(self._cat[n:], fake_in)
            self.netG.redirect_last_output(fake_in, self._cat[n:])
        out = self.netD.forward(self._cat)
        if self._cat_df is None or tuple(self._cat_df.shape) != tuple(out.shape):
            self._cat_df = B.zeros(out.numel()).view(out.shape)
        if hasattr(B, "bce_fwd_bwd") and out.is_contiguous():
            # criterion:forward and :backward of both halves (train.lua:331-349) in one launch
            s0, s1 = self.criterion.next_slot(), self.criterion.next_slot()
            B.bce_fwd_bwd(out, self.real_label, self.fake_label, n, 2, s0, s1, self._cat_df)
            errD_real, errD_fake = nn.DeviceScalar.of(s0), nn.DeviceScalar.of(s1)
        else:
            errD_real = self.criterion.forward(out[:n], self.real_label)
            errD_fake = self.criterion.forward(out[n:], self.fake_label)
            B.bce_bwd(out[:n], self.real_label, self._cat_df[:n])
            B.bce_bwd(out[n:], self.fake_label, self._cat_df[n:])
        self.netD.backward(self._cat, self._cat_df, need_input_grad=not self.skip_dead_grads)
        return errD_real + errD_fake

    def _refresh_cat_real(self, real_in):
        """set_batch(): a new batch is in the persistent buffers; the real half of the [real; fake] tensor follows here, outside
        the iteration (and outside a captured graph, which never copies it)"""
        self._batch_ver += 1
        if self.batch_d and self._cat is not None and self._cat.shape[0] == 2 * real_in.shape[0] and tuple(self._cat.shape[1:]) == tuple(real_in.shape[1:]):
            get_backend().copy(self._cat[:real_in.shape[0]], real_in)
            self._cat_real = (real_in.data_ptr(), self._batch_ver)

    def _netD_stale_output(self):
        out = self.netD.output
        return out[out.shape[0] // 2:] if self.batch_d else out

    def _netD_grad_input(self, x, df_do):
        """netD:updateGradInput(x, df_do) of fGx (train.lua:366): x is what the fake pass saw."""
        if self.batch_d:
            return self.netD.updateGradInput(x, df_do, group=(1, 2))
        return self.netD.updateGradInput(x, df_do)

    def _allreduce_avg(self, flat):
        """RCCL all-reduce-average of a flat gradient vector.  Inside a phased step the closure
        leaves the exchange to the step (`_defer_comm`)."""
        if self._comm_on() and not self._defer_comm:
            get_backend().all_reduce_avg(flat, self.world, self.group)

    def fDx(self, x):
        for _ in self._fDx_gen():
            pass
        return self.errD, self.gradParametersD

    def _backward_G(self, df_dg):
        """netG:backward(input_ctx, df_dg) + the exchange of its gradients.  In a phased step only the part of the
        pass that completes the big bucket (bottleneck + decoder, > 90 % of the bytes) runs here; `_phase_b2`
        runs the rest while that bucket is on the wire."""
        need = not self.skip_dead_grads
        if self._split_g is not None:
            # ONE uninterrupted walk; the gradients of the tail bucket (entries >= k) leave at its end, the encoder's when
            # _phase_b2 finishes the walk's group — both at the END of the data-gradient chain (a gradient launch in the middle of
            # it slowed the passes behind it by 20-40 %: DESIGN.md 8)
            k, _ = self._split_g
            self.netG.backward_split(self._g_in(), df_dg, k, need)
            return
        self.netG.backward(self._g_in(), df_dg, need_input_grad=need)
        self._allreduce_avg(self.gradParametersG)

    def step(self):
        """The loop body: optim.adam(fDx, ...) ; optim.adam(fGx, ...)  (train.lua:421-424).

        With `defer_adam_g` the update half of the second call is issued at the start of the NEXT iteration, on the
        side stream that also runs netG's forward, beside netD's real pass (which reads no generator state): the
        28 B/param HBM-bound update then overlaps MFMA-bound work.  Same arithmetic, same order on every buffer;
        `flush()` applies a pending update (call it before reading parametersG)."""
        B = get_backend()
        assert not self.shard_adam, "shard_adam keeps its own 1 / world Adam state: use step_phased() (ADVICE r2)"
        every = self.__dict__.get("_sync_every")
        if every is None:
            every = self._sync_every = _eager_sync_every()
        if every and torch.cuda.is_available() and not torch.cuda.is_current_stream_capturing():
            self._eager_n = self.__dict__.get("_eager_n", 0) + 1
            if self._eager_n % every == 0:
                torch.cuda.synchronize()
        rng = B.range if hasattr(B, "range") else _no_range         # roctx ranges of the two optimiser calls (rocprofv3 --marker-trace)
        with rng("optim.adam(fDx)"):
            optim.adam(self.fDx, self.parametersD, self.optimStateD)
        with rng("optim.adam(fGx)"):
            if (self.defer_adam_g and self.side_g is not None) or self.adam_overlap:
                self.fGx(self.parametersG)
                self._pending_g = True
            elif self._fuse_adam_ranges():
                try:
                    self.fGx(self.parametersG)
                    optim.adam_update_fused(self.parametersG, self.gradParametersG, self.optimStateG, self.netG, self.fuse_adam == "keep")
                finally:
                    self.netG.set_fused_adam(False)        # (closures called outside step() accumulate the plain way)
            else:
                optim.adam(self.fGx, self.parametersG, self.optimStateG)

    def _fuse_adam_possible(self, dp=None):
        """dp False: the single-device step(); True: the phased data-parallel step (operands gathered instead of gradients
        reduced); None: whichever of the two this trainer is set up for"""
        from .cnet import CNet
        if dp is None:
            dp = self._comm_on()
        # (the library's net object; `fused_adam_emulated`: a test double of the same protocol on a host without the kernel — tests/)
        hosted = (isinstance(self.netG, CNet) and self.netG._net is not None) or getattr(self.netG, "fused_adam_emulated", False)
        ok = (self.fuse_adam != "off" and hosted and not self.shard_adam and not self.defer_adam_g and not self.adam_overlap)
        if dp and self.dp_fused == "reduced":
            return False
        return ok and (self._comm_on() and not self._pipelined if dp else not self._comm_on())

    def _dp_rows(self):
        """the row-sharded form of the data-parallel fused update: asked for, and every rank can be dealt a block of >= 64 rows of each
        fused tensor.  Adam's m and v of the fused slices are then current in this rank's rows ONLY, which optimStateG records
        (`row_shard` = (rank, world)): continuing in another mode or world size on such a state would silently use stale moments for
        (N - 1) / N of 92 % of the generator's weights — refused until gather_adam_state() has made them whole."""
        rows = self.dp_fused == "rows" and self.world > 1 and bool(self._dpf) and self.netG.fused_adam_rows_ok(self.world)
        mark = self.optimStateG.get("row_shard")
        if mark is not None and (not rows or tuple(mark) != (self.rank, self.world)) and bool(self._dpf):
            raise RuntimeError("optimStateG holds Adam moments sharded by weight rows for rank %d of %d; this step would run as %s on rank "
                               "%d of %d: call gather_adam_state() first" % (mark[0], mark[1], "rows" if rows else self.dp_fused, self.rank, self.world))
        return rows

    def _exchange_rows(self, defer):
        """rows mode: every rank's updated row block of the fused slices to everybody.  defer: issued on the exchange stream, the
        generator's next forward waits for it in front of its bottleneck conv (vf_net_forward_wait_fused) — its first five convolutions
        run beside the transfer; a host without that hook (the module-by-module mirror, a process group instead of vf_comm_*) waits
        here.  Captured phase graphs cannot wait for an event recorded outside their capture: they wait here too."""
        B = get_backend()
        hs = []
        for per_rank in self._row_ranges or []:
            hs += B.all_gather_ranges(self.parametersG, per_rank, self.rank, self.group, async_op=True)
        self._rows_stale = False
        comm = getattr(B, "comm", None)
        tickets = [h.ticket for h in hs if hasattr(h, "ticket")]
        hook = defer and comm is not None and self._graphs is None and hasattr(self.netG, "forward_wait_fused") and len(tickets) == len(hs)
        if hook:
            self.netG.forward_wait_fused(comm, tickets)
        else:
            for h in hs:
                h.wait()
        if hs or self._row_ranges:
            bump_param_version(self.parametersG)
        return hook

    def gather_adam_state(self):
        """after steps in rows mode Adam's m and v of the fused slices are current in each rank's own rows only; this makes them whole
        on every rank (2 x 262 MB on the wire for train.lua's generator: for checkpoints of the optimiser and mode changes, not per step)"""
        if self.optimStateG.get("row_shard") is None:
            return
        B = get_backend()
        for key in ("m", "v"):
            for per_rank in self._row_ranges or []:
                B.all_gather_ranges(self.optimStateG[key], per_rank, self.rank, self.group)
        del self.optimStateG["row_shard"]

    def _fuse_adam_ranges(self):
        """marks netG's bottleneck pair for the fused update when step() may use it; the slices it covers (empty: plain update)"""
        return self.netG.set_fused_adam(True) if self._fuse_adam_possible(dp=False) else []

    def fuse_adam_slices(self):
        """[(lo, hi)] of the flat generator vectors that this trainer's step updates inside the weight-gradient kernel (empty: none)"""
        if not self._fuse_adam_possible():
            return []
        r = self.netG.set_fused_adam(True)
        if not (self._comm_on() and self._dpf):        # (a phased data-parallel step keeps its marks from phase B to phase C)
            self.netG.set_fused_adam(False)
        return r

    def fused_adam_ranges(self):
        """[(lo, hi)] of gradParametersG that step() leaves unwritten (fuse_adam == "on"); for readers of the gradient vector"""
        return self.fuse_adam_slices() if self.fuse_adam == "on" else []

    # -- Adam(G) beside the next iteration's encoder forward (single device).  92 % of the generator's parameters are
    #    the two bottleneck weight tensors (E6, D1: 32.8 M each); the first layer that reads either is E6.  The update of
    #    those two ranges runs on a side stream while the main stream updates everything else, sweeps the biases and
    #    runs E1..E5 (MFMA-bound convolutions next to an HBM-bound stream); the main stream joins in front of E6.
    #    Element for element the same arithmetic as optim.adam; `flush()` applies a pending update in one piece.
    #    Measured (configs[1], same box): 3.36 -> 3.50 ms per step — the 2048-block HBM stream slows the convolutions it
    #    shares the chip with by more than it hides — so this stays opt-in (`capture(adam_overlap=True)`).
    def enable_adam_overlap(self, on=True, min_numel=4 << 20):
        assert self._graph is None and self._graphs is None, "enable_adam_overlap before capture()"
        self.adam_overlap = False
        from .cnet import CNet
        if not on or self._comm_on() or not hasattr(get_backend(), "fork") or isinstance(self.netG, CNet):
            return self
        plan = self.netG._plan or self.netG._build_plan()
        big = [(m, o, n) for m, name, gname, o, n in self.netG._flat[2] if name == "weight" and n >= min_numel]
        idxs = [i for i, (pm, _) in enumerate(plan) if any(pm is m for m, _, _ in big)]
        if not big or len(idxs) != len(big):
            return self                    # no big tensor, or one sits inside a table member: keep the plain update
        if self.side_a is None:
            self.side_a = get_backend().fork(workspace_bytes=1 << 20)
        self._g_big = ([(o, o + n) for _, o, n in big], min(idxs))
        self.adam_overlap = True
        # the split update finishes on a side stream in mid-forward: let netG refresh its own weight planes at the start of
        # each call instead (the two side-stream tensors are the bottleneck weights, which no planes kernel reads)
        self.netG.set_weight_planes_managed(False)
        return self

    def _pending_then_forward(self, fwd):
        """apply a deferred Adam(G), sweep the conv biases (train.lua:279), run netG's forward"""
        if self._pending_g and self.adam_overlap:
            ranges, idx = self._g_big
            optim.adam_update_split(self.parametersG, self.gradParametersG, self.optimStateG, ranges, self.side_a)
            self._pending_g = False
            self.netG.zeroConvBiases()
            return fwd(before=(idx, self.side_a.join))
        self._apply_pending_g_and_sweep()
        return fwd()

    def _apply_pending_g(self):
        if self._pending_g:
            optim.adam_update(self.parametersG, self.gradParametersG, self.optimStateG)
            self._pending_g = False
            self.netG.refresh_weight_planes()

    def _apply_pending_g_and_sweep(self):
        if self._pending_g:
            self._apply_pending_g()
            self.netG.zeroConvBiases()

    def _wait_inflight(self):
        for h in self._inflight:
            if h is not None:
                h.wait()
        self._inflight = []

    def flush(self):
        """Apply a deferred Adam(G).  A captured graph always begins with that update, so after a flush the graph
        must not be replayed again (capture anew instead).  Rows mode: the exchange of the last iteration's row blocks, which the next
        iteration would have started with (replays after it are fine: they begin with an exchange of identical rows)."""
        if self._rows_stale:
            self._exchange_rows(defer=False)
        if self._pending_g:
            assert not self.shard_adam, "shard_adam has no deferred full-vector update to flush"
            self._wait_inflight()
            self._apply_pending_g()
            if (self._graph is not None and (self.defer_adam_g or self.adam_overlap)) or (self._graphs is not None and self._pipelined):
                self._graph_stale = True

    # -- the same iteration cut at the gradient exchanges (data parallel):
    #    A | all-reduce D | B | all-reduce G[tail] ∥ B2 | all-reduce G[head] | C
    # G's gradient goes out in two buckets: the tail of the flat vector (bottleneck conv + decoder, complete once the
    # backward pass has come down to the bottleneck) is on the wire while the encoder's backward convs still run.
    def _phase_a(self):
        self._defer_comm = True
        self.fDx(self.parametersD)
        self._defer_comm = False

    def _phase_b(self):
        optim.adam_update(self.parametersD, self.gradParametersD, self.optimStateD)
        self._defer_comm = True
        self._split_g = self.netG.bucket_split()
        # the bottleneck pair (92 % of G's gradient bytes) does not go on the wire: its weight gradients are left to phase C, which
        # forms them from every rank's OPERANDS — packed here into this rank's segment of the gather buffer (6 MB at batchSize 64
        # against 262 MB of gradient) and all-gathered by the step
        self._dpf = self.netG.set_fused_adam(True) if self._fuse_adam_possible(dp=True) else []
        try:
            self.fGx(self.parametersG)
            if self._dpf:
                seg = self.netG.fused_adam_pack_size()
                if self._opbuf is None or self._opbuf.numel() != seg * self.world:
                    self._opbuf = get_backend().zeros(seg * self.world)
                self.netG.fused_adam_pack(self._opbuf[self.rank * seg:(self.rank + 1) * seg])
        finally:
            self._split_g = None
            self._defer_comm = False

    def _exchange_ranges(self, lo, hi):
        """[(a, b)] of the flat generator gradient inside [lo, hi) that still travel as gradients (everything but the fused slices)"""
        out, pos = [], lo
        for a, b in sorted(self._dpf or []):
            a, b = max(a, lo), min(b, hi)
            if a >= b:
                continue
            if a > pos:
                out.append((pos, a))
            pos = max(pos, b)
        if hi > pos:
            out.append((pos, hi))
        return out

    def _phase_b2(self):
        self.netG.backward_finish()      # the encoder's weight / bias gradients, while the tail bucket is on the wire

    def _phase_c(self):
        if self.shard_adam:
            lo, hi = self._shard
            optim.adam_update(self.parametersG[lo:hi], self.gradParametersG[lo:hi], self.optimStateG_shard)
            return
        if self._dpf:
            try:
                optim.adam_update_fused(self.parametersG, self.gradParametersG, self.optimStateG, self.netG, self.fuse_adam == "keep",
                                        gathered=(self._opbuf, self.world), rows=(self.rank, self.world) if self._dp_rows() else None)
            finally:
                self.netG.set_fused_adam(False)
            return
        optim.adam_update(self.parametersG, self.gradParametersG, self.optimStateG)

    def step_phased(self):
        B = get_backend()
        pa, pb, pb2, pc = ([g.replay for g in self._graphs] if self._graphs is not None
                           else [self._phase_a, self._phase_b, self._phase_b2, self._phase_c])
        if self._graphs is not None:
            self._refresh_planes_written_outside_the_graph()
        _, off = self.netG.bucket_split()
        gG = self.gradParametersG
        if self._rows_stale:                 # the row blocks the last phase C updated: on the wire beside this iteration's first layers
            self._exchange_rows(defer=True)
        pa()
        B.all_reduce_avg(self.gradParametersD, self.world, self.group)
        pb()
        if self.shard_adam:
            pb2()
            B.reduce_scatter_avg(gG, self.world, self.rank, self.group)       # this rank's shard of the mean gradient
            pc()                                                             # Adam on that shard
            B.all_gather_shards(self.parametersG, self.world, self.rank, self.group)
            return
        hs = []
        if self._dpf:
            hs.append(B.all_gather_shards(self._opbuf, self.world, self.rank, self.group, async_op=True))
        hs += [B.all_reduce_avg(gG[a:b], self.world, self.group, async_op=True) for a, b in self._exchange_ranges(off, gG.numel())]
        pb2()
        hs += [B.all_reduce_avg(gG[a:b], self.world, self.group, async_op=True) for a, b in self._exchange_ranges(0, off)]
        for h in hs:
            if h is not None:
                h.wait()
        rows = self._dp_rows()               # (asked before phase C clears the marks)
        if rows and self._row_ranges is None:
            self._row_ranges = self.netG.fused_adam_row_ranges(self.world)
        pc()
        if rows:                             # every rank updated ITS rows of the pair: the exchange opens the next iteration (or flush())
            self._rows_stale = True
            self.optimStateG["row_shard"] = (self.rank, self.world)

    # -- pipelined data-parallel iteration: G's exchange and Adam move into the NEXT iteration, behind netD's real pass
    #    A1: netD real pass | wait G buckets (i-1) | A2: Adam(G) (i-1), netG forward, netD fake pass | all-reduce D |
    #    B: Adam(D), fGx down to the bottleneck | all-reduce G tail (async) ∥ B2: encoder backward | all-reduce G head (async)
    # The 262 MB tail bucket is then on the wire during the encoder backward AND the next iteration's netD real pass
    # (neither reads generator state).  Same arithmetic on every buffer in the same order as step_phased();
    # flush() completes the last iteration (waits, applies Adam(G)).
    def _phase_a1(self):
        self._defer_comm = True
        self._gen = self._fDx_gen()
        next(self._gen)

    def _phase_a2(self):
        for _ in self._gen:
            pass
        self._gen = None
        self._defer_comm = False

    def step_pipelined(self):
        B = get_backend()
        assert self._pipelined, "call capture_phased(pipelined=True) or set _pipelined before the first step"
        assert not self.shard_adam, "shard_adam belongs to the un-pipelined step (step_phased)"
        pa1, pa2, pb, pb2 = ([g.replay for g in self._graphs] if self._graphs is not None
                             else [self._phase_a1, self._phase_a2, self._phase_b, self._phase_b2])
        assert not self._graph_stale, "flush() was called: the captured graphs would apply Adam(G) twice; capture again"
        if self._graphs is not None:
            self._refresh_planes_written_outside_the_graph()
        _, off = self.netG.bucket_split()
        gG = self.gradParametersG
        pa1()
        self._wait_inflight()
        pa2()
        if self._graphs is not None:
            self._pending_g = False            # the replayed A2 applied it
        B.all_reduce_avg(self.gradParametersD, self.world, self.group)
        pb()
        h_tail = B.all_reduce_avg(gG[off:], self.world, self.group, async_op=True)
        pb2()
        h_head = B.all_reduce_avg(gG[:off], self.world, self.group, async_op=True) if off > 0 else None
        self._inflight = [h_tail, h_head]
        self._pending_g = True

    # -- HIP graph of one whole iteration (single-GPU): zero launch gaps, no host work per step
    def capture(self, warmup=3, defer_adam_g=False, adam_overlap=False):
        assert not self._comm_on(), "one graph covers the single-device iteration; use capture_phased() for DP"
        B = get_backend()
        if adam_overlap:
            self.enable_adam_overlap(True)
        # optional: rotate Adam(G) into the next iteration's netD-real window (measured: +2% on the video nets, nothing
        # on train.lua's — the side stream that must also run netG's forward becomes the critical path)
        self.defer_adam_g = bool(defer_adam_g) and self.side_g is not None
        for _ in range(warmup):
            self.step()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            B.use_current_stream()
            self.step()
        B.use_current_stream()
        self._graph = g
        self._graph_stale = False
        self._pending_g = self.defer_adam_g or self.adam_overlap      # the graph leaves the last iteration's Adam(G) pending
        return g

    def capture_phased(self, warmup=3, pipelined=False):
        """Four graphs (phases A, B, B2, C — or A1, A2, B, B2 when pipelined) with the RCCL all-reduces launched between
        them.  SyncBN puts collectives inside the phases, so it runs eagerly instead."""
        B = get_backend()
        if pipelined and self.batch_d:
            self.set_batch_d(False)
        self._pipelined = bool(pipelined)
        step = self.step_pipelined if pipelined else self.step_phased
        for _ in range(max(warmup, 1)):
            step()
        torch.cuda.synchronize()
        if pipelined:
            self._wait_inflight()            # the captured A2 starts with the pending Adam(G)
            phases = (self._phase_a1, self._phase_a2, self._phase_b, self._phase_b2)
        else:
            phases = (self._phase_a, self._phase_b, self._phase_b2, self._phase_c)
        graphs = []
        pool = None
        for phase in phases:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, pool=pool, capture_error_mode="thread_local"):   # RCCL's watchdog thread may poll events meanwhile
                B.use_current_stream()
                phase()
            B.use_current_stream()
            pool = g.pool()
            graphs.append(g)
        self._graphs = tuple(graphs)
        self._graph_stale = False
        if pipelined:
            # the capture recorded (did not execute) one iteration: G's gradients of the last warm-up step are still
            # pending and the next replay of A2 applies them
            self._pending_g = True
            self._inflight = []
        return self._graphs

    def capture_dp(self, warmup=3):
        """The un-pipelined data-parallel iteration (step_phased: A | all-reduce D | B | all-reduce G tail ∥ B2 | all-reduce G
        head | C) as ONE HIP graph, collectives included: vf_comm_* enqueues them on the communicator's stream behind an event
        of the compute stream and joins with another, which a capture records as a fork / join of the graph — so the tail
        bucket still travels beside the encoder's backward kernels, and nothing is launched from the host per iteration.
        Needs the C-ABI exchange (backend.comm); replay() runs it."""
        B = get_backend()
        assert self._comm_on() and getattr(B, "comm", None) is not None, "capture_dp: attach a communicator first (backend.init_comm)"
        assert not self._pipelined and not any(getattr(m, "sync_world", 1) > 1 for m in self.netD.leaves())
        for _ in range(max(warmup, 1)):
            self.step_phased()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            B.use_current_stream()
            self.step_phased()
        B.use_current_stream()
        self._graph = g
        self._graph_stale = False
        self._pending_g = False
        return g

    def _refresh_planes_written_outside_the_graph(self):
        """fDx refreshes netD's weight planes only when the host-side parameter version says they are stale, and that test is
        evaluated once, at capture time.  A parameter write OUTSIDE the graph (load_reference_flat: a checkpoint load) bumps the
        version afterwards: refresh eagerly, in front of the replay."""
        for net in (self.netD, self.netG):
            if getattr(net, "_wp_managed", False) and net._wp_stale():
                net.refresh_weight_planes()

    def replay(self):
        assert not self._graph_stale, "flush() was called: the captured graph would apply Adam(G) twice; capture() again"
        self._refresh_planes_written_outside_the_graph()
        self._graph.replay()

    def losses(self):
        self.flush()
        out = dict(errD=self.errD, errG=self.errG, errG_l2=self.errG_l2, errG_gdl=self.errG_gdl)
        return {k: (None if v is None else float(v)) for k, v in out.items()}


class CenterTrainer(_TrainerBase):
    """train.lua: centre-square inpainting; netD judges the 64x64 centre."""

    def __init__(self, opt=None, seed=1234, world=1, rank=0, group=None, sync_bn=False, fuse=True, lazy_zero=True,
                 skip_dead_grads=True, overlap=True, shard_adam=False, host=None):
        o = dict(DEFAULT_OPT_TRAIN)
        o.update(opt or {})
        self.opt = o
        sm = bool(o.get("smooth", False))
        self.netG = build_netG(o["nc"], o["nc"], o["nef"], o["ngf"], o["nBottleneck"], False, fuse, lazy_zero, sm,
                               noise_nz=o["nz"] if o["noiseGen"] else 0)
        self.netD = build_netD(o["nc"], o["ndf"], False, fuse, lazy_zero, sm, conditionAdv=bool(o["conditionAdv"]))
        if o["conditionAdv"] and skip_dead_grads:
            self.netD.modules[0].skip_grad = (0,)     # nobody reads the gradient w.r.t. the context (train.lua:371)
        if fuse:
            self.netG, self.netD, self.host = _host_nets(host, self.netG, self.netD, world if sync_bn else 1)
        else:
            self.host = "mirror"
        self.criterionMSE = nn.MSECriterion() if o["wtl2"] != 0 else None
        self._finish_init(seed, world, rank, group, sync_bn, skip_dead_grads, overlap and not (world > 1 and sync_bn), shard_adam)
        if o["conditionAdv"] and self.batch_d:
            self.set_batch_d(False)                   # a table-input netD keeps its two passes
        self.real_label, self.fake_label = 1, 0
        self.input_ctx = self.input_center = self.input_real_center = self._real_center = None
        self.noise = self._noise_fixed = None
        self.noise_seed = seed
        if o["noiseGen"]:
            optim.adam_init(self.parametersD, self.optimStateD)   # the noise draw is keyed by D's step counter

    def set_noise(self, noise):
        """Fix the generator's noise input (tests); None: a fresh draw per iteration (train.lua:319-323)."""
        self._noise_fixed = None if noise is None else to_nhwc(get_backend().from_host(noise).float())

    def _d_in(self, center):
        """netD's input: {input_ctx, input_center} when conditionAdv (train.lua:300-301)"""
        return [self.input_ctx, center] if self.opt["conditionAdv"] else center

    def _g_in(self):
        return [self.input_ctx, self.noise] if self.opt["noiseGen"] else self.input_ctx

    def _netG_forward(self, **kw):
        o = self.opt
        if o["noiseGen"]:                     # regenerate random noise (train.lua:319-323)
            if self._noise_fixed is not None:
                self.noise = self._noise_fixed
            else:
                Bn = self.input_ctx.shape[0]
                if self.noise is None or self.noise.shape[0] != Bn:
                    self.noise = get_backend().empty_act(Bn, o["nz"], 1, 1)
                get_backend().noise_fill(self.noise, self.noise_seed, normal=o["noisetype"] == "normal",
                                         counter_dev=self.optimStateD["t_dev"])
        return self.netG.forward(self._g_in(), **kw)

    def set_batch(self, real_ctx):
        """What train.lua:284-298 does on the loader's batch (a B x nc x fineSize x fineSize tensor in [-1,1]):
        clone the centre crop, paint the hole (minus the overlap band) with the channel means, and copy to the
        device buffers input_ctx / input_center / input_real_center."""
        o = self.opt
        # persistent device buffers: a captured graph keeps reading the tensors it was captured with, so a new batch
        # is written INTO them.  input_center / input_real_center are views of the protocol's names onto buffers that
        # already hold the data (the reference copies host tensors into them; here the data is on the device already)
        prev = (self.input_ctx, self._real_center) if self.input_ctx is not None else None
        self.input_ctx, self._real_center = data.center_prepare(real_ctx, o["overlapPred"], out=prev)
        self.input_real_center = self._real_center
        self.input_center = self._real_center
        self._refresh_cat_real(self._real_center)

    def _fDx_gen(self):
        """fDx as a generator that yields once, at the point where the generator net's parameters are first needed:
        everything before it (netD's real pass) is independent of netG, so a pipelined data-parallel step lets the
        previous iteration's gradient exchange run until there (`_TrainerBase.step_pipelined`)."""
        B, o = get_backend(), self.opt
        early_g = self.side_g is not None and not self._pipelined
        if not self._pending_g:
            self.netD.zeroConvBiasesWith(self.netG)      # both nets' sweeps (train.lua:279-280), one launch
        else:
            self.netD.zeroConvBiases()                   # (netG's follows its deferred Adam step)
        self.netD.zeroGradParameters()
        if self.netD._wp_stale():                  # (fGx refreshed netD's planes after optim.adam(fDx) moved its weights: stale only on
            self.netD.refresh_weight_planes()      #  the first iteration and after a checkpoint load)
        if not self._pending_g:
            self.netG.refresh_weight_planes()      # netG: updated by the previous iteration's optim.adam(fGx)
        # netG's forward does not depend on netD's real pass: issue it on a side stream (same arithmetic)
        fake = None
        if early_g:
            with self.side_g.on():
                if self._pending_g:               # deferred Adam(G) of the previous iteration, then the bias sweep
                    self._apply_pending_g()
                    self.netG.zeroConvBiases()
                fake = self._netG_forward()
        if self.batch_d:
            assert not self._pipelined
            if fake is None:
                fake = self._pending_then_forward(self._netG_forward)
            else:
                self.side_g.join()
            self.input_center = fake
            self.errD = self._netD_both(self._real_center, fake)
            self._allreduce_avg(self.gradParametersD)
            return
        # train with real (input_center:copy(real_center): the centre crop already sits in a device buffer)
        self.input_center = self._real_center
        label = self.real_label
        output = self.netD.forward(self._d_in(self.input_center))
        errD_real = self.criterion.forward(output, label)
        df_do = self.criterion.backward(output, label)
        self.netD.backward(self._d_in(self.input_center), df_do, need_input_grad=not self.skip_dead_grads)
        yield "generator parameters needed"
        # train with fake
        if fake is None:
            fake = self._pending_then_forward(self._netG_forward)
        else:
            self.side_g.join()
        self.input_center = fake               # input_center:copy(fake): netD reads the generator's output buffer
        label = self.fake_label
        output = self.netD.forward(self._d_in(self.input_center))
        errD_fake = self.criterion.forward(output, label)
        df_do = self.criterion.backward(output, label)
        self.netD.backward(self._d_in(self.input_center), df_do, need_input_grad=not self.skip_dead_grads)
        self.errD = errD_real + errD_fake
        self._allreduce_avg(self.gradParametersD)

    def fGx(self, x):
        B, o = get_backend(), self.opt
        wt, ov = o["wtl2"], o["overlapPred"]
        self.netD.zeroConvBiasesWith(self.netG)      # both nets' sweeps (train.lua:279-280), one launch
        self.netG.zeroGradParameters()
        self.netD.refresh_weight_planes()            # optim.adam(fDx) has just moved netD's weights
        label = self.real_label                      # fake labels are real for the generator cost
        output = self._netD_stale_output()           # reused from fDx (train.lua:363): stale w.r.t. D's Adam step
        self.errG, df_do = self.criterion.forward_backward(output, label)
        df_dg = self._netD_grad_input(self._d_in(self.input_center), df_do)
        if o["conditionAdv"]:
            df_dg = df_dg[1]                         # df_dg[2] because conditional GAN (train.lua:371)
        errG_total = self.errG
        if wt != 0:
            # train.lua:377-399 fused into one pass: MSE forward, MSE gradient, overlap-band weighting, wtl2 mix
            alpha = (1 - wt) if (0 < wt < 1) else 1.0
            if ov == 0:
                c0, c1, band = wt, 0.0, 0
            else:
                c0, c1, band = wt, 10 * wt - wt, ov          # inside: wtl2 ; border band: 10*wtl2
            slot = self.criterionMSE.next_slot()
            B.recon_grad_mix(df_dg, self.input_center, self.input_real_center, None, alpha, c0, c1, band, slot)
            self.errG_l2 = nn.DeviceScalar.of(slot)
            errG_total = (alpha if (0 < wt < 1) else 1.0) * self.errG + wt * self.errG_l2
        self._backward_G(df_dg)
        return errG_total, self.gradParametersG


class VidTrainer(_TrainerBase):
    """train_vid_weighted.lua / train_wholeim_input.lua: full-frame output, netD judges the whole frame."""

    def __init__(self, opt=None, seed=1234, world=1, rank=0, group=None, sync_bn=False, fuse=True, lazy_zero=True,
                 skip_dead_grads=True, overlap=True, shard_adam=False, host=None):
        o = dict(DEFAULT_OPT_VID)
        o.update(opt or {})
        self.opt = o
        nc = o["nc"] * o["predLen"]
        self.nc_in = o["nc_in"] or nc
        self.nc_out = o["nc_out"] or nc
        sm = bool(o.get("smooth", False))
        # logoNet: train_logo_withmask.lua:95-98 (last decoder stage ngf -> ngf/2 -> nc); that script's closures are
        # this class's with predLen = 1, weight_nomask = 1 (weights of ones) and wtgdl = 0
        self.netG = build_netG(self.nc_in, self.nc_out, o["nef"], o["ngf"], o["nBottleneck"], True, fuse, lazy_zero, sm,
                               half_last=bool(o.get("logoNet", False)), extra_bottleneck_stage=_ext256(o))
        self.netD = build_netD(self.nc_out, o["ndf"], True, fuse, lazy_zero, sm, extra_last_layer=_ext256(o))
        if fuse:
            self.netG, self.netD, self.host = _host_nets(host, self.netG, self.netD, world if sync_bn else 1)
        else:
            self.host = "mirror"
        self.netI = None                 # withInit: set_initializer(net) (train_vid_weighted.lua:260-264)
        self._ctx_filled = None
        self.criterionMSE = nn.MSECriterion() if o["wtl2"] != 0 else None
        self.criterionGDL = nn.GDLCriterion(1) if o["wtgdl"] != 0 else None
        self._finish_init(seed, world, rank, group, sync_bn, skip_dead_grads, overlap and not (world > 1 and sync_bn), shard_adam)
        self.real_label, self.fake_label = 1, 0
        self.input_inpainted = None

    def set_batch(self, real_ctx, real_full, real_mask):
        """The loader contract (datavid/dataset.lua:426): masked clip, full clip, Byte mask, all B x nc x H x W."""
        B = get_backend()
        new = [to_nhwc(B.from_host(t).float()) for t in (real_ctx, real_full, real_mask)]     # Byte mask -> Float (:394)
        new = [n.clone() if (torch.is_tensor(t) and n.data_ptr() == t.data_ptr()) else n       # never adopt the caller's storage
               for n, t in zip(new, (real_ctx, real_full, real_mask))]
        if self.input_inpainted is not None and all(a.shape == b.shape for a, b in zip((self._real_ctx, self._real_full, self._real_mask), new)):
            for dst, src in zip((self._real_ctx, self._real_full, self._real_mask), new):
                dst.copy_(src)       # persistent buffers: a captured graph keeps reading them
        else:
            self._real_ctx, self._real_full, self._real_mask = new
            self._inpaint_buf = torch.empty_like(self._real_full)
        # the protocol's names, as views onto buffers that already hold the data
        self.input_ctx, self.input_real, self.input_mask = self._real_ctx, self._real_full, self._real_mask
        self.input_inpainted = self._inpaint_buf
        self._refresh_cat_real(self.input_real)

    def set_initializer(self, netI):
        """opt.withInit: `netI = util.load(opt.initName)` (train_vid_weighted.lua:260-264).  The script never calls
        netI:evaluate(), so the net runs as loaded — training mode, batch statistics — and so it does here."""
        assert self._graph is None and self._graphs is None, "set_initializer before capture()"
        self.netI = netI
        return self

    def _g_in(self):
        return self._ctx_filled if self.netI is not None else self.input_ctx

    def _fill_from_initializer(self):
        """fake_init = netI:forward(input_ctx); input_ctx = inpainter.fillIn(input_ctx, input_mask, fake_init)
        (train_vid_weighted.lua:401-405).  The filled clip goes to its own buffer: the loader's batch stays intact
        for a replayed graph."""
        if self.netI is None:
            return
        fake_init = self.netI.forward(self.input_ctx)
        assert fake_init.shape == self.input_ctx.shape, "inpaint_utils.fillIn: src and dst must have the same size"
        if self._ctx_filled is None or self._ctx_filled.shape != self.input_ctx.shape:
            self._ctx_filled = torch.empty_like(self.input_ctx)
        get_backend().masked_compose(self._ctx_filled, self.input_ctx, fake_init, self.input_mask)

    def _fDx_gen(self):
        B, o = get_backend(), self.opt
        early_g = self.side_g is not None and not self._pipelined
        if not self._pending_g:
            self.netD.zeroConvBiasesWith(self.netG)      # both nets' sweeps (train.lua:279-280), one launch
        else:
            self.netD.zeroConvBiases()                   # (netG's follows its deferred Adam step)
        self.netD.zeroGradParameters()
        if self.netD._wp_stale():                  # (fGx refreshed netD's planes after optim.adam(fDx) moved its weights: stale only on
            self.netD.refresh_weight_planes()      #  the first iteration and after a checkpoint load)
        if not self._pending_g:
            self.netG.refresh_weight_planes()      # netG: updated by the previous iteration's optim.adam(fGx)
        self._fill_from_initializer()
        fake = None
        if early_g:                            # netG forward beside netD's real pass (independent work)
            with self.side_g.on():
                if self._pending_g:
                    self._apply_pending_g()
                    self.netG.zeroConvBiases()
                fake = self.netG.forward(self._g_in())
        if self.batch_d:
            assert not self._pipelined
            if fake is None:
                fake = self._pending_then_forward(lambda **kw: self.netG.forward(self._g_in(), **kw))
            else:
                self.side_g.join()
            if o["weight_nomask"] == 0:
                self.input_inpainted = self._inpaint_buf
                B.masked_compose(self.input_inpainted, self.input_real, fake, self.input_mask)
            else:
                self.input_inpainted = fake
            self.errD = self._netD_both(self.input_real, self.input_inpainted)
            self._allreduce_avg(self.gradParametersD)
            return
        label = self.real_label
        output = self.netD.forward(self.input_real)
        errD_real = self.criterion.forward(output, label)
        df_do = self.criterion.backward(output, label)
        self.netD.backward(self.input_real, df_do, need_input_grad=not self.skip_dead_grads)
        yield "generator parameters needed"
        if fake is None:
            fake = self._pending_then_forward(lambda **kw: self.netG.forward(self._g_in(), **kw))
        else:
            self.side_g.join()
        if o["weight_nomask"] == 0:                  # train_vid_weighted.lua:429-432
            self.input_inpainted = self._inpaint_buf
            B.masked_compose(self.input_inpainted, self.input_real, fake, self.input_mask)
        else:
            self.input_inpainted = fake        # input_inpainted:copy(fake): netD reads the generator's output buffer
        label = self.fake_label
        output = self.netD.forward(self.input_inpainted)
        errD_fake = self.criterion.forward(output, label)
        df_do = self.criterion.backward(output, label)
        self.netD.backward(self.input_inpainted, df_do, need_input_grad=not self.skip_dead_grads)
        self.errD = errD_real + errD_fake
        self._allreduce_avg(self.gradParametersD)

    def fGx(self, x):
        B, o = get_backend(), self.opt
        wt, lam, wtgdl = o["wtl2"], o["weight_nomask"], o["wtgdl"]
        self.netD.zeroConvBiasesWith(self.netG)      # both nets' sweeps (train.lua:279-280), one launch
        self.netG.zeroGradParameters()
        self.netD.refresh_weight_planes()            # optim.adam(fDx) has just moved netD's weights
        label = self.real_label
        output = self._netD_stale_output()
        self.errG, df_do = self.criterion.forward_backward(output, label)
        df_dg = self._netD_grad_input(self.input_real, df_do)
        errG_total = self.errG
        if wtgdl != 0:                               # forward value only (train_vid_weighted.lua:524)
            self.errG_gdl = self.criterionGDL.forward(self.input_inpainted, self.input_real)
        if wt != 0:
            assert o["overlapPred"] == 0, "train_vid_weighted.lua:499: overlapPred must be 0 here"
            # :489-503 (+ :525-528, which adds wtgdl * the MSE gradient — the reference's own quirk) in one pass:
            #   g_l2 = (2/N)(x-t) .* (mask*(1-lambda)+lambda);  df_dg = alpha*df_dg + wtl2*g_l2 + wtgdl*(2/N)(x-t)
            alpha = (1 - wt) if (0 < wt < 1) else 1.0
            if lam == 0:
                c0, c1, mask = wt + wtgdl, 0.0, None
            else:
                c0, c1, mask = wt * lam + wtgdl, wt * (1 - lam), self.input_mask
                # the reference also rewrites input_mask in place into the weights (:494); nothing reads it
                # again before the next fDx overwrites it, so the fused pass leaves it untouched.
            slot = self.criterionMSE.next_slot()
            B.recon_grad_mix(df_dg, self.input_inpainted, self.input_real, mask, alpha, c0, c1, 0, slot)
            self.errG_l2 = nn.DeviceScalar.of(slot)
            errG_total = (alpha if (0 < wt < 1) else 1.0) * self.errG + wt * self.errG_l2
        elif wtgdl != 0:
            raise RuntimeError("wtgdl ~= 0 with wtl2 == 0 indexes a nil criterionMSE in the reference (:525)")
        if wtgdl != 0:
            errG_total = errG_total + wtgdl * self.errG_gdl
        self._backward_G(df_dg)
        return errG_total, self.gradParametersG
