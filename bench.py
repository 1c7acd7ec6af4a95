#!/usr/bin/env python3
"""bench.py — whole-iteration throughput of the context-encoder GAN hot path on MI355X.

Workload (BASELINE.json configs[1]): train.lua nets with the README `inpaintCenter` recipe
(nBottleneck=4000, wtl2=0.999, overlapPred=4, fineSize=128), batchSize=64 per GPU, synthetic U[-1,1] images
resident in HBM.  A step = fDx + Adam(D) + fGx + Adam(G) (train.lua:421-424): netG 1 fwd + 1 bwd, netD 2 fwd +
2 bwd + 1 data-grad pass, BCE/MSE criteria, two fused Adam passes.  Tensors and accumulators are fp32; conv products come from an exact
3-plane bf16 split of the fp32 operands on the bf16 matrix pipe by default (--mfma f32 = native f32 MFMA).

  python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run, one rank per GPU)

Prints ONE JSON line (rank 0).  `roofline` is for the kernel with the largest share of the step, timed with HIP
events inside the library during an instrumented pass of the same workload; `cpu_baseline` is the CPU oracle
(1 thread, as the reference forces torch.setnumthreads(1)) on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_BF16_MFMA_TFLOPS = 2500.0   # dense bf16 peak (the opt-in --mfma bf16 mode prices against this)
PEAK_HBM_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E spec peak


def csrc_digest():
    """sha256 over the kernel sources (csrc/*.hip, *.h, include/vf_hip.h): ties a PMC summary to the build it was taken from"""
    import glob
    import hashlib
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(ROOT, "video-filler_amd", "csrc", "*")) + [os.path.join(ROOT, "include", "vf_hip.h")])
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()


def _pytorch_cpu_row(batch, nBottleneck, threads):
    """train.lua's nets (87-199) and closures (278-410) in stock PyTorch on the CPU: whole iterations (fDx + Adam, fGx + Adam) on a
    synthetic batch for about five seconds.  Timing only — parity lives with the oracle."""
    import torch
    import torch.nn as tnn
    torch.set_num_threads(max(1, threads))
    nef = ngf = ndf = 64

    def conv(i, o, s2=True):
        return tnn.Conv2d(i, o, 4, 2, 1) if s2 else tnn.Conv2d(i, o, 4)

    def full(i, o, s2=True):
        return tnn.ConvTranspose2d(i, o, 4, 2, 1) if s2 else tnn.ConvTranspose2d(i, o, 4)
    G = tnn.Sequential(conv(3, nef), tnn.LeakyReLU(0.2, True),
                       conv(nef, nef), tnn.BatchNorm2d(nef), tnn.LeakyReLU(0.2, True),
                       conv(nef, nef * 2), tnn.BatchNorm2d(nef * 2), tnn.LeakyReLU(0.2, True),
                       conv(nef * 2, nef * 4), tnn.BatchNorm2d(nef * 4), tnn.LeakyReLU(0.2, True),
                       conv(nef * 4, nef * 8), tnn.BatchNorm2d(nef * 8), tnn.LeakyReLU(0.2, True),
                       conv(nef * 8, nBottleneck, False), tnn.BatchNorm2d(nBottleneck), tnn.LeakyReLU(0.2, True),
                       full(nBottleneck, ngf * 8, False), tnn.BatchNorm2d(ngf * 8), tnn.ReLU(True),
                       full(ngf * 8, ngf * 4), tnn.BatchNorm2d(ngf * 4), tnn.ReLU(True),
                       full(ngf * 4, ngf * 2), tnn.BatchNorm2d(ngf * 2), tnn.ReLU(True),
                       full(ngf * 2, ngf), tnn.BatchNorm2d(ngf), tnn.ReLU(True),
                       full(ngf, 3), tnn.Tanh())
    D = tnn.Sequential(conv(3, ndf), tnn.LeakyReLU(0.2, True),
                       conv(ndf, ndf * 2), tnn.BatchNorm2d(ndf * 2), tnn.LeakyReLU(0.2, True),
                       conv(ndf * 2, ndf * 4), tnn.BatchNorm2d(ndf * 4), tnn.LeakyReLU(0.2, True),
                       conv(ndf * 4, ndf * 8), tnn.BatchNorm2d(ndf * 8), tnn.LeakyReLU(0.2, True),
                       conv(ndf * 8, 1, False), tnn.Sigmoid(), tnn.Flatten())
    optD = torch.optim.Adam(D.parameters(), lr=2e-4, betas=(0.5, 0.999))
    optG = torch.optim.Adam(G.parameters(), lr=2e-3, betas=(0.5, 0.999))
    bce, mse = tnn.BCELoss(), tnn.MSELoss()
    g = torch.Generator().manual_seed(1)
    ctx = torch.rand((batch, 3, 128, 128), generator=g) * 2 - 1
    center = ctx[:, :, 32:96, 32:96].clone()
    ones, zeros = torch.ones(batch, 1), torch.zeros(batch, 1)

    def step():
        optD.zero_grad(set_to_none=True)
        fake = G(ctx)
        errD = bce(D(center), ones) + bce(D(fake.detach()), zeros)
        errD.backward()
        optD.step()
        optG.zero_grad(set_to_none=True)
        errG = 0.001 * bce(D(fake), ones) + 0.999 * mse(fake, center)
        errG.backward()
        optG.step()
    step()                                  # (allocations, oneDNN primitive creation)
    t0, n = time.perf_counter(), 0
    while n < 8 and (n == 0 or time.perf_counter() - t0 < 5.0):
        step()
        n += 1
    dt = time.perf_counter() - t0
    return dict(value=round(n * batch / dt, 3), unit="images/s", cores=threads, kind="pytorch %s CPU, not the reference's code" % torch.__version__,
                sample="%d iterations of the same nets and losses at batchSize=%d, %.1f s" % (n, batch, dt))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=None, help="batchSize per GPU (train.lua:7); default per workload: center 64 "
                    "(configs[1]), vid16 / vid4 16 (configs[2]), wholeim 4 (configs[4]: global 32 over 8 GPUs)")
    ap.add_argument("--nBottleneck", type=int, default=None, help="default 4000 (README recipes); wholeim: 6400 (the script's own)")
    ap.add_argument("--workload", default="center", choices=["center", "vid16", "vid4", "wholeim"],
                    help="center = configs[1] (the headline); vid16 = configs[2]; wholeim = configs[4] "
                    "(train_wholeim_input.lua defaults, 27 -> 12 channels, 192/192/128, wtgdl 0.5; add --mfma bf16 for its bf16 form)")
    ap.add_argument("--step-stats", type=int, default=100, help="steps of the per-step percentile pass after the timed region "
                    "(0: skip it, e.g. under rocprofv3 --pmc)")
    ap.add_argument("--no-graph", action="store_true", help="time eager launches instead of a HIP graph (N=1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-batch", type=int, default=8)
    ap.add_argument("--no-torch-cpu-baseline", action="store_true", help="skip the optional third baseline row (the same iteration in stock "
                    "PyTorch on the host cores, ~6 s; configs[1] only)")
    ap.add_argument("--sync-bn", action="store_true", help="N>1: all-reduce BatchNorm sums (big-batch parity mode)")
    ap.add_argument("--fine-size", type=int, default=128, choices=[128, 256], help="wholeim only: 256 = the labelled NON-PARITY "
                    "extension opt.ext256 (the reference's nets fail at that size: one more stride-2 stage in netD and around "
                    "netG's bottleneck)")
    ap.add_argument("--force-dist", action="store_true", help="N=1: run the data-parallel path (RCCL init + all-reduce) anyway")
    ap.add_argument("--pipeline", action="store_true", help="N>1: the pipelined data-parallel step: G's gradient exchange and Adam(G) run "
                    "behind the NEXT iteration's netD real pass — which then has to stay a separate pass (no 2B netD batching: "
                    "-26 %% per GPU before any communication, VERDICT r2).  Default: the un-pipelined step with the single-device "
                    "iteration's batching; G's tail bucket travels beside the encoder backward")
    ap.add_argument("--no-pipeline", action="store_true", help="(default now; kept so that older command lines still parse)")
    ap.add_argument("--dp-graph", default="phased", choices=["phased", "one"], help="N>1 graph form: phased = four HIP graphs with the "
                    "collectives launched between them from the host (default); one = the whole iteration INCLUDING the collectives "
                    "as one HIP graph (vf_comm_* on its own stream, recorded as a fork / join) — needs --comm cabi/auto to succeed")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend; 'gloo' + --share-gpu rehearses the N>1 control "
                    "flow with several ranks on ONE GPU (not a measurement)")
    ap.add_argument("--shard-adam", action="store_true", help="N > 1: reduce-scatter G's gradient, Adam on 1/N of the parameters per "
                    "rank, all-gather the updated shards (un-pipelined step)")
    ap.add_argument("--comm", default="auto", choices=["auto", "cabi", "torch"], help="N > 1 over RCCL: auto (default) = the exchange "
                    "through the C-ABI (vf_comm_*) if every rank can bring it up AND its start-up self-check of every collective "
                    "passes on every rank, else — decided by all ranks together — torch.distributed's process group; cabi = "
                    "vf_comm_* or fail; torch = the process group")
    ap.add_argument("--host", default="cabi", choices=["cabi", "mirror"], help="who drives netG / netD: cabi (default) = the library's own "
                    "net object, one vf_net_* call per Torch7 method with the whole fast path inside libvf_hip.so (what a Lua host gets "
                    "through hipnn.Net); mirror = the module-by-module Python mirror of the nn protocol (every layer call crosses the "
                    "C-ABI on its own).  Same kernels, same plan")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal: every rank uses cuda:0")
    ap.add_argument("--mfma", default="f32_3xbf16", choices=["f32_3xbf16", "f32", "bf16"],
                    help="how conv products are formed: f32_3xbf16 (default; fp32 operands split exactly into 3 bf16 planes, 6 "
                    "cross terms on the bf16 pipe, fp32 accumulate: fp32-grade), f32 (native f32 MFMA), bf16 (operands rounded: "
                    "opt-in, its own tolerance, not a headline configuration)")
    ap.add_argument("--no-batch-d", action="store_true", help="N=1: run netD's real and fake passes separately (default: one batch "
                    "of 2B with BatchNorm in two groups; same arithmetic per sample)")
    ap.add_argument("--batch-d", action="store_true", help="(default now at N=1; kept so that older command lines still parse)")
    ap.add_argument("--fuse-adam", default="on", choices=["on", "keep", "off"], help="N=1: optim.adam(fGx) applied to the two bottleneck weight "
                    "tensors inside the kernel that forms their gradient (vf_wgrad_adam_outer: 24 B per weight instead of 32). on (default): "
                    "gradParametersG does not receive those two slices; keep: it does (28 B); off: accGradParameters + the plain update")
    ap.add_argument("--dp-fused", default="rows", choices=["gathered", "rows", "reduced"], help="N>1: what becomes of the bottleneck "
                    "pair's gradient (262 of G's 284 MB): gathered (default) = every rank all-gathers the pair's OPERANDS (6 MB per rank) and "
                    "forms the global-batch gradient of all rows inside the fused update; rows = the same gather, each rank updates its "
                    "1/N of the rows and the updated rows are all-gathered; reduced = the gradients are all-reduced like the rest")
    ap.add_argument("--adam-overlap", action="store_true", help="N=1: update Adam(G)'s two bottleneck weight tensors on a side stream "
                    "beside the next encoder forward (measured: -4 %%: the 2048-block HBM stream slows the convolutions it shares "
                    "the chip with by more than it hides)")
    ap.add_argument("--overlap", action="store_true", help="3 streams (dW beside dX, netG forward beside netD's real pass): measured "
                    "+0.7 %% on one GPU with the current kernels (noise level), so the default is one stream")
    ap.add_argument("--no-overlap", action="store_true", help="(default now; kept so that older command lines still parse)")
    args = ap.parse_args()
    if args.batch is None:
        args.batch = {"center": 64, "vid16": 16, "vid4": 16, "wholeim": 4}[args.workload]
    if args.nBottleneck is None:
        args.nBottleneck = 6400 if args.workload == "wholeim" else 4000

    # stdout carries exactly ONE JSON line: libraries that print banners there (RCCL does at init) go to stderr
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus %d must be launched with torch.distributed.run (one rank per GPU)" % args.gpus)
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1 or args.force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if args.backend != "nccl":          # rehearsal of the control flow (several ranks on one GPU): torch.distributed
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    from video_filler_amd.backend import get_backend
    from video_filler_amd.trainers import CenterTrainer, VidTrainer

    B = get_backend()
    store = None
    if (world > 1 or args.force_dist) and args.backend == "nccl":
        # the exchange is the C-ABI's (vf_comm_*: RCCL inside libvf_hip.so); the only host-side rendezvous is moving
        # rank 0's 128-byte id, through a TCP store at MASTER_ADDR:MASTER_PORT — no torch.distributed process group.
        # (--comm torch, or a failure to bring the communicator up, runs the same iteration over torch.distributed's RCCL
        #  process group instead: the trainers call the backend's all_reduce either way.)
        if args.comm in ("auto", "cabi"):
            from video_filler_amd.backend import bring_up_comm
            ok, store = bring_up_comm(B, world, rank)       # every rank gets the same answer (decided through the store)
            if not ok and args.comm == "cabi":
                raise SystemExit("bench.py --comm cabi: vf_comm_* could not be brought up and verified on every rank")
            elif not ok:
                sys.stderr.write("bench.py: vf_comm_* not available/verified on every rank; all ranks use torch.distributed\n")
        if B.comm is None:
            kw = {}
            if store is not None:      # the rendezvous store already exists (bring_up_comm): the process group shares it
                kw["store"] = dist.PrefixStore("vf_pg", store)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank), **kw)
    global PEAK_F32_MFMA_TFLOPS
    B.set_mfma_mode(args.mfma)
    if args.mfma == "bf16":
        PEAK_F32_MFMA_TFLOPS = PEAK_BF16_MFMA_TFLOPS      # the roofline of this run is the bf16 matrix pipe
    gen = torch.Generator().manual_seed(1234 + rank)
    if args.workload == "center":
        opt = dict(batchSize=args.batch, nBottleneck=args.nBottleneck, wtl2=0.999, overlapPred=4)
        tr = CenterTrainer(opt, seed=1234, world=world, rank=rank, sync_bn=args.sync_bn, overlap=args.overlap, shard_adam=args.shard_adam, host=args.host)
        batch = torch.rand((args.batch, 3, 128, 128), generator=gen) * 2 - 1
        tr.set_batch(batch)
        wl = "train.lua inpaintCenter (nBottleneck=%d wtl2=0.999 overlapPred=4) fineSize=128 batchSize=%d/GPU" % (
            args.nBottleneck, args.batch)
    elif args.workload == "wholeim":
        # train_wholeim_input.lua:39-43 defaults: 3x3 array of patches in (27 channels), 2x2 out (12), nef = ngf = 192,
        # ndf = 128, nBottleneck 6400; wtgdl 0.5 exercises the GDL value path; fineSize 128
        fs = args.fine_size
        opt = dict(batchSize=args.batch, nBottleneck=args.nBottleneck, nc_in=27, nc_out=12, nef=192, ngf=192, ndf=128,
                   weight_nomask=1, wtgdl=0.5, fineSize=fs, ext256=(fs == 256))
        tr = VidTrainer(opt, seed=1234, world=world, rank=rank, sync_bn=args.sync_bn, overlap=args.overlap, shard_adam=args.shard_adam, host=args.host)
        full = torch.rand((args.batch, 12, fs, fs), generator=gen) * 2 - 1
        mask = torch.zeros((args.batch, 12, fs, fs), dtype=torch.uint8)
        mask[:, :, fs // 4:3 * fs // 4, fs // 4:3 * fs // 4] = 1
        ctx = torch.rand((args.batch, 27, fs, fs), generator=gen) * 2 - 1
        ctx[:, :, fs // 4:3 * fs // 4, fs // 4:3 * fs // 4] = 2 * (110.0 / 255.0) - 1
        tr.set_batch(ctx, full, mask)
        wl = ("train_wholeim_input.lua 27->12 channels nef=ngf=192 ndf=128 nBottleneck=%d wtgdl=0.5 fineSize=%d%s "
              "batchSize=%d/GPU" % (args.nBottleneck, fs, " (NON-PARITY extension ext256: the reference's nets fail at this size)"
                                    if fs == 256 else "", args.batch))
    else:
        predLen = 16 if args.workload == "vid16" else 4
        nc = 3 * predLen
        opt = dict(batchSize=args.batch, nBottleneck=args.nBottleneck, predLen=predLen)
        tr = VidTrainer(opt, seed=1234, world=world, rank=rank, sync_bn=args.sync_bn, overlap=args.overlap, shard_adam=args.shard_adam, host=args.host)
        full = torch.rand((args.batch, nc, 128, 128), generator=gen) * 2 - 1
        mask = torch.zeros((args.batch, nc, 128, 128), dtype=torch.uint8)
        mask[:, :, 32:96, 32:96] = 1
        ctx = full.clone()
        ctx[mask != 0] = 2 * (110.0 / 255.0) - 1
        tr.set_batch(ctx, full, mask)
        wl = "train_vid_weighted.lua predLen=%d (nc=%d) nBottleneck=%d fineSize=128 batchSize=%d/GPU" % (
            predLen, nc, args.nBottleneck, args.batch)

    dp = world > 1 or args.force_dist
    tr.force_comm = args.force_dist
    tr.fuse_adam = args.fuse_adam
    tr.dp_fused = args.dp_fused
    if args.no_batch_d and tr.batch_d:
        tr.set_batch_d(False)
    use_graph = not args.no_graph and not (dp and args.sync_bn)
    pipelined = dp and args.pipeline and not args.sync_bn and not args.shard_adam
    one_graph = dp and use_graph and args.dp_graph == "one" and not pipelined and B.comm is not None
    if dp:
        if pipelined and tr.batch_d:
            tr.set_batch_d(False)
        if one_graph:
            tr.capture_dp(warmup=max(args.warmup, 2))
            run = tr.replay
        else:
            if use_graph:
                tr.capture_phased(warmup=max(args.warmup, 2), pipelined=pipelined)
            else:
                tr._pipelined = pipelined
            run = tr.step_pipelined if pipelined else tr.step_phased
        for _ in range(max(args.warmup, 10) if use_graph else args.warmup):
            run()
    elif use_graph:
        tr.capture(warmup=max(args.warmup, 2), adam_overlap=args.adam_overlap)
        run = tr.replay
        for _ in range(max(args.warmup, 10)):     # untimed replays: lets clocks settle on a fresh box
            run()
    else:
        run = tr.step
        for _ in range(args.warmup):
            run()

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            if B.comm is not None:
                B.comm_barrier()        # every rank's streams have drained and every rank is here (host-blocking)
            else:
                dist.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        B.all_reduce(t, op="max")           # vf_comm_allreduce_inline (or torch.distributed in the gloo rehearsal)
        dt = float(t.item())
    # per-step distribution (median and p10/p90 over >= 100 iterations): a separate pass after the timed
    # region, one event pair per step on the launch stream, nothing synchronises inside it
    step_stats = None
    if rank == 0 and world == 1 and args.step_stats > 0:
        nd = args.step_stats
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(nd + 1)]
        evs[0].record()
        for i in range(nd):
            run()
            evs[i + 1].record()
            # eager steps are ~270 launches each: bound what is in flight (a graph replay is ONE launch and needs no bound).
            # 100 un-synchronised eager steps = 27 k queued dispatches; under rocprofv3 --pmc that ended in a SIGSEGV inside
            # the runtime's dispatch path (profiler-side per-dispatch state, not this library)
            if not use_graph and (i & 7) == 7:
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        ts = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(nd))
        step_stats = dict(n=nd, p10=round(ts[nd // 10], 4), p50=round(ts[nd // 2], 4), p90=round(ts[(9 * nd) // 10], 4))
    losses = tr.losses()

    # ---- instrumented pass: per-kernel HIP-event timing of the same workload (eager launches)
    kernels, roofline = {}, None
    if rank == 0:
        B.use_current_stream()
        nprof = 3
        # Pass 1: same stream set-up as the timed region (with overlap on, a kernel's duration includes what it
        # shares the chip with).  Eager launches so that each one can carry its own start/stop events.
        tr._graphs = None
        tr.force_comm = False
        # rank 0 runs this pass ALONE: no collective may be reached from here on (the other ranks are waiting in the
        # final barrier), so the trainer becomes a single-device one
        tr._pipelined = False
        tr.world = 1
        for net in (tr.netG, tr.netD):
            for m in net.leaves():
                if hasattr(m, "sync_world"):
                    m.sync_world, m.sync_group = 1, None
        B.prof_begin()
        for _ in range(nprof):
            tr.step()
        kernels = B.prof_end()
        isolated = None
        if tr.side_g is not None or tr.netD.side is not None:
            # Pass 2: one stream, every kernel alone on the chip (what `--no-overlap` times)
            tr.netD.side = tr.netG.side = None
            tr.side_g = None
            B.prof_begin()
            for _ in range(nprof):
                tr.step()
            isolated = B.prof_end()
        tot = sum(k["ms"] for k in kernels.values())
        # the dominant kernel: the kernel TEMPLATE with the largest share of the step (every k_igemm<...> instantiation
        # is one kernel source; which of its symbols a layer lands on is a tiling decision), then that template's
        # largest symbol — a single symbol, so that avg_launch_us can be checked against rocprofv3's per-symbol csv
        def family(n):
            for f in ("pconv", "igemm", "wgrad", "slab_reduce", "bn", "bias_grad"):
                if n.startswith(f):
                    return f
            return n
        fam = {}
        for n, k in kernels.items():
            fam[family(n)] = fam.get(family(n), 0.0) + k["ms"]
        top_family = max(fam, key=fam.get)
        # ... and among that template's symbols the one with the most time; symbols within 5 % of the top time are a tie (which of two
        # near-equal symbols leads changes from box to box), broken towards the SLOWER rate so that the headline cannot improve by a
        # symbol swap (VERDICT r3 weak #5b).  roofline.family_weighted / worst_symbol carry the rest of the family.
        members_all = [(n, k) for n, k in kernels.items() if family(n) == top_family]
        top_ms = max(k["ms"] for _, k in members_all)

        def _rate(k):
            work = k["flops"] if k["flops"] > 0 else k["bytes"]
            return work / k["ms"] if k["ms"] > 0 else 0.0
        name, dom = min(((n, k) for n, k in members_all if k["ms"] >= 0.95 * top_ms), key=lambda kv: _rate(kv[1]))
        avg_ms = dom["ms"] / dom["launches"]
        # which roofline bounds it: a kernel that reports both its algorithmic FLOPs and bytes is priced against the LONGER of the
        # two floors (configs[4]'s batch-4 passes over a 629 MB weight tensor are matrix-core kernels bounded by the weight read)
        t_mfma = dom["flops"] / (PEAK_F32_MFMA_TFLOPS * 1e12)
        t_hbm = dom["bytes"] / (PEAK_HBM_GBS * 1e9)
        if dom["flops"] > 0 and t_mfma >= t_hbm:
            ach = dom["flops"] / dom["launches"] / (avg_ms * 1e-3) / 1e12
            roofline = dict(bound="mfma", kernel=name, achieved=round(ach, 2), peak=PEAK_F32_MFMA_TFLOPS, unit="TFLOP/s",
                            frac=round(ach / PEAK_F32_MFMA_TFLOPS, 4), traffic=None, avg_launch_us=round(avg_ms * 1e3, 2),
                            launches_per_step=dom["launches"] / nprof, share_of_step=round(dom["ms"] / tot, 3))
            # the same measurement against the pipe the kernel actually issues on.  `frac` prices the ALGORITHMIC fp32
            # FLOPs against the f32-input MFMA peak (what a native fp32 kernel is bounded by); in the default mode each
            # fp32 product is formed from 6 bf16 MFMAs, so the bf16 pipe executes 6x the algorithmic FLOPs:
            mult = {"f32_3xbf16": 6.0, "bf16": 1.0, "f32": None}[args.mfma]
            # the kernel's OWN ceiling (VERDICT r3 weak #5a): the pipe it issues on divided by the MFMAs it issues per product —
            # 2 500 / 6 = 417 TFLOP/s algorithmic in the three-plane mode, 2 500 with bf16-rounded operands, 157.3 on the f32
            # pipe.  `frac` (against the f32-input peak, the contract's "dense MFMA peak for the dtype") can exceed what a native
            # fp32 kernel could reach; `frac_of_pipe_bound` cannot exceed 1 for any kernel
            pipe_bound = PEAK_BF16_MFMA_TFLOPS / mult if mult is not None else PEAK_F32_MFMA_TFLOPS
            roofline["pipe_bound"] = round(pipe_bound, 1)
            roofline["frac_of_pipe_bound"] = round(ach / pipe_bound, 4)
            # the family as a whole and its WORST symbol, so that the headline cannot move by a symbol swap (weak #5b)
            members = {n: k for n, k in kernels.items() if family(n) == top_family and k["flops"] > 0 and k["ms"] > 0}
            f_flops, f_ms = sum(k["flops"] for k in members.values()), sum(k["ms"] for k in members.values())
            f_ach = f_flops / (f_ms * 1e-3) / 1e12
            roofline["family_weighted"] = dict(achieved=round(f_ach, 2), frac=round(f_ach / PEAK_F32_MFMA_TFLOPS, 4),
                                               frac_of_pipe_bound=round(f_ach / pipe_bound, 4), ms_per_step=round(f_ms / nprof, 4),
                                               gflop_per_step=round(f_flops / nprof / 1e9, 1))
            wn, wk = min(members.items(), key=lambda kv: kv[1]["flops"] / kv[1]["ms"])
            w_ach = wk["flops"] / (wk["ms"] * 1e-3) / 1e12
            roofline["worst_symbol"] = dict(kernel=wn, achieved=round(w_ach, 2), frac=round(w_ach / PEAK_F32_MFMA_TFLOPS, 4),
                                            frac_of_pipe_bound=round(w_ach / pipe_bound, 4),
                                            avg_launch_us=round(wk["ms"] / wk["launches"] * 1e3, 2),
                                            launches_per_step=wk["launches"] / nprof)
            if mult is not None:
                roofline["bf16_pipe"] = dict(executed_tflops=round(mult * ach, 1), peak=PEAK_BF16_MFMA_TFLOPS,
                                             frac=round(mult * ach / PEAK_BF16_MFMA_TFLOPS, 4),
                                             algorithmic_frac_of_bf16_peak=round(ach / PEAK_BF16_MFMA_TFLOPS, 4),
                                             note="executed = %gx algorithmic (bf16 MFMAs issued per fp32 product); frac = share of the "
                                                  "dense bf16 matrix peak the kernel keeps busy" % mult)
        else:
            ach = dom["bytes"] / dom["launches"] / (avg_ms * 1e-3) / 1e9
            roofline = dict(bound="hbm", kernel=name, achieved=round(ach, 1), peak=PEAK_HBM_GBS, unit="GB/s",
                            frac=round(ach / PEAK_HBM_GBS, 4), traffic=None, avg_launch_us=round(avg_ms * 1e3, 2),
                            launches_per_step=dom["launches"] / nprof, share_of_step=round(dom["ms"] / tot, 3))
            # a matrix-core family whose dominant symbol has its longer floor on the HBM side (the bf16-operand mode: one MFMA per
            # product, a 4.3 GFLOP pass is 1.7 us of pipe time): the matrix-pipe view of the same family all the same
            mult = {"f32_3xbf16": 6.0, "bf16": 1.0, "f32": None}[args.mfma]
            members = {n: k for n, k in kernels.items() if family(n) == top_family and k["flops"] > 0 and k["ms"] > 0}
            if members and mult is not None:
                pipe_bound = PEAK_BF16_MFMA_TFLOPS / mult
                if dom["flops"] > 0:
                    d_ach = dom["flops"] / dom["launches"] / (avg_ms * 1e-3) / 1e12
                    roofline["mfma_view"] = dict(achieved=round(d_ach, 2), unit="TFLOP/s", pipe_bound=round(pipe_bound, 1),
                                                 frac_of_pipe_bound=round(d_ach / pipe_bound, 4))
                f_flops, f_ms = sum(k["flops"] for k in members.values()), sum(k["ms"] for k in members.values())
                f_ach = f_flops / (f_ms * 1e-3) / 1e12
                roofline["family_weighted"] = dict(achieved=round(f_ach, 2), frac_of_pipe_bound=round(f_ach / pipe_bound, 4),
                                                   ms_per_step=round(f_ms / nprof, 4), gflop_per_step=round(f_flops / nprof / 1e9, 1))
                wn, wk = min(members.items(), key=lambda kv: kv[1]["flops"] / kv[1]["ms"])
                w_ach = wk["flops"] / (wk["ms"] * 1e-3) / 1e12
                roofline["worst_symbol"] = dict(kernel=wn, achieved=round(w_ach, 2), frac_of_pipe_bound=round(w_ach / pipe_bound, 4),
                                                avg_launch_us=round(wk["ms"] / wk["launches"] * 1e3, 2),
                                                launches_per_step=wk["launches"] / nprof)
        roofline["kernel_family"] = dict(name=top_family, share_of_step=round(fam[top_family] / tot, 3),
                                         symbols={n: round(k["ms"] / tot, 3) for n, k in kernels.items() if family(n) == top_family})
        if isolated is not None and name in isolated and isolated[name]["flops"] > 0:
            iso = isolated[name]
            iavg = iso["ms"] / iso["launches"]
            iach = iso["flops"] / iso["launches"] / (iavg * 1e-3) / 1e12
            roofline["isolated"] = dict(achieved=round(iach, 2), frac=round(iach / PEAK_F32_MFMA_TFLOPS, 4),
                                        avg_launch_us=round(iavg * 1e3, 2),
                                        note="same kernel with nothing else on the chip (single stream; bench.py --no-overlap)")
        # whole-step view: every MFMA kernel's algorithmic FLOPs over the measured step time (streams overlap, so
        # per-kernel durations under-state what the chip as a whole sustains)
        mfma_flops = sum(v["flops"] for v in kernels.values()) / nprof
        step_s = dt / args.steps
        roofline["whole_step"] = dict(mfma_gflop_per_step=round(mfma_flops / 1e9, 1), tflops=round(mfma_flops / step_s / 1e12, 2),
                                      frac=round(mfma_flops / step_s / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
                                      note="all MFMA kernels' algorithmic FLOPs / ms_per_step (includes every non-MFMA kernel's time)")
        # HBM bytes per launch from the PMC counters: collected in their own rocprofv3 passes (scripts/pmc_bench_traffic.sh,
        # committed under profiles/) — counters cannot be read from inside this run
        try:
            import glob
            # one summary per workload (scripts/pmc_bench_traffic.sh <workload>): r03_pmc_bench_traffic.json = center (configs[1]),
            # ..._vid16.json = configs[2], ..._wholeim.json = configs[4]; taken at the workload's default batch size and product mode
            suffix = "" if args.workload == "center" else "_" + args.workload
            pmc_file = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_bench_traffic%s.json" % suffix)))[-1]
            pmc_all = json.load(open(pmc_file))
            pmc = pmc_all["kernels"].get(name)
            # the counters belong to ONE build of the kernels: the summary records the digest of csrc/ it was collected
            # with (scripts/pmc_bench_traffic.py) and a summary of any other build is refused
            fresh = pmc_all.get("csrc_sha256") == csrc_digest()
            default_batch = {"center": 64, "vid16": 16, "vid4": 16, "wholeim": 4}[args.workload]
            if (pmc is not None and pmc_all.get("workload", "center") == args.workload and args.batch == default_batch
                    and args.mfma == "f32_3xbf16" and fresh):
                roofline["traffic"] = int(pmc["hbm_MB_per_launch"] * 1e6)
                roofline["traffic_unit"] = "HBM bytes per launch (FETCH_SIZE x2 + WRITE_SIZE)"
                roofline["traffic_source"] = os.path.relpath(pmc_file, ROOT)
            elif not fresh:
                roofline["traffic_note"] = "%s was collected with another build of csrc/ (digest mismatch): not used" % os.path.relpath(pmc_file, ROOT)
        except (IndexError, OSError, KeyError, ValueError):
            pass
        # per kernel: time, rates, and the fraction of the bound that applies to it — matrix-core kernels of the conv families
        # against the pipe they issue on (see pipe_bound above), byte-counted kernels against the HBM peak
        mult_k = {"f32_3xbf16": 6.0, "bf16": 1.0, "f32": None}[args.mfma]
        pipe_k = PEAK_BF16_MFMA_TFLOPS / mult_k if mult_k is not None else PEAK_F32_MFMA_TFLOPS

        def _entry(n, v):
            tf = v["flops"] / (v["ms"] * 1e-3) / 1e12 if v["flops"] > 0 and v["ms"] > 0 else None
            gb = v["bytes"] / (v["ms"] * 1e-3) / 1e9 if v["bytes"] > 0 and v["ms"] > 0 else None
            e = dict(launches_per_step=v["launches"] / nprof, ms_per_step=round(v["ms"] / nprof, 4),
                     tflops=None if tf is None else round(tf, 2), gbs=None if gb is None else round(gb, 1))
            planes = n.startswith(("pconv", "pwgrad")) or "bf16" in n
            if tf is not None:
                e["frac_of_pipe_bound"] = round(tf / (pipe_k if planes else PEAK_F32_MFMA_TFLOPS), 4)
            if gb is not None:
                e["frac_of_hbm"] = round(gb / PEAK_HBM_GBS, 4)
            return e
        kernels = {k: _entry(k, v) for k, v in sorted(kernels.items(), key=lambda kv: -kv[1]["ms"])}

    # ---- CPU baseline: the oracle, 1 thread, bounded sample
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        import numpy as np
        from oracle import oracle as O
        O.set_num_threads(1)
        cb = args.cpu_batch
        if args.workload == "center":
            ref = O.CenterTrainer(dict(nBottleneck=args.nBottleneck, wtl2=0.999, overlapPred=4), np.random.default_rng(1234))
            ref.set_batch(O.synth_center_batch(cb, np.random.default_rng(1235)))
        elif args.workload == "wholeim":
            cb = min(cb, 2)       # 31 GFLOP of convolutions per sample and iteration: two samples bound the sample at ~1 min
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            from helpers import FastRng
            ref = O.VidTrainer(dict(opt), FastRng(1234))
            ref.set_batch(*O.synth_vid_batch(cb, np.random.default_rng(1235), 27, 12))
        else:
            cb = min(cb, 2)
            predLen = 16 if args.workload == "vid16" else 4
            ref = O.VidTrainer(dict(nBottleneck=args.nBottleneck, predLen=predLen), np.random.default_rng(1234))
            ref.set_batch(*O.synth_vid_batch(cb, np.random.default_rng(1235), 3 * predLen))
        # a bounded sample of 10-30 s of CPU work: whole iterations until 10 s have passed (at most 8)
        c0 = time.perf_counter()
        nit = 0
        while nit < 8 and (nit == 0 or time.perf_counter() - c0 < 10.0):
            ref.step()
            nit += 1
        cdt = time.perf_counter() - c0
        cpu = dict(value=round(nit * cb / cdt, 3), unit="images/s" if args.workload == "center" else "clips/s", cores=1, kind="port",
                   sample="%d full iteration%s (fDx+Adam+fGx+Adam) of the same nets at batchSize=%d, %.1f s" % (nit, "s" if nit > 1 else "", cb, cdt))
        # second row (SURVEY 8(d) ii): the same iteration with OpenMP over all host cores the box gives this process
        try:
            ncores = len(os.sched_getaffinity(0))
        except AttributeError:
            ncores = os.cpu_count() or 1
        ncores = min(ncores, 16)      # a one-GPU box's CPU share (more threads than that oversubscribe: 256 ran 5x SLOWER)
        if ncores > 1:
            O.set_num_threads(ncores)
            c0 = time.perf_counter()
            ref.step()
            cdt2 = time.perf_counter() - c0
            O.set_num_threads(1)
            cpu["all_cores"] = dict(value=round(cb / cdt2, 3), unit="images/s", cores=ncores,
                                    sample="the same iteration, OpenMP over %d threads (the one-GPU box's CPU share), %.1f s" % (ncores, cdt2))

        # third row (SURVEY 8(d) iii, optional): the same train.lua iteration in stock PyTorch on the host cores — not the reference's
        # code (Torch7's CPU `nn` is what the oracle restates), a familiar yardstick beside it.  configs[1] only; bounded sample.
        if args.workload == "center" and not args.no_torch_cpu_baseline:
            try:
                cpu["pytorch_cpu"] = _pytorch_cpu_row(cb, args.nBottleneck, ncores)
            except Exception as e:      # noqa: BLE001 — a yardstick must never cost the bench line
                cpu["pytorch_cpu"] = dict(error="%s: %s" % (type(e).__name__, e))

    if rank == 0:
        n_img = world * args.batch * args.steps
        out = {
            "metric": ("netG+netD fwd+bwd images/sec, 128x128 center-mask" if args.workload == "center" else
                       "netG+netD fwd+bwd clips/sec, %dx%d (%s)" % (args.fine_size if args.workload == "wholeim" else 128,
                                                                      args.fine_size if args.workload == "wholeim" else 128,
                                                                      args.workload + (" ext256" if args.fine_size == 256 else ""))),
            "value": round(n_img / dt, 2),
            "unit": "images/s" if args.workload == "center" else "clips/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": {"f32": "f32", "f32_3xbf16": "f32",
                      "bf16": "bf16 MFMA operands, f32 accumulate/BN/criteria/Adam (opt-in mode)"}[args.mfma],
            "data": "synthetic",
            "config": {"workload": wl, "global_batch": world * args.batch, "launch": ((("one hipGraph with the bucketed all-reduces recorded in it; " if one_graph else "hipGraph x4 + bucketed RCCL all-reduce between; ") + ("G buckets in flight during the encoder backward and the next iteration's netD real pass" if pipelined else "G tail bucket in flight during the encoder backward")) if dp else "hipGraph") if use_graph else "eager", "streams": 3 if args.overlap else 1,
                       "mfma": {"f32_3xbf16": "fp32 operands split exactly into 3 bf16 planes, 6 cross terms on v_mfma_f32_32x32x16_bf16, "
                                              "f32 accumulate (fp32-grade: same parity tolerances as native)",
                                "f32": "native v_mfma_f32_32x32x2_f32", "bf16": "operands rounded to bf16"}[args.mfma],
                       "adam_G": ("bottleneck weight tensors on a side stream beside the next encoder forward" if tr.adam_overlap else
                                  ("the bottleneck pair's update inside its weight-gradient kernel (vf_wgrad_adam_outer%s), one launch for the rest"
                                   % ("; gradient also stored" if tr.fuse_adam == "keep" else "") if tr.fuse_adam_slices() else "one launch at the end of the iteration")),
                       "netD_passes": "real+fake as one batch of 2B, BatchNorm per half" if tr.batch_d else "separate (as the reference)",
                       "host": ("vf_net_* (C-ABI net object: forward / backward / updateGradInput are one library call each)" if tr.host == "cabi"
                                else "nn.py mirror (module by module over the C-ABI)"),
                       "bn": ("sync" if args.sync_bn else "local") if world > 1 else "single-device",
                       "dp_fused": (args.dp_fused if dp and tr.fuse_adam != "off" else None),
                       "exchange": (("vf_comm_* (C-ABI, RCCL; verified at start-up on every rank)" if B.comm is not None else "torch.distributed (%s)" % args.backend) if dp else None)},
            "roofline": roofline,
            "cpu_baseline": cpu,
            "step_ms": step_stats,
            "losses": losses,
            "kernels": kernels,
        }
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    if world > 1 or args.force_dist:
        # ranks > 0 wait here while rank 0 runs its instrumented pass and the CPU baseline: on the host (a store key), not
        # inside a collective whose kernel would spin on their GPUs for minutes
        if store is not None:
            if rank == 0:
                store.set("vf_bench_done", "1")
                t_end = time.time() + 120
                while world > 1 and store.add("vf_bench_left", 0) < world - 1 and time.time() < t_end:
                    time.sleep(0.05)        # the store lives in this process: stay until everyone has read the key
            elif world > 1:
                import datetime
                store.wait(["vf_bench_done"], datetime.timedelta(seconds=1800))
                store.add("vf_bench_left", 1)
        elif world > 1 and dist.is_initialized():
            dist.barrier()
        if B.comm is not None:
            B.destroy_comm()
        if dist.is_initialized():
            dist.destroy_process_group()


if __name__ == "__main__":
    main()
