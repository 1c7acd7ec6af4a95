// vf_comm.hip — the data-parallel exchange of the training iteration, behind the C-ABI.
//
// The reference has no multi-GPU path (train.lua:42 `gpu = 1`, one cutorch device); north_star's N > 1 configuration is
// plain data parallelism: every rank runs the closures on its shard, the flat gradient vectors (train.lua:240-241
// `getParameters()`) are averaged over ranks before each Adam step, and — optionally — BatchNorm's sums are added over
// ranks so the statistics are the big batch's.  Those are the only exchange steps, and they live here so that a Lua
// (or any FFI) host gets them from the same library as the kernels: one RCCL communicator per process (= per GPU), a
// dedicated stream for the gradient buckets so they travel beside the backward kernels still running on the
// context's stream, and an inline variant on the context's own stream for SyncBN's per-layer sums (which the very
// next kernel needs — nothing to overlap; being on the stream makes it part of a hipGraph capture).
//
// RCCL is bound at run time (dlopen at the first vf_comm_* call): the single-GPU library has no link dependency on it,
// and inside a process that already carries an RCCL (PyTorch's) that copy is the one used.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstdlib>
#include <cstring>

#include "vf_common.h"

namespace {

struct Rccl {
  void* h = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;      // optional (older RCCLs): NULL -> CommDestroy
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*ReduceScatter)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

Rccl g_rccl;

int rccl_bind() {
  if (g_rccl.h) return 0;
  const char* names[] = {getenv("VF_RCCL_LIB"), "librccl.so", "librccl.so.1"};
  void* h = nullptr;
  for (const char* n : names)            // a copy the process already carries (PyTorch's) comes first
    if (n && !h) h = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
  for (const char* n : names)
    if (n && !h) h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
  VF_REQUIRE(h != nullptr, "vf_comm: librccl.so not found (%s); set VF_RCCL_LIB", dlerror());
#define VF_SYM(field, name)                                                        \
  *(void**)(&g_rccl.field) = dlsym(h, name);                                       \
  VF_REQUIRE(g_rccl.field != nullptr, "vf_comm: RCCL has no symbol %s", name)
  VF_SYM(GetUniqueId, "ncclGetUniqueId");
  VF_SYM(CommInitRank, "ncclCommInitRank");
  VF_SYM(CommDestroy, "ncclCommDestroy");
  VF_SYM(AllReduce, "ncclAllReduce");
  VF_SYM(Broadcast, "ncclBroadcast");
  VF_SYM(ReduceScatter, "ncclReduceScatter");
  VF_SYM(AllGather, "ncclAllGather");
  VF_SYM(GetErrorString, "ncclGetErrorString");
#undef VF_SYM
  *(void**)(&g_rccl.CommAbort) = dlsym(h, "ncclCommAbort");
  g_rccl.h = h;
  return 0;
}

#define VF_CHECK_RCCL(expr)                                                                              \
  do {                                                                                                   \
    ncclResult_t _r = (expr);                                                                            \
    if (_r != ncclSuccess) {                                                                             \
      vf_set_error("%s:%d: %s failed: %s", __FILE__, __LINE__, #expr, g_rccl.GetErrorString(_r));        \
      return 3;                                                                                          \
    }                                                                                                    \
  } while (0)

constexpr int RING = 64;        // collectives in flight on the exchange stream before tickets wrap

}  // namespace

struct vf_comm {
  ncclComm_t comm = nullptr;
  int world = 1, rank = 0, device = 0;
  hipStream_t stream = nullptr;         // the exchange stream (gradient buckets)
  hipEvent_t ready = nullptr;           // "the producer stream has written the bucket"
  hipEvent_t done[RING] = {};           // one per ticket
  unsigned issued = 0;
};

static int dtype_of(int dtype, ncclDataType_t* out) {
  VF_REQUIRE(dtype == 0 || dtype == 1, "vf_comm: dtype %d (0 = f32, 1 = f64)", dtype);
  *out = dtype == 0 ? ncclFloat32 : ncclFloat64;
  return 0;
}

// The mean over ONE rank is the sum over one rank.  RCCL nevertheless answers ncclAvg on a one-rank communicator with its
// pre-multiply pass (a kernel that reads and writes the whole bucket to scale it by 1.0: 0.14-0.19 ms per 71 M-float bucket,
// profiles/r03_g_kernel_stats_center_force_dist_*.csv `oneRankReduce<FuncPreMulSum>`), while an in-place ncclSum there
// enqueues nothing.  With two or more ranks the factor 1/world rides in the ring's first step and costs no pass of its own.
static ncclRedOp_t avg_op(const vf_comm* c) { return c->world == 1 ? ncclSum : ncclAvg; }

static int op_of(const vf_comm* c, int op, ncclRedOp_t* out) {
  VF_REQUIRE(op >= 0 && op <= 3, "vf_comm: op %d (0 = sum, 1 = avg, 2 = max, 3 = min)", op);
  const ncclRedOp_t ops[4] = {ncclSum, avg_op(c), ncclMax, ncclMin};
  *out = ops[op];
  return 0;
}

// rank 0: 128 opaque bytes that every rank passes to vf_comm_init; the host moves them (a file, a socket, MPI, a
// torch.distributed store — the library does not care).
VF_API int vf_comm_unique_id(void* id128) {
  VF_REQUIRE(id128 != nullptr, "vf_comm_unique_id: NULL");
  static_assert(sizeof(ncclUniqueId) == 128, "VF_COMM_ID_BYTES");
  if (int rc = rccl_bind()) return rc;
  ncclUniqueId id;
  VF_CHECK_RCCL(g_rccl.GetUniqueId(&id));
  memcpy(id128, &id, sizeof(id));
  return 0;
}

// everything vf_comm_init builds after `new vf_comm`: a failure at any step returns non-zero with `c` half-built, and the
// caller releases exactly what exists (comm_release)
static int comm_build(vf_comm* c, const void* id128) {
  VF_CHECK_HIP(hipGetDevice(&c->device));
  ncclUniqueId id;
  memcpy(&id, id128, sizeof(id));
  VF_CHECK_RCCL(g_rccl.CommInitRank(&c->comm, c->world, id, c->rank));
  VF_CHECK_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  VF_CHECK_HIP(hipEventCreateWithFlags(&c->ready, hipEventDisableTiming));
  for (int i = 0; i < RING; ++i) VF_CHECK_HIP(hipEventCreateWithFlags(&c->done[i], hipEventDisableTiming));
  return 0;
}

// releases whatever of a communicator exists (also a half-built one); never touches the error message of the failure
// that brought us here
static void comm_release(vf_comm* c, bool abort_comm) {
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->comm) {
    if (abort_comm && g_rccl.CommAbort) (void)g_rccl.CommAbort(c->comm);
    else (void)g_rccl.CommDestroy(c->comm);
  }
  for (int i = 0; i < RING; ++i)
    if (c->done[i]) (void)hipEventDestroy(c->done[i]);
  if (c->ready) (void)hipEventDestroy(c->ready);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

// 0 when RCCL can be bound in this process (non-collective: a host checks this on EVERY rank, and lets the ranks agree,
// before any of them enters the collective vf_comm_init — a rank that cannot load RCCL must not leave the others waiting in it)
VF_API int vf_comm_available(void) { return rccl_bind(); }

// one communicator per process, on the CURRENT device (hipSetDevice first); collective over all `world` ranks.
// On failure nothing is leaked: the partial RCCL communicator is aborted, stream and events destroyed, *out left NULL.
VF_API int vf_comm_init(vf_comm** out, const void* id128, int world, int rank) {
  VF_REQUIRE(out != nullptr && id128 != nullptr, "vf_comm_init: NULL argument");
  VF_REQUIRE(world >= 1 && rank >= 0 && rank < world, "vf_comm_init: rank %d of %d", rank, world);
  *out = nullptr;
  if (int rc = rccl_bind()) return rc;
  vf_comm* c = new vf_comm();
  c->world = world;
  c->rank = rank;
  if (int rc = comm_build(c, id128)) {
    comm_release(c, true);
    return rc;
  }
  *out = c;
  return 0;
}

VF_API int vf_comm_world(const vf_comm* c) { return c ? c->world : 0; }
VF_API int vf_comm_rank(const vf_comm* c) { return c ? c->rank : -1; }

// In-place all-reduce of buf[0..count) on the exchange stream, ordered after everything the context's stream has been
// given so far; the context's stream is NOT held up.  *ticket names the collective for vf_comm_wait.
VF_API int vf_comm_allreduce_async(vf_comm* c, vf_ctx* ctx, void* buf, int64_t count, int dtype, int op, int* ticket) {
  VF_REQUIRE(c != nullptr && ctx != nullptr && ticket != nullptr, "vf_comm_allreduce_async: NULL argument");
  VF_REQUIRE(count >= 0 && (count == 0 || buf != nullptr), "vf_comm_allreduce_async: bad buffer");
  ncclDataType_t dt;
  ncclRedOp_t ro;
  if (int rc = dtype_of(dtype, &dt)) return rc;
  if (int rc = op_of(c, op, &ro)) return rc;
  VF_CHECK_HIP(hipEventRecord(c->ready, ctx->stream));
  VF_CHECK_HIP(hipStreamWaitEvent(c->stream, c->ready, 0));
  if (count > 0) VF_CHECK_RCCL(g_rccl.AllReduce(buf, buf, (size_t)count, dt, ro, c->comm, c->stream));
  const int t = (int)(c->issued++ % RING);
  VF_CHECK_HIP(hipEventRecord(c->done[t], c->stream));
  *ticket = t;
  return 0;
}

// the mean over ranks of a flat fp32 gradient bucket (averaged inside the collective: no extra pass over the bucket)
VF_API int vf_comm_allreduce_avg_async(vf_comm* c, vf_ctx* ctx, float* buf, int64_t n, int* ticket) {
  return vf_comm_allreduce_async(c, ctx, buf, n, 0, 1, ticket);
}

// The two halves of an all-reduce, for an optimiser sharded over the ranks: buf holds world * shard_count floats.
//   reduce_scatter_avg: rank r ends up with the MEAN over ranks of shard r, in place at buf + r * shard_count (the other
//                       shards of buf are scratch afterwards);
//   allgather:          every rank's shard r (at buf + r * shard_count) reaches all ranks, in place.
// Same stream discipline as vf_comm_allreduce_async.  With the gradient vector reduce-scattered, Adam runs on 1 / world of the
// parameters per rank and the updated shards are gathered: the bytes on the wire of one all-reduce, 1 / world of the update.
VF_API int vf_comm_reduce_scatter_avg_async(vf_comm* c, vf_ctx* ctx, float* buf, int64_t shard_count, int* ticket) {
  VF_REQUIRE(c != nullptr && ctx != nullptr && ticket != nullptr, "vf_comm_reduce_scatter_avg_async: NULL argument");
  VF_REQUIRE(shard_count >= 0 && (shard_count == 0 || buf != nullptr), "vf_comm_reduce_scatter_avg_async: bad buffer");
  VF_CHECK_HIP(hipEventRecord(c->ready, ctx->stream));
  VF_CHECK_HIP(hipStreamWaitEvent(c->stream, c->ready, 0));
  if (shard_count > 0)
    VF_CHECK_RCCL(g_rccl.ReduceScatter(buf, buf + (int64_t)c->rank * shard_count, (size_t)shard_count, ncclFloat32, avg_op(c), c->comm, c->stream));
  const int t = (int)(c->issued++ % RING);
  VF_CHECK_HIP(hipEventRecord(c->done[t], c->stream));
  *ticket = t;
  return 0;
}
VF_API int vf_comm_allgather_async(vf_comm* c, vf_ctx* ctx, float* buf, int64_t shard_count, int* ticket) {
  VF_REQUIRE(c != nullptr && ctx != nullptr && ticket != nullptr, "vf_comm_allgather_async: NULL argument");
  VF_REQUIRE(shard_count >= 0 && (shard_count == 0 || buf != nullptr), "vf_comm_allgather_async: bad buffer");
  VF_CHECK_HIP(hipEventRecord(c->ready, ctx->stream));
  VF_CHECK_HIP(hipStreamWaitEvent(c->stream, c->ready, 0));
  if (shard_count > 0)
    VF_CHECK_RCCL(g_rccl.AllGather(buf + (int64_t)c->rank * shard_count, buf, (size_t)shard_count, ncclFloat32, c->comm, c->stream));
  const int t = (int)(c->issued++ % RING);
  VF_CHECK_HIP(hipEventRecord(c->done[t], c->stream));
  *ticket = t;
  return 0;
}

// rank `root`'s buf[0..count) to every rank, on the exchange stream (same discipline as vf_comm_allgather_async).  The row blocks of a
// tensor whose rows do NOT split evenly over the ranks (vf_net_fused_adam_row_range) travel as one such broadcast per rank, where
// equal blocks travel as one all-gather.
VF_API int vf_comm_broadcast_async(vf_comm* c, vf_ctx* ctx, float* buf, int64_t count, int root, int* ticket) {
  VF_REQUIRE(c != nullptr && ctx != nullptr && ticket != nullptr, "vf_comm_broadcast_async: NULL argument");
  VF_REQUIRE(root >= 0 && root < c->world && count >= 0 && (count == 0 || buf != nullptr), "vf_comm_broadcast_async: root %d of %d, %lld floats",
             root, c->world, (long long)count);
  VF_CHECK_HIP(hipEventRecord(c->ready, ctx->stream));
  VF_CHECK_HIP(hipStreamWaitEvent(c->stream, c->ready, 0));
  if (count > 0) VF_CHECK_RCCL(g_rccl.Broadcast(buf, buf, (size_t)count, ncclFloat32, root, c->comm, c->stream));
  const int t = (int)(c->issued++ % RING);
  VF_CHECK_HIP(hipEventRecord(c->done[t], c->stream));
  *ticket = t;
  return 0;
}

// the context's stream waits (on the device; the host does not block) for collective `ticket` — and, the exchange stream
// being in order, for every collective issued before it
VF_API int vf_comm_wait(vf_comm* c, vf_ctx* ctx, int ticket) {
  VF_REQUIRE(c != nullptr && ctx != nullptr, "vf_comm_wait: NULL argument");
  VF_REQUIRE(ticket >= 0 && ticket < RING, "vf_comm_wait: ticket %d", ticket);
  VF_CHECK_HIP(hipStreamWaitEvent(ctx->stream, c->done[ticket], 0));
  return 0;
}

// In-place all-reduce ON the context's stream (SyncBN's per-layer sums: the next kernel on the stream reads them).
// Nothing else is touched, so a stream capture records it like any kernel.
VF_API int vf_comm_allreduce_inline(vf_comm* c, vf_ctx* ctx, void* buf, int64_t count, int dtype, int op) {
  VF_REQUIRE(c != nullptr && ctx != nullptr, "vf_comm_allreduce_inline: NULL argument");
  VF_REQUIRE(count >= 0 && (count == 0 || buf != nullptr), "vf_comm_allreduce_inline: bad buffer");
  ncclDataType_t dt;
  ncclRedOp_t ro;
  if (int rc = dtype_of(dtype, &dt)) return rc;
  if (int rc = op_of(c, op, &ro)) return rc;
  if (count > 0) VF_CHECK_RCCL(g_rccl.AllReduce(buf, buf, (size_t)count, dt, ro, c->comm, ctx->stream));
  return 0;
}

// rank `root`'s buf to everyone (initial weights: every replica must start from the same parameters), on the
// context's stream
VF_API int vf_comm_broadcast(vf_comm* c, vf_ctx* ctx, void* buf, int64_t count, int dtype, int root) {
  VF_REQUIRE(c != nullptr && ctx != nullptr, "vf_comm_broadcast: NULL argument");
  VF_REQUIRE(root >= 0 && root < c->world, "vf_comm_broadcast: root %d of %d", root, c->world);
  ncclDataType_t dt;
  if (int rc = dtype_of(dtype, &dt)) return rc;
  if (count > 0) VF_CHECK_RCCL(g_rccl.Broadcast(buf, buf, (size_t)count, dt, root, c->comm, ctx->stream));
  return 0;
}

// host-blocking: every rank's exchange stream and context stream have drained, and every rank has reached this call
VF_API int vf_comm_barrier(vf_comm* c, vf_ctx* ctx) {
  VF_REQUIRE(c != nullptr && ctx != nullptr, "vf_comm_barrier: NULL argument");
  float* token = nullptr;
  VF_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  VF_CHECK_HIP(hipMalloc((void**)&token, sizeof(float)));
  VF_CHECK_HIP(hipMemsetAsync(token, 0, sizeof(float), c->stream));
  VF_CHECK_RCCL(g_rccl.AllReduce(token, token, 1, ncclFloat32, ncclSum, c->comm, c->stream));
  VF_CHECK_HIP(hipStreamSynchronize(c->stream));
  VF_CHECK_HIP(hipFree(token));
  return 0;
}

VF_API int vf_comm_destroy(vf_comm* c) {
  if (!c) return 0;
  comm_release(c, false);
  return 0;
}
