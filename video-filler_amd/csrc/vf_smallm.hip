// vf_smallm.hip — the bottleneck GEMMs at a small batch: M = batchSize <= 8 rows against a weight matrix of hundreds of MB
// (train_wholeim_input.lua at batchSize 4: conv nef*8 -> 6400 on a 4x4 map, 24576 x 6400 weights = 629 MB, and the full-conv back;
// forward and data-gradient of each: four passes per iteration).
//
// 2 M flops per weight: the job is reading the weights once.  The tiled matrix-core kernel of vf_conv.hip (64 x 128 tiles,
// LDS stages, a barrier per K step, split-K) ran these passes at 3.4-3.8 TB/s; here a wave streams weight rows with 16-byte
// loads straight into fp32 FMAs — no LDS, no barrier, no matrix core (its tile would be 4 live rows of 32) — and leaves split-K
// slabs in the layout vf_conv.hip's combine kernels already take (bias, activation, derivative mask, BatchNorm statistics):
//
//   row-dot  (weights [N][K], K contiguous: conv forward on the 4x4 map, full-conv data-gradient)
//            y[b][n] = sum_k A[b][k] W[n][k]: a wave owns 8 weight rows and one K range; per step of 256 floats every lane
//            holds a float4 of each of the 8 rows and of the M activation rows (L2 hits, shared by the 8 rows), 32 M FMAs;
//            one cross-lane reduction at the end.
//   axpy     (weights [K][N], N contiguous: full-conv forward from the 1x1 map, conv data-gradient)
//            y[b][col] += A[b][c] W[c][col]: a wave owns 256 columns (a float4 per lane) and one range of c; the M
//            activation values of a c are wave-uniform (scalar loads), the weight row segment is one coalesced 1 KB load.
//
// fp32 products and sums (at least as accurate as the three-plane matrix-core mode these passes otherwise use); in the
// bf16-operand mode both operands are rounded to bf16 on their way in, as everywhere.  Deterministic: fixed summation order.
#include <algorithm>
#include <cstdlib>

#include "vf_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

__device__ __forceinline__ float sm_rne(float f) {
  const unsigned u = __float_as_uint(f);
  return __uint_as_float(((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16) << 16);
}
__device__ __forceinline__ f32x4 sm_rne4(f32x4 v) {
  f32x4 o = {sm_rne(v[0]), sm_rne(v[1]), sm_rne(v[2]), sm_rne(v[3])};
  return o;
}

// ---- row-dot: slab[ks][b][n] = sum_{k in range ks} A[b][k] W[n][k]
// grid: (N / 8 row groups) x ksplit waves, four waves per block (consecutive row groups of the same K range)
template <int MT>
__global__ __launch_bounds__(256) void k_smallm_rowdot(const float* __restrict__ A, const float* __restrict__ W, float* __restrict__ slab,
                                                       int M, int N, int K, int steps_per_split, int ngroups, int rb) {
  constexpr int R = 8;
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));      // (wave-uniform, and known to be)
  const int ks = wv / ngroups, grp = wv - ks * ngroups;
  const int n0 = grp * R;
  const int64_t k0 = (int64_t)ks * steps_per_split * 256 + 4 * lane;
  const float* wp = W + (int64_t)n0 * K + k0;
  const float* ap = A + k0;
  float acc[R][MT];
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int b = 0; b < MT; ++b) acc[r][b] = 0.f;
  struct Stage { f32x4 w[R], x[MT]; };
  // (every load of a stage is unconditional — activation rows past M read row 0 and are zeroed by a multiply, the stage after the
  //  last re-reads the last one: straight-line stages whose wait counts are plain to see)
  float xm[MT];
#pragma unroll
  for (int b = 0; b < MT; ++b) xm[b] = b < M ? 1.f : 0.f;
  auto load = [&](int s, Stage& st) {
    const int64_t o = (int64_t)s * 256;
#pragma unroll
    for (int b = 0; b < MT; ++b) st.x[b] = *(const f32x4*)(ap + (int64_t)(b < M ? b : 0) * K + o);
#pragma unroll
    for (int r = 0; r < R; ++r) st.w[r] = __builtin_nontemporal_load((const f32x4*)(wp + (int64_t)r * K + o));      // (read once)
  };
  auto fma = [&](Stage& st) {
#pragma unroll
    for (int b = 0; b < MT; ++b) st.x[b] = st.x[b] * xm[b];
    if (rb) {
#pragma unroll
      for (int b = 0; b < MT; ++b) st.x[b] = sm_rne4(st.x[b]);
#pragma unroll
      for (int r = 0; r < R; ++r) st.w[r] = sm_rne4(st.w[r]);
    }
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int b = 0; b < MT; ++b)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[r][b] = fmaf(st.w[r][e], st.x[b][e], acc[r][b]);
  };
  Stage s0, s1;
  load(0, s0);
  const int last = steps_per_split - 1;
  for (int s = 0; s < steps_per_split; s += 2) {          // (steps_per_split is even: vf_internal_smallm_plan)
    load(s + 1, s1);
    __builtin_amdgcn_sched_barrier(0);
    fma(s0);
    __builtin_amdgcn_sched_barrier(0);
    load(min(s + 2, last), s0);                            // (past the end: the last stage once more, never consumed)
    __builtin_amdgcn_sched_barrier(0);
    fma(s1);
    __builtin_amdgcn_sched_barrier(0);
  }
  // the 64 lanes' partial sums meet in LDS: value v = r * MT + b of this wave is summed over the lanes, in lane order, by lane v.
  // (History, DESIGN.md 4.9: the first form of this kernel ended in a __shfl_xor butterfly with all 32 results stored by lane 0,
  //  and gave run-to-run different slabs when other processes shared the GPU.  Round 3 blamed the butterfly; round 4 found the
  //  cause in the FMA phase above: storing PAIRS OF ROWS from one lane made the compiler pack the chains as
  //  `v_pk_fma_f32 acc[r:r+1], w[r:r+1], x.hi ... op_sel:[0,1,0]`, a form that on gfx950 occasionally drops its low result in lanes
  //  48-63 under GPU sharing (checker kernel: the per-lane partials in front of the reduction equal the chain with exactly that
  //  FMA missing; the same instruction with src0/src1 exchanged, or split into two v_fma_f32, never failed).  With this epilogue
  //  the compiler packed pairs of activation rows instead (`op_sel:[1,0,0]`); since the end of round 4 the whole library is built
  //  with -packed-fp32-ops (build.py) and these chains are plain v_fma_f32 — scripts/check_pk_opsel.py asserts, in the CPU suite,
  //  that the shipped code objects hold no packed FP32 arithmetic at all.)
  __shared__ float red[4][R * MT][65];
  const int wl = threadIdx.x >> 6;
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int b = 0; b < MT; ++b) red[wl][r * MT + b][lane] = acc[r][b];
  __syncthreads();
  if (lane < R * MT) {
    float t = 0.f;
#pragma unroll 8
    for (int j = 0; j < 64; ++j) t += red[wl][lane][j];
    const int r = lane / MT, bb = lane - r * MT;
    if (bb < M) slab[(int64_t)ks * M * N + (int64_t)bb * N + n0 + r] = t;
  }
}

// ---- axpy: slab[ks][b][col] = sum_{c in range ks} A[b][c] W[c][col]
// grid: (N / 256 column groups) x ksplit waves, four waves per block (consecutive column groups of the same c range)
template <int MT>
__global__ __launch_bounds__(256) void k_smallm_axpy(const float* __restrict__ A, const float* __restrict__ W, float* __restrict__ slab,
                                                     int M, int N, int C, int c_per_split, int ngroups, int rb) {
  constexpr int CH = 8;                       // weight rows in flight per stage
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));      // (so that A[b][c] below is a scalar load)
  const int ks = wv / ngroups, grp = wv - ks * ngroups;
  const int col = grp * 256 + 4 * lane;
  const int c0 = ks * c_per_split;
  const float* wp = W + (int64_t)c0 * N + col;
  f32x4 acc[MT];
#pragma unroll
  for (int b = 0; b < MT; ++b) acc[b] = f32x4{0.f, 0.f, 0.f, 0.f};
  struct Stage { f32x4 w[CH]; };
  auto load = [&](int c, Stage& st) {
#pragma unroll
    for (int j = 0; j < CH; ++j) st.w[j] = __builtin_nontemporal_load((const f32x4*)(wp + (int64_t)(c + j) * N));      // (read once)
  };
  auto fma = [&](int c, Stage& st) {
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      f32x4 w = st.w[j];
      if (rb) w = sm_rne4(w);
#pragma unroll
      for (int b = 0; b < MT; ++b) {
        float a = A[(int64_t)(b < M ? b : 0) * C + c0 + c + j];       // wave-uniform: a scalar load (rows past M: row 0, unused)
        if (rb) a = sm_rne(a);
        acc[b] += a * w;
      }
    }
  };
  Stage s0, s1;
  load(0, s0);
  const int lastc = c_per_split - CH;
  for (int c = 0; c < c_per_split; c += 2 * CH) {        // (c_per_split is a multiple of 2 * CH: vf_internal_smallm_plan)
    load(c + CH, s1);
    __builtin_amdgcn_sched_barrier(0);
    fma(c, s0);
    __builtin_amdgcn_sched_barrier(0);
    load(min(c + 2 * CH, lastc), s0);                     // (past the end: the last stage once more, never consumed)
    __builtin_amdgcn_sched_barrier(0);
    fma(c + CH, s1);
    __builtin_amdgcn_sched_barrier(0);
  }
  float* o = slab + (int64_t)ks * M * N + col;
#pragma unroll
  for (int b = 0; b < MT; ++b)
    if (b < M) *(f32x4*)(o + (int64_t)b * N) = acc[b];
}

}  // namespace

// Plans a small-M pass.  form 0: row-dot (W = [N][K]); form 1: axpy (W = [K][N]).  Returns the split count (>= 2), or 0 when the
// shape is not these kernels' (the caller keeps its tiled path).
int vf_internal_smallm_plan(int form, int M, int N, int K, size_t ws_bytes) {
  static const bool off = getenv("VF_NO_SMALLM") && atoi(getenv("VF_NO_SMALLM"));
  // (M = 16, train_vid_weighted.lua's batch: the axpy form took 84 us per 131 MB pass against 37 us for the tiled kernel —
  //  64 FMAs and 16 scalar operands per 16 bytes of weights make it issue-bound; the row-dot form would hold 128 accumulators)
  if (off || M < 1 || M > 8) return 0;
  static const int want_waves = getenv("VF_SMALLM_WAVES") ? atoi(getenv("VF_SMALLM_WAVES")) : 4096;      // ~4 per SIMD
  int ksplit = 0;
  if (form == 0) {
    if (K % 256 != 0 || N % 32 != 0) return 0;          // (four 8-row groups per block)
    const int steps = K / 256, groups = N / 8;
    for (int s = 2; s <= steps; ++s)
      if (steps % s == 0 && (steps / s) % 2 == 0) {      // whole double steps per split
        ksplit = s;
        if ((int64_t)groups * s >= want_waves) break;
      }
  } else {
    if (N % 1024 != 0 || K % 16 != 0) return 0;          // (four 256-column groups per block)
    const int chunks = K / 16, groups = N / 256;
    for (int s = 2; s <= chunks; ++s)
      if (chunks % s == 0) {
        ksplit = s;
        if ((int64_t)groups * s >= want_waves) break;
      }
  }
  if (ksplit < 2 || (size_t)ksplit * M * N * sizeof(float) > ws_bytes) return 0;
  return ksplit;
}

int vf_internal_smallm_launch(vf_ctx* ctx, int form, const float* A, const float* W, float* slab, int M, int N, int K, int ksplit) {
  const int rb = ctx->mfma_bf16 == 1;
  const double flops = 2.0 * M * (double)N * K, bytes = 4.0 * ((double)N * K + (double)M * K + (double)ksplit * M * N);
  if (form == 0) {
    const int groups = N / 8, steps = K / 256 / ksplit;
    const dim3 grid((unsigned)((int64_t)groups * ksplit / 4));
    if (M <= 4)
      VF_LAUNCH_TIMED(ctx, "smallm_rowdot", flops, bytes, k_smallm_rowdot<4>, grid, dim3(256), A, W, slab, M, N, K, steps, groups, rb);
    else
      VF_LAUNCH_TIMED(ctx, "smallm_rowdot", flops, bytes, k_smallm_rowdot<8>, grid, dim3(256), A, W, slab, M, N, K, steps, groups, rb);
  } else {
    const int groups = N / 256, cps = K / ksplit;
    const dim3 grid((unsigned)((int64_t)groups * ksplit / 4));
    if (M <= 4)
      VF_LAUNCH_TIMED(ctx, "smallm_axpy", flops, bytes, k_smallm_axpy<4>, grid, dim3(256), A, W, slab, M, N, K, cps, groups, rb);
    else
      VF_LAUNCH_TIMED(ctx, "smallm_axpy", flops, bytes, k_smallm_axpy<8>, grid, dim3(256), A, W, slab, M, N, K, cps, groups, rb);
  }
  VF_LAUNCH_CHECK();
  return 0;
}
