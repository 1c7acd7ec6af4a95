// vf_wgrad_small.hip — weight gradients of the two bottleneck layers (train.lua:104 conv nef*8 -> nBottleneck on a 4x4 map,
// train.lua:134 full-conv nBottleneck -> ngf*8 onto a 4x4 map): THNN accGradParameters with K = batch.
//
// Both are the plain product  dW[n][col] = sum_{b < B} U[b][n] * V[b][col]  with U = [B][Nu] (the 1x1-map side: gradOutput of the
// conv, input of the full-conv) and V = [B][16 * Cv] (the 4x4-map side, whose (kh, kw, c) order IS the weight's column order):
// 4000 x 8192 outputs from K = 16 (train_vid_weighted.lua at batchSize 16), 6400 x 24576 from K = 4 (train_wholeim_input.lua).
// With a small batch the output write is the whole job — 131 MB / 629 MB per layer — and the tiled kernels of vf_conv.hip,
// built around a long K loop (LDS stages, a barrier per step, one or two blocks per CU), ran it at 3.1-3.7 TB/s where a memset
// of the same bytes reaches 6.6-7.3.  Used for K <= 32 (see vf_internal_wgrad_smallk).
//
// K-major operands are exactly what v_mfma_f32_32x32x2_f32 takes WITHOUT any transposition: lane l holds A[row l % 32][k = l / 32]
// and B[k = l / 32][col l % 32], so lanes 0-31 read 32 consecutive channels of batch row 2j, lanes 32-63 of row 2j + 1 —
// coalesced, straight from global (the operands are small and live in L2), no LDS, no barrier.  A lane loads a float2 of U
// (rows n0 + 2 r + {0, 1}: two interleaved 32-row blocks) and a float2 of V (columns c0 + 2 c + {0, 1}: two interleaved
// 32-column blocks), which makes every output store a float2 of adjacent columns: 32 lanes x 8 B = one 256-byte run.
// A wave owns 64 x 64 outputs (4 accumulator tiles, three waves per SIMD), four independent waves per block.  fp32 products, fp32 accumulation in
// batch order — the arithmetic of matrix-core mode 0, at least as accurate as the three-plane mode it stands in for.
#include <algorithm>
#include <cstdlib>

#include "vf_common.h"

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

// NJ = column blocks per wave (float2 / float4 of V per lane): 2 -> a 64 x 64 wave tile, 4 accumulator tiles (64 AGPRs: four or
// five waves per SIMD cover the L2 latency of the operand loads by themselves); 4 -> 64 x 128 (one wave per SIMD: slower, kept
// for the record)
template <int NJ>
__global__ __launch_bounds__(256) void k_wgrad_smallk(const float* __restrict__ U, const float* __restrict__ V, float* __restrict__ dW,
                                                      int K, int Nu, int Ncols, int tiles_c, float beta) {
  typedef float fvec __attribute__((ext_vector_type(NJ)));
  const int lane = threadIdx.x & 63;
  const int wt = blockIdx.x * 4 + (threadIdx.x >> 6);         // wave tile: column tiles fastest (the four waves share U rows)
  const int tr = wt / tiles_c, tc = wt - tr * tiles_c;
  const int n0 = tr * 64, c0 = tc * (32 * NJ);
  if (n0 >= Nu) return;
  const int lr = lane & 31, lh = lane >> 5;
  const int nA = n0 + 2 * lr;                                  // this lane's two rows (A operand)
  const bool okA = nA + 1 < Nu;                                // (Nu is even: a lane's pair is in or out as a whole)
  const float* upc = U + (okA ? nA : 0);                       // (a lane whose rows are past Nu reads rows 0, 1 and discards them)
  const float* vp = V + c0 + NJ * lr;
  f32x16 acc[2][NJ];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  // two batch rows per MFMA: lanes 0-31 take row k, lanes 32-63 row k + 1; rows past K read as zeros (an odd K, and the ring's
  // run-out: up to two surplus pairs of zero MFMAs).  The operand loads run TWO pairs ahead of the MFMAs through a ring of
  // three register sets: an L2 round trip is longer than the four MFMAs of a pair, and without the ring every pair waited it out.
  // (loads are unconditional — clamped row — and what they return is zeroed by a multiply only where the MFMAs consume it, two
  //  pairs later: no branch and no use sits between a load and the MFMAs issued in front of it)
  struct Pair { f32x2 a; fvec b; float ma, mb; };
  auto ld = [&](int k, Pair& p) {
    const int kk = k + lh;
    const bool live = kk < K;
    const int kc = live ? kk : K - 1;
    p.a = *(const f32x2*)(upc + (int64_t)kc * Nu);
    p.b = *(const fvec*)(vp + (int64_t)kc * Ncols);
    p.mb = live ? 1.f : 0.f;
    p.ma = (live && okA) ? 1.f : 0.f;
  };
  auto mm = [&](const Pair& p) {
    const f32x2 a = p.a * p.ma;
    fvec b;
#pragma unroll
    for (int j = 0; j < NJ; ++j) b[j] = p.b[j] * p.mb;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
  };
  Pair p0, p1, p2;
  ld(0, p0);
  ld(2, p1);
  for (int k = 0; k < K; k += 6) {          // (the scheduling barriers keep every load in FRONT of the MFMAs it is to overlap)
    ld(k + 4, p2);
    __builtin_amdgcn_sched_barrier(0);
    mm(p0);
    __builtin_amdgcn_sched_barrier(0);
    ld(k + 6, p0);
    __builtin_amdgcn_sched_barrier(0);
    mm(p1);
    __builtin_amdgcn_sched_barrier(0);
    ld(k + 8, p1);
    __builtin_amdgcn_sched_barrier(0);
    mm(p2);
    __builtin_amdgcn_sched_barrier(0);
  }
  // D layout: column block index lane & 31, row index (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
  auto store_all = [&](auto ACCUM) {
    constexpr bool accum = decltype(ACCUM)::value != 0;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = n0 + 2 * ((r & 3) + 8 * (r >> 2) + 4 * lh) + i;
        if (n >= Nu) continue;
        float* o = dW + (int64_t)n * Ncols + c0 + NJ * lr;
        fvec v;
#pragma unroll
        for (int j = 0; j < NJ; ++j) v[j] = acc[i][j][r];
        if constexpr (accum) v += beta * *(const fvec*)o;
        *(fvec*)o = v;
      }
  };
  if (beta != 0.f) store_all(VfIntC<1>{});
  else store_all(VfIntC<0>{});
}

}  // namespace

// dW[Nu][Ncols] = beta * dW + sum_k U[k][Nu] * V[k][Ncols].  Returns -1 when the shape is not this kernel's (the caller keeps
// its tiled path), 0 when launched, > 0 on a launch error.
int vf_internal_wgrad_smallk(vf_ctx* ctx, const float* U, const float* V, float* dW, int K, int Nu, int Ncols, float beta) {
  static const bool off = getenv("VF_NO_WGRAD_SMALLK") && atoi(getenv("VF_NO_WGRAD_SMALLK"));
  if (off || ctx->mfma_bf16 == 1) return -1;                  // (the bf16-operand mode rounds its operands: not this kernel's arithmetic)
  // measured (scripts/bench_bottleneck_wgrad.py): K = 4: 104 us against 192 (the write at memset speed, 6.0 TB/s); K = 16: 27.6
  // against 39.5; K = 64: 57 against 54 — there the fp32 MFMAs themselves (4.2 GFLOP at the f32 pipe's rate) take as long as
  // the tiled three-plane kernel, which besides shares its launch with the other layers of the group
  static const int kmax = getenv("VF_WGRAD_SMALLK_MAX") ? atoi(getenv("VF_WGRAD_SMALLK_MAX")) : 32;
  if (K < 1 || K > kmax || Ncols % 128 != 0 || Nu % 2 != 0 || Nu < 64) return -1;
  if ((((uintptr_t)U) & 7) || (((uintptr_t)V) & 15) || (((uintptr_t)dW) & 15)) return -1;
  static const int nj = getenv("VF_WGRAD_SMALLK_NJ") ? atoi(getenv("VF_WGRAD_SMALLK_NJ")) : 2;
  const int tiles_c = Ncols / (32 * nj), tiles_r = (int)vf_cdiv(Nu, 64);
  const int64_t wtiles = (int64_t)tiles_c * tiles_r;
  VfProf prof(ctx, "wgrad_smallk_f32", 2.0 * (double)K * Nu * Ncols, 4.0 * ((double)Nu * Ncols * (beta != 0.f ? 2 : 1) + (double)K * (Nu + Ncols)));
  if (nj == 4)
    hipLaunchKernelGGL(k_wgrad_smallk<4>, dim3((unsigned)vf_cdiv(wtiles, 4)), dim3(256), 0, ctx->stream, U, V, dW, K, Nu, Ncols, tiles_c, beta);
  else
    hipLaunchKernelGGL(k_wgrad_smallk<2>, dim3((unsigned)vf_cdiv(wtiles, 4)), dim3(256), 0, ctx->stream, U, V, dW, K, Nu, Ncols, tiles_c, beta);
  VF_LAUNCH_CHECK();
  return 0;
}
