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
// Round 5: from K = 64 (the batch of configs[1]; world x 64 under data parallelism) the FUSED form takes its products from the bf16 pipe
// like every other mode-3 kernel: k_fused_planes_prep splits both operands exactly into three bf16 planes in the fragment order of
// v_mfma_f32_32x32x16_bf16, and the K loop is 6 x (K / 16) MFMAs of 32 cycles per accumulator tile where the fp32 pipe needs K / 2 of 64
// (K = 64, 4000 x 8192: 188 -> 180 us with the pre-pass; K = 512, eight ranks' operands: 430 -> 355; the iteration -0.35 %).
#include <algorithm>
#include <cstdlib>

#include "vf_common.h"

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace {

// NJ = column blocks per wave (float2 / float4 of V per lane): 2 -> a 64 x 64 wave tile, 4 accumulator tiles (64 AGPRs: four or
// five waves per SIMD cover the L2 latency of the operand loads by themselves); 4 -> 64 x 128 (one wave per SIMD: slower, kept
// for the record)
// FUSE: optim.adam applied in the epilogue (vf_wgrad_adam_outer): the gradient never makes the round trip through memory —
// 24 B per weight (x, m, v read and written) instead of 4 (this kernel's store) + 28 (k_adam); A.g != NULL also stores it.
struct VfAdamFuse {
  float *x, *m, *v, *g;
  const int32_t* state;       // k_adam_prep's: [1] = bit pattern of the step size
  float b1, omb1, b2, omb2, eps;
  int round_bf16;             // operands rounded to bf16 (nearest even) on their way in: the arithmetic of matrix-core mode 1
  // operands gathered from several ranks (data parallel): batch row k lives in segment k / kps at row k % kps, the segments
  // seg floats apart (one rank's packed operands each); g = gscale * sum (1 / world: the mean over ranks).  One rank: kps = K.
  int kps, kshift;            // kshift: log2(kps) when it is a power of two, else -1
  int64_t seg;
  float gscale;
  // K >= 64 in the three-plane mode: the operands pre-split into bf16 planes in FRAGMENT order by k_fused_planes_prep (NULL: the fp32
  // matrix-core path below); kg = K / 16 k-groups
  const bf16x8* uf;
  const bf16x8* vf;
  int kg;
};
__device__ __forceinline__ float ws_rne(float f) {
  const unsigned u = __float_as_uint(f);
  return __uint_as_float(((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16) << 16);
}
// RB: rows of x, m, v in flight per epilogue batch; OCC: waves per SIMD the register budget is held to; PF: the first batch is
// loaded BEFORE the K loop, so it travels under the matrix-core phase
// wt: this wave's tile, column tiles fastest (the four waves of a block share U rows); ldu: floats per batch row of U (>= Nu)
template <int NJ, bool FUSE, int RB, bool PF>
__device__ __forceinline__ void wgrad_smallk_body(const float* __restrict__ U, const float* __restrict__ V, float* __restrict__ dW, int K, int Nu,
                                                  int ldu, int Ncols, int tiles_c, float beta, const VfAdamFuse& A, int wt) {
  typedef float fvec __attribute__((ext_vector_type(NJ)));
  const int lane = threadIdx.x & 63;
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
    if (FUSE && A.seg != 0) {       // (wave-uniform: gathered operands only)
      const int sg = A.kshift >= 0 ? kc >> A.kshift : kc / A.kps, kr = kc - sg * A.kps;
      p.a = *(const f32x2*)(upc + sg * A.seg + (int64_t)kr * ldu);
      p.b = *(const fvec*)(vp + sg * A.seg + (int64_t)kr * Ncols);
    } else {
      p.a = *(const f32x2*)(upc + (int64_t)kc * ldu);
      p.b = *(const fvec*)(vp + (int64_t)kc * Ncols);
    }
    p.mb = live ? 1.f : 0.f;
    p.ma = (live && okA) ? 1.f : 0.f;
  };
  auto mm = [&](const Pair& p) {
    f32x2 a = p.a * p.ma;
    fvec b;
#pragma unroll
    for (int j = 0; j < NJ; ++j) b[j] = p.b[j] * p.mb;
    if constexpr (FUSE) {
      if (A.round_bf16) {
        a[0] = ws_rne(a[0]);
        a[1] = ws_rne(a[1]);
#pragma unroll
        for (int j = 0; j < NJ; ++j) b[j] = ws_rne(b[j]);
      }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
  };
  // ---- FUSE: the epilogue's batches (rows r = rb .. rb + RB of accumulator row block i)
  struct Bat { fvec x[RB], m[RB], v[RB]; int off[RB]; };
  const int cofs = c0 + NJ * lr;
  auto bat_load = [&](int i, int rb, Bat& b) {
#pragma unroll
    for (int q = 0; q < RB; ++q) {
      const int r = rb + q;
      const int n = n0 + 2 * ((r & 3) + 8 * (r >> 2) + 4 * lh) + i;
      b.off[q] = (n < Nu ? n : n0) * Ncols + cofs;
      b.x[q] = *(const fvec*)(A.x + b.off[q]);
      b.m[q] = *(const fvec*)(A.m + b.off[q]);
      b.v[q] = *(const fvec*)(A.v + b.off[q]);
    }
  };
  Bat pre;
  if constexpr (FUSE && PF) bat_load(0, 0, pre);
  bool planes_done = false;
  if constexpr (FUSE && NJ == 2) {
    if (A.uf != nullptr) {      // (wave-uniform)
      // ---- the three-plane form: 16 batch rows per MFMA on the bf16 pipe, six product terms (smallest first: the order of every
      // mode-3 kernel of the library); the operands arrive pre-split in fragment order — one coalesced 1 KB load per (row block, plane,
      // k-group) — so the K = 64 gradient costs 96 MFMAs of 32 cycles where the fp32 pipe needs 128 of 64
      const bf16x8* ua = A.uf + ((int64_t)(tr * 2) * A.kg * 3) * 64 + lane;       // [row tile][i][k-group][plane][lane]
      const bf16x8* vb = A.vf + ((int64_t)(tc * 2) * A.kg * 3) * 64 + lane;       // [column tile][j][k-group][plane][lane]
      const int gstride = 3 * 64, istride = A.kg * 3 * 64;
      // (U fragments two sets deep, V fragments one: 144 registers beside the epilogue's batch — both two deep spilled)
      bf16x8 fa[2][2][3], fb[2][3];                                               // [set][i][plane], [j][plane]
      auto lda = [&](int g, int set) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int q = 0; q < 3; ++q) fa[set][i][q] = ua[(int64_t)i * istride + g * gstride + q * 64];
      };
      auto ldb = [&](int g) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int q = 0; q < 3; ++q) fb[j][q] = vb[(int64_t)j * istride + g * gstride + q * 64];
      };
      lda(0, 0);
      for (int g = 0; g < A.kg; g += 2) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          if (g + h >= A.kg) break;
          ldb(g + h);
          if (g + h + 1 < A.kg) lda(g + h + 1, h ^ 1);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[h][i][1], fb[j][1], acc[i][j], 0, 0, 0);
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[h][i][0], fb[j][2], acc[i][j], 0, 0, 0);
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[h][i][2], fb[j][0], acc[i][j], 0, 0, 0);
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[h][i][0], fb[j][1], acc[i][j], 0, 0, 0);
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[h][i][1], fb[j][0], acc[i][j], 0, 0, 0);
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[h][i][0], fb[j][0], acc[i][j], 0, 0, 0);
            }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      planes_done = true;
    }
  }
  if (!planes_done) {
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
  if constexpr (FUSE) {
    // RB rows at a time: their x, m, v (3 RB loads of NJ floats, 256-byte runs per half wave) are in flight together, OCC waves
    // per SIMD cover for one another
    const float step = __int_as_float(A.state[1]);
    auto bat_apply = [&](auto IC, auto RBC, Bat& b) {
      constexpr int i = decltype(IC)::value, rb = decltype(RBC)::value;
#pragma unroll
      for (int q = 0; q < RB; ++q) {
        const int r = rb + q;
        if (n0 + 2 * ((r & 3) + 8 * (r >> 2) + 4 * lh) + i >= Nu) continue;
        fvec gv;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          gv[j] = acc[i][j][r] * A.gscale;
          float xe = b.x[q][j], me = b.m[q][j], ve = b.v[q][j];
          vf_adam_upd(xe, gv[j], me, ve, A.b1, A.omb1, A.b2, A.omb2, A.eps, step);
          b.x[q][j] = xe;
          b.m[q][j] = me;
          b.v[q][j] = ve;
        }
        *(fvec*)(A.x + b.off[q]) = b.x[q];
        *(fvec*)(A.m + b.off[q]) = b.m[q];
        *(fvec*)(A.v + b.off[q]) = b.v[q];
        if (A.g) *(fvec*)(A.g + b.off[q]) = gv;
      }
    };
    vf_static_for<2>([&](auto IC) {
      vf_static_for<16 / RB>([&](auto BC) {
        constexpr int i = decltype(IC)::value, rb = decltype(BC)::value * RB;
        if constexpr (PF && i == 0 && rb == 0) {
          bat_apply(IC, VfIntC<rb>{}, pre);
        } else {
          Bat b;
          bat_load(i, rb, b);
          bat_apply(IC, VfIntC<rb>{}, b);
        }
        __builtin_amdgcn_sched_barrier(0);      // (the next rows' loads stay behind these stores)
      });
    });
  } else {
    if (beta != 0.f) store_all(VfIntC<1>{});
    else store_all(VfIntC<0>{});
  }
}

template <int NJ, bool FUSE = false, int RB = 4, int OCC = 3, bool PF = false>
__global__ __launch_bounds__(256, FUSE ? OCC : 1) void k_wgrad_smallk(const float* __restrict__ U, const float* __restrict__ V, float* __restrict__ dW,
                                                      int K, int Nu, int Ncols, int tiles_c, float beta, const VfAdamFuse A) {
  wgrad_smallk_body<NJ, FUSE, RB, PF>(U, V, dW, K, Nu, Nu, Ncols, tiles_c, beta, A, blockIdx.x * 4 + (threadIdx.x >> 6));
}

// the fused update of SEVERAL layers (the bottleneck pair: two tensors) in one launch: a block's four wave tiles belong to one
// layer (each layer's tile count is padded to a multiple of four); a tail of one layer overlaps the start of the next instead of
// draining the chip between two launches
struct VfFusedTable {
  int nl;
  int blk_off[VF_FUSED_MAX + 1];      // first block of layer i
  int tiles_c[VF_FUSED_MAX];
  int K[VF_FUSED_MAX], Nu[VF_FUSED_MAX], ldu[VF_FUSED_MAX], Ncols[VF_FUSED_MAX];
  const float* U[VF_FUSED_MAX];
  const float* V[VF_FUSED_MAX];
  VfAdamFuse A[VF_FUSED_MAX];
};
// Operands of the fused update as bf16 planes in FRAGMENT order (the three-plane form above): entry (tile t, half h, k-group g, plane q,
// lane l) = the eight k values 16 g + 8 (l / 32) .. + 7 of operand row / column 64 t + 2 (l % 32) + h, exact three-way split (hi + mid + lo
// == x bit for bit: vf_pgemm.hip pg_split4's arithmetic), zeros past the matrix edge.  One thread per (operand, t, h, g, l): eight strided
// fp32 loads (32 lanes cover one 256-byte run between them and their h-partner), three 16-byte stores.  1.5 + 3.1 MB per layer at K = 64.
struct VfPrepTable {
  int nl;
  int blk_off[VF_FUSED_MAX + 1];
  int ut[VF_FUSED_MAX], vt[VF_FUSED_MAX];      // U row tiles, V column tiles (64 wide; the V entries follow the U entries)
  bf16x8* uf[VF_FUSED_MAX];
  bf16x8* vf[VF_FUSED_MAX];
};
__global__ __launch_bounds__(256) void k_fused_planes_prep(const VfFusedTable T, const VfPrepTable P) {
  int l = 0;
  while (l + 1 < P.nl && (int)blockIdx.x >= P.blk_off[l + 1]) ++l;
  const VfAdamFuse& A = T.A[l];
  const int kg = A.kg;
  const int64_t e = ((int64_t)((int)blockIdx.x - P.blk_off[l]) * 256 + threadIdx.x);      // entry (t, h, g, lane) with lane fastest
  const int lane = (int)(e & 63);
  const int64_t r = e >> 6;
  const int g = (int)(r % kg);
  const int64_t th = r / kg;                    // 2 t + h
  const bool is_v = th >= 2 * (int64_t)P.ut[l];
  const int64_t th2 = is_v ? th - 2 * (int64_t)P.ut[l] : th;
  const int t = (int)(th2 >> 1), h = (int)(th2 & 1);
  const int ncols = is_v ? T.Ncols[l] : T.Nu[l];
  if (th >= 2 * (int64_t)(P.ut[l] + P.vt[l])) return;      // (padding threads of the layer's last block)
  const int col = t * 64 + 2 * (lane & 31) + h;
  const float* base = is_v ? T.V[l] : T.U[l];
  const int ld = is_v ? T.Ncols[l] : T.ldu[l];
  float x[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int k = 16 * g + 8 * (lane >> 5) + q;
    float v = 0.f;
    if (k < T.K[l] && col < ncols) {
      if (A.seg != 0) {
        const int sg = A.kshift >= 0 ? k >> A.kshift : k / A.kps, kr = k - sg * A.kps;
        v = base[sg * A.seg + (int64_t)kr * ld + col];
      } else {
        v = base[(int64_t)k * ld + col];
      }
    }
    x[q] = v;
  }
  bf16x8* dst = (is_v ? P.vf[l] : P.uf[l]) + (((int64_t)(t * 2 + h) * kg + g) * 3) * 64 + lane;
#pragma unroll
  for (int p = 0; p < 3; ++p) {
    typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
    u16x8 o;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const unsigned u = __float_as_uint(x[q]);
      o[q] = (unsigned short)(u >> 16);
      x[q] -= __uint_as_float(u & 0xffff0000u);
    }
    *(u16x8*)(dst + p * 64) = o;
  }
}

template <int RB, int OCC>
__global__ __launch_bounds__(256, OCC) void k_adam_fused_multi(const VfFusedTable T) {
  int l = 0;
  while (l + 1 < T.nl && (int)blockIdx.x >= T.blk_off[l + 1]) ++l;
  const int wt = ((int)blockIdx.x - T.blk_off[l]) * 4 + (threadIdx.x >> 6);
  wgrad_smallk_body<2, true, RB, false>(T.U[l], T.V[l], (float*)nullptr, T.K[l], T.Nu[l], T.ldu[l], T.Ncols[l], T.tiles_c[l], 0.f, T.A[l], wt);
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
    hipLaunchKernelGGL((k_wgrad_smallk<4, false>), dim3((unsigned)vf_cdiv(wtiles, 4)), dim3(256), 0, ctx->stream, U, V, dW, K, Nu, Ncols, tiles_c, beta, VfAdamFuse{});
  else
    hipLaunchKernelGGL((k_wgrad_smallk<2, false>), dim3((unsigned)vf_cdiv(wtiles, 4)), dim3(256), 0, ctx->stream, U, V, dW, K, Nu, Ncols, tiles_c, beta, VfAdamFuse{});
  VF_LAUNCH_CHECK();
  return 0;
}

// Measured and rejected (round 3): a persistent producer / consumer form — blocks of three wave PAIRS, the producer of a pair walks the
// K loop and hands the 64 x 64 accumulator tile through 16 KB of LDS to its consumer, which streams x, m, v; bounded flag waits in
// LDS; one launch for all layers.  Correct (the bit-for-bit tests passed on it) and slower everywhere: K = 64 301 us against 180,
// K = 4 (configs[4], no matrix-core phase to hide) 1287 against 735, K = 512 965 against 436 — half the waves of a CU issue
// loads, so half the bytes are in flight, and the other half's MFMAs have one or two waves per SIMD to hide their operand latency
// with instead of three.  Start-up phase offsets between the waves of the plain form (so that one wave's MFMAs fall into the
// others' memory phase) only added their own delay: 178 -> 188 / 201 / 256 us for offsets of 1 / 3 / 8 x 4096 cycles.
// The bottleneck pair's weight gradient with optim.adam in its epilogue:  g = sum_k U[k][:]^T V[k][:]  (K = batch) is formed in the
// matrix-core accumulators and consumed there — x, m, v [Nu][Ncols] are read and written once (24 B per weight; + 4 when g, which
// may be NULL, is stored too), where the two-kernel form writes g, then reads it again beside x, m, v (32 B).  t_dev is the
// optimiser's device state AFTER vf_adam_prep (the step size of this update), as for vf_adam_apply.  Returns 2 (an error) for a
// shape the kernel does not take: ask vf_wgrad_adam_outer_supported first.
VF_API int vf_wgrad_adam_outer_supported(int K, int Nu, int Ncols) {
  return K >= 1 && Ncols % 128 == 0 && Nu % 2 == 0 && Nu >= 64 && (int64_t)Nu * Ncols < ((int64_t)1 << 31);
}
int vf_internal_adam_fused_multi(vf_ctx* ctx, const VfFusedLayer* layers, int nl, double beta1, double beta2, double eps,
                                 const int32_t* t_dev) {
  VF_REQUIRE(ctx && layers && t_dev && nl >= 1 && nl <= VF_FUSED_MAX, "fused update: %d layers (at most %d per launch)", nl, VF_FUSED_MAX);
  VfFusedTable T;
  memset(&T, 0, sizeof(T));
  T.nl = nl;
  double flops = 0, bytes = 0;
  for (int i = 0; i < nl; ++i) {
    const VfFusedLayer& L = layers[i];
    VF_REQUIRE(L.U && L.V && L.x && L.m && L.v, "vf_wgrad_adam_outer: NULL argument");
    VF_REQUIRE(vf_wgrad_adam_outer_supported(L.K, L.Nu, L.Ncols), "vf_wgrad_adam_outer: K %d, %d x %d is not this kernel's shape", L.K, L.Nu,
               L.Ncols);
    VF_REQUIRE(L.kps >= 1 && L.K % L.kps == 0 && (L.K == L.kps || L.seg % 4 == 0),
               "vf_wgrad_adam_outer: %d batch rows in segments of %d (stride %lld floats)", L.K, L.kps, (long long)L.seg);
    const int ldu = L.ldu ? L.ldu : L.Nu, row0 = L.ldu ? L.row0 : 0;
    VF_REQUIRE(row0 >= 0 && row0 % 2 == 0 && row0 + L.Nu <= ldu, "vf_wgrad_adam_outer: rows [%d, %d) of %d", row0, row0 + L.Nu, ldu);
    VF_REQUIRE(!((((uintptr_t)L.U) | ((uintptr_t)L.V)) & 7) &&
                   !((((uintptr_t)L.x) | ((uintptr_t)L.m) | ((uintptr_t)L.v) | ((uintptr_t)L.g_out)) & 15),
               "vf_wgrad_adam_outer: U, V must be 8-byte aligned, x, m, v, g 16-byte aligned");
    const int64_t roff = (int64_t)row0 * L.Ncols;
    VfAdamFuse& F = T.A[i];
    F.x = L.x + roff; F.m = L.m + roff; F.v = L.v + roff; F.g = L.g_out ? L.g_out + roff : nullptr;
    F.state = t_dev;
    F.b1 = (float)beta1; F.omb1 = (float)(1.0 - beta1);
    F.b2 = (float)beta2; F.omb2 = (float)(1.0 - beta2);
    F.eps = (float)eps;
    F.round_bf16 = ctx->mfma_bf16 == 1;
    F.kps = L.kps; F.kshift = vf_is_pow2(L.kps) ? vf_ilog2(L.kps) : -1;
    F.seg = L.K == L.kps ? 0 : L.seg;       // (one segment: plain row addressing)
    F.gscale = L.gscale;
    T.U[i] = L.U + row0; T.V[i] = L.V;
    T.K[i] = L.K; T.Nu[i] = L.Nu; T.ldu[i] = ldu; T.Ncols[i] = L.Ncols;
    T.tiles_c[i] = L.Ncols / 64;
    const int64_t tiles = (int64_t)T.tiles_c[i] * vf_cdiv(L.Nu, 64);
    T.blk_off[i + 1] = T.blk_off[i] + (int)vf_cdiv(tiles, 4);
    const double n = (double)L.Nu * L.Ncols;
    // (2 K flops per 24 bytes is far below the matrix pipe's ridge: bench.py prices it by its bytes)
    flops += 2.0 * L.K * n;
    bytes += (L.g_out ? 28.0 : 24.0) * n + 4.0 * L.K * ((double)L.Nu + L.Ncols);
  }
  // ---- three-plane mode, K >= 64: the gradient on the bf16 pipe from operands pre-split in fragment order (k_fused_planes_prep into the
  // context's workspace; the fp32 pipe's 128 MFMAs of 64 cycles per 64 x 64 tile become 96 of 32).  VF_ADAM_PLANES=0 keeps the fp32 form.
  {
    static const int env_pl = getenv("VF_ADAM_PLANES") ? atoi(getenv("VF_ADAM_PLANES")) : 1;
    bool ok = env_pl && ctx->mfma_bf16 == 3;
    size_t need = 0;
    for (int i = 0; i < nl && ok; ++i) {
      ok = T.K[i] >= 64 && T.K[i] % 16 == 0;
      need += (size_t)(vf_cdiv(T.Nu[i], 64) + T.Ncols[i] / 64) * 2 * (T.K[i] / 16) * 3 * 64 * 16;
    }
    if (ok && need <= vf_ws_avail(ctx)) {
      VfPrepTable P;
      memset(&P, 0, sizeof(P));
      P.nl = nl;
      bf16x8* w = (bf16x8*)vf_ws_ptr(ctx);
      for (int i = 0; i < nl; ++i) {
        const int kg = T.K[i] / 16;
        P.ut[i] = (int)vf_cdiv(T.Nu[i], 64);
        P.vt[i] = T.Ncols[i] / 64;
        P.uf[i] = w;
        w += (size_t)P.ut[i] * 2 * kg * 3 * 64;
        P.vf[i] = w;
        w += (size_t)P.vt[i] * 2 * kg * 3 * 64;
        T.A[i].uf = P.uf[i];
        T.A[i].vf = P.vf[i];
        T.A[i].kg = kg;
        P.blk_off[i + 1] = P.blk_off[i] + (int)vf_cdiv((int64_t)(P.ut[i] + P.vt[i]) * 2 * kg * 64, 256);
      }
      VfProf prof(ctx, "adam_fused_operand_planes", 0.0, (double)need * (1.0 + 4.0 / 6.0));
      hipLaunchKernelGGL(k_fused_planes_prep, dim3((unsigned)P.blk_off[nl]), dim3(256), 0, ctx->stream, T, P);
      VF_LAUNCH_CHECK();
    }
  }
  // measured (scripts/bench_fused_adam.py; K = 64, 4000 x 8192): RB 4 / three waves per SIMD 180 us; RB 8 / two waves 181; RB 16 / two
  // waves 233 (spills); first batch loaded ahead of the K loop 174-180.  Round 4: all layers in ONE launch (a table in the
  // kernel arguments) — the tail of one tensor's update overlaps the start of the next.
  // (the first epilogue batch loaded ahead of the now shorter K loop: 168 registers and 164 bytes of scratch — not instantiated)
  VF_LAUNCH_TIMED(ctx, "adam_fused_wgrad", flops, bytes, (k_adam_fused_multi<4, 3>), dim3((unsigned)T.blk_off[nl]), dim3(256), T);
  VF_LAUNCH_CHECK();
  return 0;
}
VF_API int vf_wgrad_adam_outer_gathered(vf_ctx* ctx, const float* U, const float* V, int K, int rows_per_seg, int64_t seg_stride, int Nu,
                                        int Ncols, float* x, float* m, float* v, float* g, float gscale, double beta1, double beta2,
                                        double eps, const int32_t* t_dev) {
  VF_REQUIRE(ctx && U && V && x && m && v && t_dev, "vf_wgrad_adam_outer: NULL argument");
  VfFusedLayer L;
  L.U = U; L.V = V; L.x = x; L.m = m; L.v = v; L.g_out = g;
  L.K = K; L.Nu = Nu; L.Ncols = Ncols;
  L.kps = rows_per_seg; L.seg = seg_stride; L.gscale = gscale;
  L.row0 = 0; L.ldu = 0;
  return vf_internal_adam_fused_multi(ctx, &L, 1, beta1, beta2, eps, t_dev);
}
// rows [row0, row0 + nrows) of the [Nu][Ncols] tensors only (x, m, v, g point at row 0; U's batch rows stay Nu floats long): the
// data-parallel update sharded by weight rows — every row block gets, bit for bit, what the whole-tensor call gives it
VF_API int vf_wgrad_adam_outer_rows(vf_ctx* ctx, const float* U, const float* V, int K, int rows_per_seg, int64_t seg_stride, int Nu,
                                    int Ncols, int row0, int nrows, float* x, float* m, float* v, float* g, float gscale, double beta1,
                                    double beta2, double eps, const int32_t* t_dev) {
  VF_REQUIRE(ctx && U && V && x && m && v && t_dev, "vf_wgrad_adam_outer_rows: NULL argument");
  VfFusedLayer L;
  L.U = U; L.V = V; L.x = x; L.m = m; L.v = v; L.g_out = g;
  L.K = K; L.Nu = nrows; L.Ncols = Ncols;
  L.kps = rows_per_seg; L.seg = seg_stride; L.gscale = gscale;
  L.row0 = row0; L.ldu = Nu;
  return vf_internal_adam_fused_multi(ctx, &L, 1, beta1, beta2, eps, t_dev);
}
VF_API int vf_wgrad_adam_outer(vf_ctx* ctx, const float* U, const float* V, int K, int Nu, int Ncols, float* x, float* m, float* v,
                               float* g, double beta1, double beta2, double eps, const int32_t* t_dev) {
  return vf_wgrad_adam_outer_gathered(ctx, U, V, K, K, 0, Nu, Ncols, x, m, v, g, 1.f, beta1, beta2, eps, t_dev);
}
