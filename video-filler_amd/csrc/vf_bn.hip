// vf_bn.hip — nn.SpatialBatchNormalization (train forward, backward, eval forward) on NHWC rows [npix][C].
//
// Reference: THNN BatchNormalization.c (double accumulators) / THCUNN BatchNormalization.cu, reached from
// train.lua:92-101,125,135-144,189-193.  HBM-bound: every kernel walks rows with the channel axis on the
// lanes (16-byte loads, fully coalesced), per-thread fp32 partial sums over <= 64 rows, then DOUBLE partials
// combined in a fixed order (deterministic).  Statistics are shifted by the running mean so that
// var = E[(x-s)^2] - E[x-s]^2 does not cancel catastrophically; the two-phase split (stats | finalize+apply)
// is the hook where a data-parallel caller all-reduces the per-channel sums (SyncBN).
// (Measured and rejected: a single-launch form in which a block owns one float4 channel column over all rows — 22 us
// per 2 MB tensor against ~17 us for the three launches; too few blocks to pull L2 bandwidth.  A second attempt that keeps
// the block's slice in registers — one read, one launch, 16 channels per block — took 16 us for the same tensor
// against 9.4 us for the three launches after their partial-row count was scaled down: 32 blocks of strided 64-byte
// row segments are latency-bound.)
#include <algorithm>

#include "vf_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

// Exact three-way split of four fp32 values into bf16 planes (hi + mid + lo == x bit for bit; the arithmetic of
// vf_pgemm.hip's pg_split4): the BatchNorm apply / backward passes write the planes of their output beside it, so the
// convolution that consumes it (vf_pconv_*) finds its operand already split.  plane q of element i: planes[q * pstride + i].
// npl = 1 (the bf16-operand mode): ONE plane, rounded to nearest-even — the rounding vf_conv.hip's BF = 1 kernels apply inside
// the GEMM, done once by the producer (vf_pgemm.hip pg_rne16).
__device__ __forceinline__ unsigned bn_rne16(float v) {
  const unsigned u = __float_as_uint(v);
  return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}
__device__ __forceinline__ void bn_store_planes(unsigned short* __restrict__ planes, int64_t pstride, int64_t i, f32x4 v, int npl) {
  if (npl == 1) {
    u32x2 o;
    o[0] = bn_rne16(v[0]) | (bn_rne16(v[1]) << 16);
    o[1] = bn_rne16(v[2]) | (bn_rne16(v[3]) << 16);
    *(u32x2*)(planes + i) = o;
    return;
  }
  float r0 = v[0], r1 = v[1], r2 = v[2], r3 = v[3];
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    const unsigned u0 = __float_as_uint(r0), u1 = __float_as_uint(r1), u2 = __float_as_uint(r2), u3 = __float_as_uint(r3);
    u32x2 o;
    o[0] = __builtin_amdgcn_perm(u1, u0, 0x07060302u);
    o[1] = __builtin_amdgcn_perm(u3, u2, 0x07060302u);
    *(u32x2*)(planes + q * pstride + i) = o;
    if (q < 2) {
      r0 -= __uint_as_float(u0 & 0xffff0000u);
      r1 -= __uint_as_float(u1 & 0xffff0000u);
      r2 -= __uint_as_float(u2 & 0xffff0000u);
      r3 -= __uint_as_float(u3 & 0xffff0000u);
    }
  }
}

struct BnGeom {
  int cq;              // float4 columns per block (power of two <= 64)
  int rp;              // row lanes per block = 256 / cq
  int gy;              // column chunks
  int gx;              // row slabs
  int rows_per_block;
};

static BnGeom bn_geom(int64_t npix, int C, int target_blocks = 2048) {
  BnGeom g;
  const int C4 = C / 4;
  int cq = 1;
  while (cq < C4 && cq < 64) cq <<= 1;
  g.cq = cq;
  g.rp = 256 / cq;
  g.gy = (int)vf_cdiv(C4, cq);
  // HBM-bound: aim at ~2048 blocks (8 per CU) so enough 16-byte loads are in flight; a block walks whole
  // multiples of its rp row lanes
  const int64_t gx_target = std::max<int64_t>(1, target_blocks / g.gy);
  int64_t rpb = std::max<int64_t>(g.rp, vf_cdiv(npix, gx_target));
  rpb = vf_cdiv(rpb, g.rp) * g.rp;
  g.rows_per_block = (int)rpb;
  g.gx = (int)vf_cdiv(npix, rpb);
  return g;
}

// Reduction kernels write one partial row per block and a second stage walks those rows column by column: fewer,
// fatter blocks keep that second stage short (2048 rows made it the longest of BatchNorm's three launches).
static int bn_stat_blocks(int64_t npix, int C, int bytes_per_block) {
  static const int forced = getenv("VF_BN_STAT_BLOCKS") ? atoi(getenv("VF_BN_STAT_BLOCKS")) : 0;
  if (forced > 0) return forced;
  // measured (scripts/bench_bn.py sweep, gpurun_out/bn_sweep.log): one block per 32 KB of a tensor read once (forward
  // statistics, bias gradients), per 16 KB of a tensor read beside two others (backward), between 128 and 512 blocks
  const int64_t b = npix * C * 4 / bytes_per_block;
  return (int)std::min<int64_t>(512, std::max<int64_t>(128, b));
}

// ---- forward statistics: partial[slab][2][C] (double)
__global__ __launch_bounds__(256) void k_bn_stats(const float* __restrict__ x, const float* __restrict__ shift,
                                                  double* __restrict__ part, int64_t npix, int C, int cq, int rows_per_block) {
  // blockIdx.z = batch group (netD's real and fake halves in one tensor): rows [z*npix, (z+1)*npix), its own partials
  x += (int64_t)blockIdx.z * npix * C;
  part += (int64_t)blockIdx.z * gridDim.x * 2 * C;
  const int rp = 256 / cq;
  const int tx = threadIdx.x % cq, ty = threadIdx.x / cq;
  const int c4 = blockIdx.y * cq + tx;
  const int C4 = C >> 2;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block, r1 = min(npix, r0 + rows_per_block);
  f32x4 s = {0, 0, 0, 0}, ss = {0, 0, 0, 0};
  if (c4 < C4) {
    const f32x4 sh = shift ? *(const f32x4*)(shift + 4 * c4) : f32x4{0, 0, 0, 0};
    for (int64_t r = r0 + ty; r < r1; r += rp) {
      const f32x4 v = *(const f32x4*)(x + r * C + 4 * c4) - sh;
      s += v;
      ss += v * v;
    }
  }
  __shared__ f32x4 red[2][256];
  red[0][threadIdx.x] = s;
  red[1][threadIdx.x] = ss;
  __syncthreads();
  if (ty == 0 && c4 < C4) {
    double a[4] = {0, 0, 0, 0}, b[4] = {0, 0, 0, 0};
    for (int j = 0; j < rp; ++j) {
      const f32x4 u = red[0][j * cq + tx], w = red[1][j * cq + tx];
      for (int e = 0; e < 4; ++e) {
        a[e] += (double)u[e];
        b[e] += (double)w[e];
      }
    }
    double* o = part + (int64_t)blockIdx.x * 2 * C;
    for (int e = 0; e < 4; ++e) {
      o[4 * c4 + e] = a[e];
      o[C + 4 * c4 + e] = b[e];
    }
  }
}

// Second stage of every two-stage reduction: partial[slab][ncol] (double) -> per-column totals.
// A block owns 16 columns x 16 slab lanes: lane (cx, ry) walks rows ry, ry + 16, ... of its column with four independent
// sums in flight (128-byte row segments, coalesced over cx), the 16 slab lanes of a column meet in LDS in a fixed order:
// deterministic.  (The first form — one wave per column, its 64 lanes a partial row apart — read one cache line per lane
// and took 5-9 us for the 512-1024 rows the GEMM epilogues leave; this one is bounded by the launch.)
//   EPI 0: sums[col] = total                                   (two-phase BN API, backward statistics)
//   EPI 1: BatchNorm finalize fused: the block owns 8 channels = columns {c, C+c}; writes save_mean/save_invstd and
//          updates the running statistics (THNN BatchNormalization_updateOutput, train branch); also stores sums
//   EPI 2: bias gradient: gb[col] = beta*gb[col] + total
template <int EPI>
__global__ __launch_bounds__(256) void k_reduce_partials(const double* __restrict__ part, int nslab, int ncol,
                                                         double* __restrict__ sums, float* __restrict__ running_mean,
                                                         float* __restrict__ running_var, float* __restrict__ save_mean,
                                                         float* __restrict__ save_invstd, double n, float momentum, float eps,
                                                         float* __restrict__ gb, float beta, int groups) {
  const int cx = threadIdx.x & 15, ry = threadIdx.x >> 4;
  if constexpr (EPI == 0) {       // blockIdx.y = batch group: its own partials and totals
    part += (int64_t)blockIdx.y * nslab * ncol;
    sums += (int64_t)blockIdx.y * ncol;
  }
  int col;
  bool ok;
  if constexpr (EPI == 1) {
    const int C = ncol >> 1;
    const int c = blockIdx.x * 8 + (cx & 7);
    ok = c < C;
    col = (cx >> 3) * C + c;
  } else {
    col = blockIdx.x * 16 + cx;
    ok = col < ncol;
  }
  __shared__ double red[16][17];
  auto column_total = [&](const double* pbase) {      // valid in the threads with ry == 0
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    if (ok) {
      // 512 partial rows per pass: the lane's 32 loads are ALL in flight before the first sum (the GEMM epilogues leave 512-1024 rows in
      // another XCD's L2 or in memory: with four loads per loop iteration the walk was eight dependent round trips, 5-7 us for a launch
      // that moves 64 KB).  Same four running sums, fed in the same order as the streaming loop it replaces: bit-identical totals.
      const double* p = pbase + col;
      for (int base = 0; base < nslab; base += 512) {
        double v[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) {
          const int k = base + ry + 16 * i;
          v[i] = k < nslab ? p[(int64_t)k * ncol] : 0.0;
        }
#pragma unroll
        for (int g = 0; g < 8; ++g) {
          const int k0 = base + ry + 64 * g;
          if (k0 + 48 < nslab) {
            s0 += v[4 * g];
            s1 += v[4 * g + 1];
            s2 += v[4 * g + 2];
            s3 += v[4 * g + 3];
          } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
              if (k0 + 16 * j < nslab) s0 += v[4 * g + j];
          }
        }
      }
    }
    __syncthreads();                                   // (the previous group's totals have been read)
    red[ry][cx] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    double t = 0;
    if (ry == 0) {
#pragma unroll
      for (int j = 0; j < 16; ++j) t += red[j][cx];
    }
    return t;
  };
  if constexpr (EPI == 0) {
    const double t = column_total(part);
    if (ry == 0 && ok) sums[col] = t;
  } else if constexpr (EPI == 2) {
    const double t = column_total(part);
    if (ry == 0 && ok) gb[col] = (beta != 0.f ? beta * gb[col] : 0.f) + (float)t;
  } else {
    // groups > 1: the batch groups are finalized one after the other, as separate forward calls would — every group's
    // sums were taken about the SAME shift (the running mean before the first update), the running averages move once
    // per group in group order
    __shared__ double tot[16];
    const int C = ncol >> 1;
    const int c = blockIdx.x * 8 + (cx & 7);
    double shift = 0;
    if (ok) shift = running_mean[c];
    for (int g = 0; g < groups; ++g) {
      const double t = column_total(part + (int64_t)g * nslab * ncol);
      if (ry == 0) tot[cx] = t;
      __syncthreads();
      if (ry == 0 && ok && cx < 8) {
        const double q1 = tot[cx], q2 = tot[cx + 8];
        if (sums) {
          sums[(int64_t)g * ncol + c] = q1;
          sums[(int64_t)g * ncol + C + c] = q2;
        }
        const double mean = shift + q1 / n;
        double m2 = q2 - q1 * q1 / n;  // = sum (x - mean)^2
        if (m2 < 0) m2 = 0;
        const float invstd = (m2 == 0 && eps == 0.f) ? 0.f : (float)(1.0 / sqrt(m2 / n + (double)eps));
        save_mean[(int64_t)g * C + c] = (float)mean;
        save_invstd[(int64_t)g * C + c] = invstd;
        running_mean[c] = (float)(momentum * mean + (1.0 - momentum) * running_mean[c]);
        const double unbiased = m2 / (n - 1.0);  // n == 1 -> inf/NaN, as the reference
        running_var[c] = (float)(momentum * unbiased + (1.0 - momentum) * running_var[c]);
      }
    }
  }
}

// column sums of a row-major [P][C] fp32 matrix (conv bias gradients): gb = beta*gb + sum_p g[p][:]
__global__ __launch_bounds__(256) void k_colsum4(const float* __restrict__ g, double* __restrict__ part, int64_t P, int C,
                                                 int cq, int rows_per_block) {
  const int rp = 256 / cq;
  const int tx = threadIdx.x % cq, ty = threadIdx.x / cq;
  const int c4 = blockIdx.y * cq + tx;
  const int C4 = C >> 2;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block, r1 = min(P, r0 + rows_per_block);
  f32x4 s = {0, 0, 0, 0};
  if (c4 < C4)
    for (int64_t r = r0 + ty; r < r1; r += rp) s += *(const f32x4*)(g + r * C + 4 * c4);
  __shared__ f32x4 red[256];
  red[threadIdx.x] = s;
  __syncthreads();
  if (ty == 0 && c4 < C4) {
    double a[4] = {0, 0, 0, 0};
    for (int j = 0; j < rp; ++j) {
      const f32x4 u = red[j * cq + tx];
      for (int e = 0; e < 4; ++e) a[e] += (double)u[e];
    }
    double* o = part + (int64_t)blockIdx.x * C;
    for (int e = 0; e < 4; ++e) o[4 * c4 + e] = a[e];
  }
}
// scalar variant for C % 4 != 0 (C = 3 image channels, C = 1): threads along rows, one column per blockIdx.y
__global__ __launch_bounds__(256) void k_colsum1(const float* __restrict__ g, double* __restrict__ part, int64_t P, int C,
                                                 int rows_per_block) {
  const int c = blockIdx.y;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block, r1 = min(P, r0 + rows_per_block);
  float s = 0.f;
  for (int64_t r = r0 + threadIdx.x; r < r1; r += 256) s += g[r * C + c];
  __shared__ double red[256];
  red[threadIdx.x] = (double)s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) part[(int64_t)blockIdx.x * C + c] = red[0];
}
// mean/invstd + running statistics (THNN BatchNormalization_updateOutput, train branch)
__global__ void k_bn_finalize(const double* __restrict__ sums, float* __restrict__ running_mean,
                              float* __restrict__ running_var, float* __restrict__ save_mean,
                              float* __restrict__ save_invstd, double n, int C, float momentum, float eps) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const double shift = running_mean[c];
  const double s1 = sums[c], s2 = sums[C + c];
  const double mean = shift + s1 / n;
  double m2 = s2 - s1 * s1 / n;  // = sum (x - mean)^2
  if (m2 < 0) m2 = 0;
  float invstd;
  if (m2 == 0 && eps == 0.f)
    invstd = 0.f;
  else
    invstd = (float)(1.0 / sqrt(m2 / n + (double)eps));
  save_mean[c] = (float)mean;
  save_invstd[c] = invstd;
  running_mean[c] = (float)(momentum * mean + (1.0 - momentum) * running_mean[c]);
  const double unbiased = m2 / (n - 1.0);  // n == 1 -> inf/NaN, as the reference
  running_var[c] = (float)(momentum * unbiased + (1.0 - momentum) * running_var[c]);
}

// y = act(((x - mean) * invstd) * gamma + beta)
__global__ __launch_bounds__(256) void k_bn_apply(const float* __restrict__ x, float* __restrict__ y,
                                                  const float* __restrict__ gamma, const float* __restrict__ beta,
                                                  const float* __restrict__ mean, const float* __restrict__ invstd,
                                                  int64_t npix, int C, int cq, int rows_per_block, int act, float slope,
                                                  unsigned short* __restrict__ planes, int64_t pstride, int npl) {
  x += (int64_t)blockIdx.z * npix * C;        // batch group z: its rows, its statistics
  y += (int64_t)blockIdx.z * npix * C;
  if (planes) planes += (int64_t)blockIdx.z * npix * C;
  mean += (int64_t)blockIdx.z * C;
  invstd += (int64_t)blockIdx.z * C;
  const int rp = 256 / cq;
  const int tx = threadIdx.x % cq, ty = threadIdx.x / cq;
  const int c4 = blockIdx.y * cq + tx;
  if (c4 >= (C >> 2)) return;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block, r1 = min(npix, r0 + rows_per_block);
  const f32x4 mu = *(const f32x4*)(mean + 4 * c4), is = *(const f32x4*)(invstd + 4 * c4);
  const f32x4 ga = gamma ? *(const f32x4*)(gamma + 4 * c4) : f32x4{1, 1, 1, 1};
  const f32x4 be = beta ? *(const f32x4*)(beta + 4 * c4) : f32x4{0, 0, 0, 0};
  for (int64_t r = r0 + ty; r < r1; r += rp) {
    f32x4 v = *(const f32x4*)(x + r * C + 4 * c4);
    v = ((v - mu) * is) * ga + be;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = vf_act_apply(v[e], act, slope);
    *(f32x4*)(y + r * C + 4 * c4) = v;
    if (planes) bn_store_planes(planes, pstride, r * C + 4 * c4, v, npl);
  }
}

__global__ void k_bn_eval_coeff(const float* __restrict__ running_mean, const float* __restrict__ running_var,
                                float* __restrict__ mean, float* __restrict__ invstd, int C, float eps) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  mean[c] = running_mean[c];
  invstd[c] = (float)(1.0 / sqrt((double)running_var[c] + (double)eps));
}

// ---- backward statistics: sum(g), sum(g * (x - mean)), g = gy masked by the fused activation
__global__ __launch_bounds__(256) void k_bn_bwd_stats(const float* __restrict__ x, const float* __restrict__ yact,
                                                      const float* __restrict__ gy, const float* __restrict__ mean,
                                                      double* __restrict__ part, int64_t npix, int C, int cq,
                                                      int rows_per_block, int act, float slope) {
  {
    const int64_t go = (int64_t)blockIdx.z * npix * C;      // batch group z
    x += go;
    gy += go;
    if (yact) yact += go;
    mean += (int64_t)blockIdx.z * C;
    part += (int64_t)blockIdx.z * gridDim.x * 2 * C;
  }
  const int rp = 256 / cq;
  const int tx = threadIdx.x % cq, ty = threadIdx.x / cq;
  const int c4 = blockIdx.y * cq + tx;
  const int C4 = C >> 2;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block, r1 = min(npix, r0 + rows_per_block);
  f32x4 s = {0, 0, 0, 0}, dp = {0, 0, 0, 0};
  if (c4 < C4) {
    const f32x4 mu = *(const f32x4*)(mean + 4 * c4);
    for (int64_t r = r0 + ty; r < r1; r += rp) {
      const int64_t o = r * C + 4 * c4;
      f32x4 g = *(const f32x4*)(gy + o);
      if (act != VF_ACT_NONE) {
        const f32x4 ya = *(const f32x4*)(yact + o);
#pragma unroll
        for (int e = 0; e < 4; ++e) g[e] = vf_act_grad(ya[e], g[e], act, slope);
      }
      const f32x4 xv = *(const f32x4*)(x + o);
      s += g;
      dp += g * (xv - mu);
    }
  }
  __shared__ f32x4 red[2][256];
  red[0][threadIdx.x] = s;
  red[1][threadIdx.x] = dp;
  __syncthreads();
  if (ty == 0 && c4 < C4) {
    double a[4] = {0, 0, 0, 0}, b[4] = {0, 0, 0, 0};
    for (int j = 0; j < rp; ++j) {
      const f32x4 u = red[0][j * cq + tx], w = red[1][j * cq + tx];
      for (int e = 0; e < 4; ++e) {
        a[e] += (double)u[e];
        b[e] += (double)w[e];
      }
    }
    double* o = part + (int64_t)blockIdx.x * 2 * C;
    for (int e = 0; e < 4; ++e) {
      o[4 * c4 + e] = a[e];
      o[C + 4 * c4 + e] = b[e];
    }
  }
}

// gx = (g - sum/n - (x-mean)*k) * invstd * gamma,  k = dotp*invstd^2/n   (THNN BatchNormalization_backward)
__global__ __launch_bounds__(256) void k_bn_bwd_apply(const float* __restrict__ x, const float* __restrict__ yact,
                                                      const float* __restrict__ gy, float* __restrict__ gx,
                                                      float* __restrict__ ggamma, float* __restrict__ gbeta,
                                                      const float* __restrict__ gamma, const float* __restrict__ mean,
                                                      const float* __restrict__ invstd, const double* __restrict__ sums,
                                                      int64_t npix, double n, int C, int cq, int rows_per_block, int act,
                                                      float slope, float pbeta, unsigned short* __restrict__ planes, int64_t pstride, int npl) {
  const int rp = 256 / cq;
  const int tx = threadIdx.x % cq, ty = threadIdx.x / cq;
  const int c4 = blockIdx.y * cq + tx;
  if (c4 >= (C >> 2)) return;
  if (blockIdx.x == 0 && blockIdx.z == 0 && ty == 0) {
    // gamma/beta gradients: the groups' contributions in group order, each rounded to fp32 and accumulated as the
    // separate backward calls would (pbeta on the first, 1 on the rest)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float gg = (ggamma && pbeta != 0.f) ? pbeta * ggamma[4 * c4 + e] : 0.f;
      float gb = (gbeta && pbeta != 0.f) ? pbeta * gbeta[4 * c4 + e] : 0.f;
      for (int g = 0; g < (int)gridDim.z; ++g) {
        const double sm = sums[(int64_t)g * 2 * C + 4 * c4 + e], dp = sums[(int64_t)g * 2 * C + C + 4 * c4 + e];
        gg = gg + (float)(dp * invstd[(int64_t)g * C + 4 * c4 + e]);
        gb = gb + (float)sm;
      }
      if (ggamma) ggamma[4 * c4 + e] = gg;
      if (gbeta) gbeta[4 * c4 + e] = gb;
    }
  }
  if (!gx) return;
  {
    const int64_t go = (int64_t)blockIdx.z * npix * C;      // batch group z
    x += go;
    gy += go;
    gx += go;
    if (yact) yact += go;
    if (planes) planes += go;
    mean += (int64_t)blockIdx.z * C;
    invstd += (int64_t)blockIdx.z * C;
    sums += (int64_t)blockIdx.z * 2 * C;
  }
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block, r1 = min(npix, r0 + rows_per_block);
  const f32x4 mu = *(const f32x4*)(mean + 4 * c4), is = *(const f32x4*)(invstd + 4 * c4);
  const f32x4 ga = gamma ? *(const f32x4*)(gamma + 4 * c4) : f32x4{1, 1, 1, 1};
  f32x4 kk, gm;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const double sum = sums[4 * c4 + e], dotp = sums[C + 4 * c4 + e];
    kk[e] = (float)(dotp * is[e] * is[e] / n);
    gm[e] = (float)(sum / n);
  }
  for (int64_t r = r0 + ty; r < r1; r += rp) {
    const int64_t o = r * C + 4 * c4;
    f32x4 g = *(const f32x4*)(gy + o);
    if (act != VF_ACT_NONE) {
      const f32x4 ya = *(const f32x4*)(yact + o);
#pragma unroll
      for (int e = 0; e < 4; ++e) g[e] = vf_act_grad(ya[e], g[e], act, slope);
    }
    const f32x4 xv = *(const f32x4*)(x + o);
    const f32x4 out = (g - gm - (xv - mu) * kk) * is * ga;
    *(f32x4*)(gx + o) = out;
    if (planes) bn_store_planes(planes, pstride, o, out, npl);
  }
}

// ---- every conv bias gradient of one backward walk in TWO launches (instead of two per layer): the column sums are
// not needed before Adam, and each layer's gradOutput lives in its module's own buffer until the walk is over.  The
// host builds the descriptor table once per (buffers, betas) combination and keeps it on the device.
struct VfColsumDesc {      // mirrored by video-filler_amd/backend.py (COLSUM_DESC); 64 bytes
  const float* g;          // [P][C] gradOutput
  float* gb;               // [C]   gradBias:  gb = beta*gb + column sums
  double* part;            // [gx][C] scratch partials
  int64_t P;
  int C, cq, rows_per_block, gx, gy;
  int blk1_off, blk2_off;  // first block of this layer in stage 1 / stage 2
  float beta;
};
static_assert(sizeof(VfColsumDesc) == 64, "descriptor layout is shared with the host mirror");

__device__ __forceinline__ int vf_find_layer(const VfColsumDesc* __restrict__ d, int n, int blk, bool stage2) {
  int l = 0;
  while (l + 1 < n && blk >= (stage2 ? d[l + 1].blk2_off : d[l + 1].blk1_off)) ++l;
  return l;
}
__global__ __launch_bounds__(256) void k_colsum4_multi(const VfColsumDesc* __restrict__ d, int n) {
  const int l = vf_find_layer(d, n, blockIdx.x, false);
  const VfColsumDesc L = d[l];
  const int local = blockIdx.x - L.blk1_off;
  const int bx = local % L.gx, by = local / L.gx;
  const int rp = 256 / L.cq;
  const int tx = threadIdx.x % L.cq, ty = threadIdx.x / L.cq;
  const int c4 = by * L.cq + tx;
  const int C4 = L.C >> 2;
  const int64_t r0 = (int64_t)bx * L.rows_per_block, r1 = min(L.P, r0 + L.rows_per_block);
  f32x4 s = {0, 0, 0, 0};
  if (c4 < C4)
    for (int64_t r = r0 + ty; r < r1; r += rp) s += *(const f32x4*)(L.g + r * L.C + 4 * c4);
  __shared__ f32x4 red[256];
  red[threadIdx.x] = s;
  __syncthreads();
  if (ty == 0 && c4 < C4) {
    double a[4] = {0, 0, 0, 0};
    for (int j = 0; j < rp; ++j) {
      const f32x4 u = red[j * L.cq + tx];
      for (int e = 0; e < 4; ++e) a[e] += (double)u[e];
    }
    double* o = L.part + (int64_t)bx * L.C;
    for (int e = 0; e < 4; ++e) o[4 * c4 + e] = a[e];
  }
}
__global__ __launch_bounds__(256) void k_reduce_bias_multi(const VfColsumDesc* __restrict__ d, int n) {
  const int l = vf_find_layer(d, n, blockIdx.x, true);
  const VfColsumDesc L = d[l];
  const int lane = threadIdx.x & 63, slot = threadIdx.x >> 6;
  const int col = (blockIdx.x - L.blk2_off) * 4 + slot;
  double s0 = 0, s1 = 0;
  if (col < L.C) {
    const double* p = L.part + col;
    int k = lane;
    for (; k + 64 < L.gx; k += 128) {
      s0 += p[(int64_t)k * L.C];
      s1 += p[(int64_t)(k + 64) * L.C];
    }
    for (; k < L.gx; k += 64) s0 += p[(int64_t)k * L.C];
  }
  double t = s0 + s1;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) t += __shfl_down(t, off, 64);
  if (lane == 0 && col < L.C) L.gb[col] = (L.beta != 0.f ? L.beta * L.gb[col] : 0.f) + (float)t;
}

// ================================================================================================ host
static int run_stats(vf_ctx* ctx, const BnGeom& g, double* sums, int C, int groups = 1) {
  hipLaunchKernelGGL((k_reduce_partials<0>), dim3((int)vf_cdiv(2 * C, 16), groups), dim3(256), 0, ctx->stream,
                     (const double*)vf_ws_ptr(ctx), g.gx, 2 * C, sums, (float*)nullptr, (float*)nullptr, (float*)nullptr,
                     (float*)nullptr, 0.0, 0.f, 0.f, (float*)nullptr, 0.f, 1);
  VF_LAUNCH_CHECK();
  return 0;
}

// gb[c] = beta*gb[c] + sum_p g[p][c]  — shared with vf_conv.hip (bias gradients)
int vf_internal_colsum(vf_ctx* ctx, const float* g, float* gb, int64_t P, int C, float beta) {
  VfProf prof(ctx, "bias_grad", 0.0, 4.0 * (double)P * C);
  double* part = (double*)vf_ws_ptr(ctx);
  int nslab;
  if (C % 4 == 0 && (((uintptr_t)g) & 15) == 0) {
    const BnGeom ge = bn_geom(P, C, bn_stat_blocks(P, C, 32768));
    nslab = ge.gx;
    VF_REQUIRE((size_t)(nslab + 1) * C * sizeof(double) <= vf_ws_avail(ctx), "workspace too small for bias-grad partials");
    hipLaunchKernelGGL(k_colsum4, dim3(ge.gx, ge.gy), dim3(256), 0, ctx->stream, g, part, P, C, ge.cq, ge.rows_per_block);
  } else {
    const int64_t rpb = std::max<int64_t>(256, vf_cdiv(P, std::max(1, 1024 / C)));
    nslab = (int)vf_cdiv(P, rpb);
    VF_REQUIRE((size_t)(nslab + 1) * C * sizeof(double) <= vf_ws_avail(ctx), "workspace too small for bias-grad partials");
    hipLaunchKernelGGL(k_colsum1, dim3(nslab, C), dim3(256), 0, ctx->stream, g, part, P, C, (int)rpb);
  }
  VF_LAUNCH_CHECK();
  hipLaunchKernelGGL((k_reduce_partials<2>), dim3((int)vf_cdiv(C, 16)), dim3(256), 0, ctx->stream, (const double*)part, nslab,
                     C, (double*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr, 0.0, 0.f, 0.f,
                     gb, beta, 1);
  VF_LAUNCH_CHECK();
  return 0;
}

VF_API int vf_bn_stats(vf_ctx* ctx, const float* x, const float* shift, double* sums, int64_t npix, int C) {
  VF_REQUIRE(C % 4 == 0, "BatchNorm channel count must be a multiple of 4 (got %d)", C);
  const BnGeom g = bn_geom(npix, C, bn_stat_blocks(npix, C, 32768));
  VF_REQUIRE((size_t)g.gx * 2 * C * sizeof(double) <= vf_ws_avail(ctx), "workspace too small for BN partials");
  VfProf prof(ctx, "bn_stats", 0.0, 4.0 * (double)npix * C);
  hipLaunchKernelGGL(k_bn_stats, dim3(g.gx, g.gy), dim3(256), 0, ctx->stream, x, shift, (double*)vf_ws_ptr(ctx), npix, C, g.cq,
                     g.rows_per_block);
  VF_LAUNCH_CHECK();
  return run_stats(ctx, g, sums, C);
}

VF_API int vf_bn_finalize(vf_ctx* ctx, const double* sums, float* running_mean, float* running_var, float* save_mean,
                          float* save_invstd, int64_t n_total, int C, float momentum, float eps) {
  hipLaunchKernelGGL(k_bn_finalize, dim3((int)vf_cdiv(C, 256)), dim3(256), 0, ctx->stream, sums, running_mean, running_var,
                     save_mean, save_invstd, (double)n_total, C, momentum, eps);
  VF_LAUNCH_CHECK();
  return 0;
}

VF_API int vf_bn_apply(vf_ctx* ctx, const float* x, float* y, const float* gamma, const float* beta, const float* mean,
                       const float* invstd, int64_t npix, int C, int act, float slope) {
  VF_REQUIRE(C % 4 == 0, "BatchNorm channel count must be a multiple of 4 (got %d)", C);
  const BnGeom g = bn_geom(npix, C);
  VfProf prof(ctx, "bn_apply", 0.0, 8.0 * (double)npix * C);
  hipLaunchKernelGGL(k_bn_apply, dim3(g.gx, g.gy), dim3(256), 0, ctx->stream, x, y, gamma, beta, mean, invstd, npix, C, g.cq,
                     g.rows_per_block, act, slope, (unsigned short*)nullptr, (int64_t)0, 3);
  VF_LAUNCH_CHECK();
  return 0;
}

static int bn_apply_groups(vf_ctx* ctx, const float* x, float* y, const float* gamma, const float* beta, const float* mean,
                           const float* invstd, int64_t npix, int C, int groups, int act, float slope, void* planes = nullptr) {
  VF_REQUIRE(C % 4 == 0, "BatchNorm channel count must be a multiple of 4 (got %d)", C);
  const BnGeom g = bn_geom(npix, C);
  VfProf prof(ctx, planes ? "bn_apply_planes" : "bn_apply", 0.0, (planes ? 14.0 : 8.0) * (double)npix * C * groups);
  hipLaunchKernelGGL(k_bn_apply, dim3(g.gx, g.gy, groups), dim3(256), 0, ctx->stream, x, y, gamma, beta, mean, invstd, npix, C,
                     g.cq, g.rows_per_block, act, slope, (unsigned short*)planes, npix * C * groups, ctx->mfma_bf16 == 1 ? 1 : 3);
  VF_LAUNCH_CHECK();
  return 0;
}

static int bn_train_fwd_groups(vf_ctx* ctx, const float* x, float* y, const float* gamma, const float* beta,
                               float* running_mean, float* running_var, float* save_mean, float* save_invstd, double* sums,
                               int64_t npix, int C, int groups, float momentum, float eps, int act, float slope,
                               void* y_planes = nullptr) {
  VF_REQUIRE(C % 4 == 0, "BatchNorm channel count must be a multiple of 4 (got %d)", C);
  VF_REQUIRE(groups >= 1 && groups <= 64, "BatchNorm batch groups: %d", groups);
  const BnGeom g = bn_geom(npix, C, bn_stat_blocks(npix, C, 32768));
  VF_REQUIRE((size_t)groups * g.gx * 2 * C * sizeof(double) <= vf_ws_avail(ctx), "workspace too small for BN partials");
  {
    VfProf prof(ctx, "bn_stats", 0.0, 4.0 * (double)npix * C * groups);
    hipLaunchKernelGGL(k_bn_stats, dim3(g.gx, g.gy, groups), dim3(256), 0, ctx->stream, x, (const float*)running_mean,
                       (double*)vf_ws_ptr(ctx), npix, C, g.cq, g.rows_per_block);
    VF_LAUNCH_CHECK();
    // second stage + finalize in one launch (the single-device path needs no hook between them)
    hipLaunchKernelGGL((k_reduce_partials<1>), dim3((int)vf_cdiv(C, 8)), dim3(256), 0, ctx->stream, (const double*)vf_ws_ptr(ctx),
                       g.gx, 2 * C, sums, running_mean, running_var, save_mean, save_invstd, (double)npix, momentum, eps,
                       (float*)nullptr, 0.f, groups);
    VF_LAUNCH_CHECK();
  }
  return bn_apply_groups(ctx, x, y, gamma, beta, save_mean, save_invstd, npix, C, groups, act, slope, y_planes);
}

VF_API int vf_bn_train_fwd(vf_ctx* ctx, const float* x, float* y, const float* gamma, const float* beta,
                           float* running_mean, float* running_var, float* save_mean, float* save_invstd, double* sums,
                           int64_t npix, int C, float momentum, float eps, int act, float slope) {
  return bn_train_fwd_groups(ctx, x, y, gamma, beta, running_mean, running_var, save_mean, save_invstd, sums, npix, C, 1, momentum,
                             eps, act, slope);
}

VF_API int vf_bn_train_fwd_groups(vf_ctx* ctx, const float* x, float* y, const float* gamma, const float* beta,
                                  float* running_mean, float* running_var, float* save_mean, float* save_invstd, double* sums,
                                  int64_t npix_per_group, int C, int groups, float momentum, float eps, int act, float slope) {
  return bn_train_fwd_groups(ctx, x, y, gamma, beta, running_mean, running_var, save_mean, save_invstd, sums, npix_per_group, C,
                             groups, momentum, eps, act, slope);
}

VF_API int vf_bn_eval_fwd(vf_ctx* ctx, const float* x, float* y, const float* gamma, const float* beta,
                          const float* running_mean, const float* running_var, int64_t npix, int C, float eps, int act,
                          float slope) {
  VF_REQUIRE((size_t)2 * C * sizeof(float) <= vf_ws_avail(ctx), "workspace too small");
  float* mean = (float*)vf_ws_ptr(ctx);
  float* invstd = mean + C;
  hipLaunchKernelGGL(k_bn_eval_coeff, dim3((int)vf_cdiv(C, 256)), dim3(256), 0, ctx->stream, running_mean, running_var, mean,
                     invstd, C, eps);
  VF_LAUNCH_CHECK();
  return vf_bn_apply(ctx, x, y, gamma, beta, mean, invstd, npix, C, act, slope);
}

static int bn_bwd_stats_groups(vf_ctx* ctx, const float* x, const float* y_act, const float* gy, const float* save_mean,
                               double* sums, int64_t npix, int C, int groups, int act, float slope) {
  VF_REQUIRE(C % 4 == 0, "BatchNorm channel count must be a multiple of 4 (got %d)", C);
  VF_REQUIRE(act == VF_ACT_NONE || y_act != nullptr, "fused activation backward needs the activated output");
  VF_REQUIRE(groups >= 1 && groups <= 64, "BatchNorm batch groups: %d", groups);
  const BnGeom g = bn_geom(npix, C, bn_stat_blocks(npix, C, 16384));
  VF_REQUIRE((size_t)groups * g.gx * 2 * C * sizeof(double) <= vf_ws_avail(ctx), "workspace too small for BN partials");
  VfProf prof(ctx, "bn_bwd_stats", 0.0, 4.0 * (double)npix * C * groups * (act != VF_ACT_NONE ? 3 : 2));
  hipLaunchKernelGGL(k_bn_bwd_stats, dim3(g.gx, g.gy, groups), dim3(256), 0, ctx->stream, x, y_act, gy, save_mean,
                     (double*)vf_ws_ptr(ctx), npix, C, g.cq, g.rows_per_block, act, slope);
  VF_LAUNCH_CHECK();
  return run_stats(ctx, g, sums, C, groups);
}

static int bn_bwd_apply_groups(vf_ctx* ctx, const float* x, const float* y_act, const float* gy, float* gx, float* ggamma,
                               float* gbeta, const float* gamma, const float* save_mean, const float* save_invstd,
                               const double* sums, int64_t npix, int64_t n_total, int C, int groups, int act, float slope,
                               float pbeta, void* planes = nullptr) {
  VF_REQUIRE(C % 4 == 0, "BatchNorm channel count must be a multiple of 4 (got %d)", C);
  VF_REQUIRE(!planes || gx, "gradient planes go with the gradient itself");
  const BnGeom g = bn_geom(npix, C);
  VfProf prof(ctx, planes ? "bn_bwd_apply_planes" : "bn_bwd_apply", 0.0,
              gx ? (double)npix * C * groups * (4.0 * (act != VF_ACT_NONE ? 4 : 3) + (planes ? 6.0 : 0.0)) : 0.0);
  hipLaunchKernelGGL(k_bn_bwd_apply, dim3(g.gx, g.gy, groups), dim3(256), 0, ctx->stream, x, y_act, gy, gx, ggamma, gbeta, gamma,
                     save_mean, save_invstd, sums, npix, (double)n_total, C, g.cq, g.rows_per_block, act, slope, pbeta,
                     (unsigned short*)planes, npix * C * groups, ctx->mfma_bf16 == 1 ? 1 : 3);
  VF_LAUNCH_CHECK();
  return 0;
}

VF_API int vf_bn_bwd_stats(vf_ctx* ctx, const float* x, const float* y_act, const float* gy, const float* save_mean,
                           double* sums, int64_t npix, int C, int act, float slope) {
  return bn_bwd_stats_groups(ctx, x, y_act, gy, save_mean, sums, npix, C, 1, act, slope);
}

VF_API int vf_bn_bwd_apply(vf_ctx* ctx, const float* x, const float* y_act, const float* gy, float* gx, float* ggamma,
                           float* gbeta, const float* gamma, const float* save_mean, const float* save_invstd,
                           const double* sums, int64_t npix, int64_t n_total, int C, int act, float slope, float pbeta) {
  return bn_bwd_apply_groups(ctx, x, y_act, gy, gx, ggamma, gbeta, gamma, save_mean, save_invstd, sums, npix, n_total, C, 1, act,
                             slope, pbeta);
}

VF_API int vf_bn_bwd(vf_ctx* ctx, const float* x, const float* y_act, const float* gy, float* gx, float* ggamma,
                     float* gbeta, const float* gamma, const float* save_mean, const float* save_invstd, double* sums,
                     int64_t npix, int C, int act, float slope, float pbeta) {
  if (int rc = bn_bwd_stats_groups(ctx, x, y_act, gy, save_mean, sums, npix, C, 1, act, slope)) return rc;
  return bn_bwd_apply_groups(ctx, x, y_act, gy, gx, ggamma, gbeta, gamma, save_mean, save_invstd, sums, npix, npix, C, 1, act,
                             slope, pbeta);
}

VF_API int vf_bn_bwd_groups(vf_ctx* ctx, const float* x, const float* y_act, const float* gy, float* gx, float* ggamma,
                            float* gbeta, const float* gamma, const float* save_mean, const float* save_invstd, double* sums,
                            int64_t npix_per_group, int C, int groups, int act, float slope, float pbeta) {
  if (int rc = bn_bwd_stats_groups(ctx, x, y_act, gy, save_mean, sums, npix_per_group, C, groups, act, slope)) return rc;
  return bn_bwd_apply_groups(ctx, x, y_act, gy, gx, ggamma, gbeta, gamma, save_mean, save_invstd, sums, npix_per_group,
                             npix_per_group, C, groups, act, slope, pbeta);
}

// the grouped forms that also write the bf16 planes of their output (for a vf_pconv_* consumer)
VF_API int vf_bn_train_fwd_planes(vf_ctx* ctx, const float* x, float* y, const float* gamma, const float* beta,
                                  float* running_mean, float* running_var, float* save_mean, float* save_invstd, double* sums,
                                  int64_t npix_per_group, int C, int groups, float momentum, float eps, int act, float slope,
                                  void* y_planes) {
  return bn_train_fwd_groups(ctx, x, y, gamma, beta, running_mean, running_var, save_mean, save_invstd, sums, npix_per_group, C,
                             groups, momentum, eps, act, slope, y_planes);
}
VF_API int vf_bn_bwd_planes(vf_ctx* ctx, const float* x, const float* y_act, const float* gy, float* gx, float* ggamma,
                            float* gbeta, const float* gamma, const float* save_mean, const float* save_invstd, double* sums,
                            int64_t npix_per_group, int C, int groups, int act, float slope, float pbeta, void* gx_planes) {
  if (int rc = bn_bwd_stats_groups(ctx, x, y_act, gy, save_mean, sums, npix_per_group, C, groups, act, slope)) return rc;
  return bn_bwd_apply_groups(ctx, x, y_act, gy, gx, ggamma, gbeta, gamma, save_mean, save_invstd, sums, npix_per_group,
                             npix_per_group, C, groups, act, slope, pbeta, gx_planes);
}

// ---- statistics already summed per tile by the GEMM that produced the tensor (vf_bn_fuse_next_fwd / _bwd, VfBnSt):
// `part` = [groups][rows_per_group][2][C] doubles.  Forward: finalize (+ running statistics) and normalise, two launches;
// backward: column totals and the gradient pass, two launches — the tensor is never read just to be summed.
VF_API int vf_bn_train_fwd_pre(vf_ctx* ctx, const double* part, int rows_per_group, const float* x, float* y, const float* gamma,
                               const float* beta, float* running_mean, float* running_var, float* save_mean, float* save_invstd,
                               double* sums, int64_t npix_per_group, int C, int groups, float momentum, float eps, int act,
                               float slope, void* y_planes) {
  VF_REQUIRE(C % 4 == 0, "BatchNorm channel count must be a multiple of 4 (got %d)", C);
  VF_REQUIRE(part && rows_per_group > 0 && groups >= 1 && groups <= 64, "vf_bn_train_fwd_pre: bad partials");
  {
    VfProf prof(ctx, "bn_finalize", 0.0, 8.0 * (double)rows_per_group * 2 * C * groups);
    hipLaunchKernelGGL((k_reduce_partials<1>), dim3((int)vf_cdiv(C, 8)), dim3(256), 0, ctx->stream, part, rows_per_group, 2 * C, sums,
                       running_mean, running_var, save_mean, save_invstd, (double)npix_per_group, momentum, eps, (float*)nullptr, 0.f,
                       groups);
    VF_LAUNCH_CHECK();
  }
  return bn_apply_groups(ctx, x, y, gamma, beta, save_mean, save_invstd, npix_per_group, C, groups, act, slope, y_planes);
}
VF_API int vf_bn_bwd_pre(vf_ctx* ctx, const double* part, int rows_per_group, const float* x, const float* g_masked, float* gx,
                         float* ggamma, float* gbeta, const float* gamma, const float* save_mean, const float* save_invstd,
                         double* sums, int64_t npix_per_group, int C, int groups, float pbeta, void* gx_planes) {
  VF_REQUIRE(C % 4 == 0, "BatchNorm channel count must be a multiple of 4 (got %d)", C);
  VF_REQUIRE(part && rows_per_group > 0 && groups >= 1 && groups <= 64, "vf_bn_bwd_pre: bad partials");
  {
    VfProf prof(ctx, "bn_bwd_finalize", 0.0, 8.0 * (double)rows_per_group * 2 * C * groups);
    hipLaunchKernelGGL((k_reduce_partials<0>), dim3((int)vf_cdiv(2 * C, 16), groups), dim3(256), 0, ctx->stream, part, rows_per_group,
                       2 * C, sums, (float*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr, 0.0, 0.f, 0.f, (float*)nullptr,
                       0.f, 1);
    VF_LAUNCH_CHECK();
  }
  return bn_bwd_apply_groups(ctx, x, nullptr, g_masked, gx, ggamma, gbeta, gamma, save_mean, save_invstd, sums, npix_per_group,
                             npix_per_group, C, groups, VF_ACT_NONE, 0.f, pbeta, gx_planes);
}

// ---- all bias gradients of a backward walk (see VfColsumDesc)
VF_API int vf_bias_grad_plan(int64_t P, int C, int* cq, int* rows_per_block, int* gx, int* gy) {
  VF_REQUIRE(P > 0 && C > 0 && C % 4 == 0, "vf_bias_grad_plan: C must be a positive multiple of 4 (got %d)", C);
  const BnGeom g = bn_geom(P, C, bn_stat_blocks(P, C, 32768));
  *cq = g.cq; *rows_per_block = g.rows_per_block; *gx = g.gx; *gy = g.gy;
  return 0;
}
VF_API int vf_bias_grad_multi(vf_ctx* ctx, const void* desc_dev, int n, int blocks1, int blocks2) {
  VF_REQUIRE(desc_dev != nullptr && n > 0 && blocks1 > 0 && blocks2 > 0, "vf_bias_grad_multi: empty plan");
  VfProf prof(ctx, "bias_grad_multi", 0.0, 0.0);
  hipLaunchKernelGGL(k_colsum4_multi, dim3(blocks1), dim3(256), 0, ctx->stream, (const VfColsumDesc*)desc_dev, n);
  VF_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_reduce_bias_multi, dim3(blocks2), dim3(256), 0, ctx->stream, (const VfColsumDesc*)desc_dev, n);
  VF_LAUNCH_CHECK();
  return 0;
}
