// vf_conv_generic.hip — nn.SpatialConvolution of ANY kernel size / stride / padding, NHWC fp32.
//
// The reference's main nets are 4x4 convolutions on power-of-two maps (vf_conv.hip: implicit GEMM on the matrix cores).
// Its option branches add three other shapes, all thin:
//   * conditionAdv (train.lua:158-170): nc -> ndf, 5x5, stride 2, pad 2 on the 128x128 context and pad 2+32 on the
//     64x64 prediction (3 input channels: K = 75);
//   * noiseGen (train.lua:109-113): nz -> nz, 1x1 on the 1x1 noise map (a [B, nz] x [nz, nz] product).
// These are a few percent of an iteration's FLOPs with K far below an MFMA tile's depth; they run on the vector ALUs
// with the weights staged in LDS.  Same weight layout as everywhere else: [Cout][kh][kw][Cin].
#include <algorithm>

#include "vf_common.h"

int vf_internal_colsum(vf_ctx* ctx, const float* g, float* gb, int64_t P, int C, float beta);

namespace {

struct GConv {
  int B, H, W, Cin, Cout, k, stride, pad, Ho, Wo;
};

constexpr int GC_LDS_FLOATS = 12 * 1024;   // 48 KB of weights per block

// ---------------------------------------------------------------------------------------------------- forward
// One thread per (output pixel, output channel); a wave covers 64 consecutive channels of one pixel (or of the next
// pixels when Cout < 64), so the input taps are wave-uniform or nearly so (served by one transaction) and the weights
// are read from LDS as [K][Cout] rows: consecutive lanes, consecutive banks.
__global__ void __launch_bounds__(256) k_gconv_fwd(const float* __restrict__ x, const float* __restrict__ w,
                                                   const float* __restrict__ bias, float* __restrict__ y, GConv g,
                                                   int act, float slope, int w_in_lds) {
  extern __shared__ float wl[];   // [K][Cout]
  const int K = g.k * g.k * g.Cin;
  if (w_in_lds) {
    for (int i = threadIdx.x; i < K * g.Cout; i += blockDim.x) {
      const int co = i / K, kk = i - co * K;
      wl[kk * g.Cout + co] = w[i];
    }
    __syncthreads();
  }
  const int64_t total = (int64_t)g.B * g.Ho * g.Wo * g.Cout;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int co = (int)(i % g.Cout);
    int64_t p = i / g.Cout;
    const int ox = (int)(p % g.Wo);
    p /= g.Wo;
    const int oy = (int)(p % g.Ho);
    const int b = (int)(p / g.Ho);
    float acc = bias ? bias[co] : 0.f;
    const int iy0 = oy * g.stride - g.pad, ix0 = ox * g.stride - g.pad;
    for (int kh = 0; kh < g.k; ++kh) {
      const int iy = iy0 + kh;
      if (iy < 0 || iy >= g.H) continue;
      for (int kw = 0; kw < g.k; ++kw) {
        const int ix = ix0 + kw;
        if (ix < 0 || ix >= g.W) continue;
        const float* xp = x + (((int64_t)b * g.H + iy) * g.W + ix) * g.Cin;
        const int kb = (kh * g.k + kw) * g.Cin;
        if (w_in_lds) {
          for (int c = 0; c < g.Cin; ++c) acc = fmaf(xp[c], wl[(kb + c) * g.Cout + co], acc);
        } else {
          const float* wp = w + (int64_t)co * K + kb;
          for (int c = 0; c < g.Cin; ++c) acc = fmaf(xp[c], wp[c], acc);
        }
      }
    }
    y[i] = vf_act_apply(acc, act, slope);
  }
}

// ---------------------------------------------------------------------------------------------------- data gradient
// gx[b,iy,ix,ci] = sum over the taps (kh,kw) with (iy+pad-kh) % stride == 0 (same for x) and the output pixel inside
// the map, of sum_co gy[b,oy,ox,co] * w[co][kh][kw][ci].  One thread per input element; weights in LDS as
// [kh][kw][ci][co] so the inner loop over co walks both operands contiguously.
__global__ void __launch_bounds__(256) k_gconv_bwd_data(const float* __restrict__ gy, const float* __restrict__ w,
                                                        float* __restrict__ gx, GConv g, int w_in_lds) {
  extern __shared__ float wl[];   // [k*k*Cin][Cout]
  const int K = g.k * g.k * g.Cin;
  if (w_in_lds) {
    for (int i = threadIdx.x; i < K * g.Cout; i += blockDim.x) {
      const int co = i / K, kk = i - co * K;
      wl[kk * g.Cout + co] = w[i];
    }
    __syncthreads();
  }
  const int64_t total = (int64_t)g.B * g.H * g.W * g.Cin;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int ci = (int)(i % g.Cin);
    int64_t p = i / g.Cin;
    const int ix = (int)(p % g.W);
    p /= g.W;
    const int iy = (int)(p % g.H);
    const int b = (int)(p / g.H);
    float acc = 0.f;
    for (int kh = 0; kh < g.k; ++kh) {
      const int ty = iy + g.pad - kh;
      if (ty < 0 || ty % g.stride) continue;
      const int oy = ty / g.stride;
      if (oy >= g.Ho) continue;
      for (int kw = 0; kw < g.k; ++kw) {
        const int tx = ix + g.pad - kw;
        if (tx < 0 || tx % g.stride) continue;
        const int ox = tx / g.stride;
        if (ox >= g.Wo) continue;
        const float* gp = gy + (((int64_t)b * g.Ho + oy) * g.Wo + ox) * g.Cout;
        const int kk = (kh * g.k + kw) * g.Cin + ci;
        if (w_in_lds) {
          const float* wp = wl + kk * g.Cout;
          for (int co = 0; co < g.Cout; ++co) acc = fmaf(gp[co], wp[co], acc);
        } else {
          for (int co = 0; co < g.Cout; ++co) acc = fmaf(gp[co], w[(int64_t)co * K + kk], acc);
        }
      }
    }
    gx[i] = acc;
  }
}

// ---------------------------------------------------------------------------------------------------- weight gradient
// gw[co][kk] = sum_p gy[p][co] * patch(p)[kk],  kk = (kh,kw,ci).  Block (bx, by): pixels [bx*PB, (bx+1)*PB), outputs
// [by*OUT_PER_BLOCK, ...).  Pixels are staged GC_PT at a time (their gy rows and their zero-padded input patches) in
// LDS; each thread carries GC_ACC accumulators.  Partial sums go to slab[bx][Cout*K]; a second kernel adds the slabs
// in a fixed order (deterministic).
constexpr int GC_PT = 16;
constexpr int GC_ACC = 8;
constexpr int GC_OUT_PER_BLOCK = 256 * GC_ACC;

__global__ void __launch_bounds__(256) k_gconv_bwd_weight(const float* __restrict__ x, const float* __restrict__ gy,
                                                          float* __restrict__ slab, GConv g, int pix_per_block) {
  extern __shared__ float sh[];   // gy rows [GC_PT][Cout] then patches [GC_PT][K]
  const int K = g.k * g.k * g.Cin;
  const int nout = g.Cout * K;
  float* sg = sh;
  float* sp = sh + GC_PT * g.Cout;
  const int64_t P = (int64_t)g.B * g.Ho * g.Wo;
  const int64_t p0 = (int64_t)blockIdx.x * pix_per_block;
  const int64_t p1 = p0 + pix_per_block < P ? p0 + pix_per_block : P;
  const int o0 = blockIdx.y * GC_OUT_PER_BLOCK;
  int oco[GC_ACC], okk[GC_ACC];
  float acc[GC_ACC];
#pragma unroll
  for (int a = 0; a < GC_ACC; ++a) {
    const int o = o0 + a * 256 + threadIdx.x;
    oco[a] = o < nout ? o / K : -1;
    okk[a] = o < nout ? o - (o / K) * K : 0;
    acc[a] = 0.f;
  }
  for (int64_t pb = p0; pb < p1; pb += GC_PT) {
    const int np = (int)(p1 - pb < GC_PT ? p1 - pb : GC_PT);
    __syncthreads();
    for (int i = threadIdx.x; i < np * g.Cout; i += blockDim.x) sg[i] = gy[pb * g.Cout + i];
    for (int i = threadIdx.x; i < np * K; i += blockDim.x) {
      const int q = i / K, kk = i - q * K;
      const int ci = kk % g.Cin, t = kk / g.Cin;
      const int kw = t % g.k, kh = t / g.k;
      int64_t p = pb + q;
      const int ox = (int)(p % g.Wo);
      p /= g.Wo;
      const int oy = (int)(p % g.Ho);
      const int b = (int)(p / g.Ho);
      const int iy = oy * g.stride - g.pad + kh, ix = ox * g.stride - g.pad + kw;
      float v = 0.f;
      if (iy >= 0 && iy < g.H && ix >= 0 && ix < g.W) v = x[(((int64_t)b * g.H + iy) * g.W + ix) * g.Cin + ci];
      sp[i] = v;
    }
    __syncthreads();
#pragma unroll
    for (int a = 0; a < GC_ACC; ++a) {
      if (oco[a] < 0) continue;
      float s = acc[a];
      for (int q = 0; q < np; ++q) s = fmaf(sg[q * g.Cout + oco[a]], sp[q * K + okk[a]], s);
      acc[a] = s;
    }
  }
#pragma unroll
  for (int a = 0; a < GC_ACC; ++a) {
    const int o = o0 + a * 256 + threadIdx.x;
    if (o < nout) slab[(int64_t)blockIdx.x * nout + o] = acc[a];
  }
}

__global__ void k_gconv_slab_sum(const float* __restrict__ slab, int nslab, int n, float* __restrict__ gw, float beta) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float s = 0.f;
  for (int k = 0; k < nslab; ++k) s += slab[(int64_t)k * n + i];
  gw[i] = beta != 0.f ? beta * gw[i] + s : s;
}

GConv make(int B, int H, int W, int Cin, int Cout, int k, int stride, int pad) {
  GConv g{B, H, W, Cin, Cout, k, stride, pad, (H + 2 * pad - k) / stride + 1, (W + 2 * pad - k) / stride + 1};
  return g;
}

int check(const GConv& g) {
  VF_REQUIRE(g.B > 0 && g.Cin > 0 && g.Cout > 0 && g.k > 0 && g.stride > 0 && g.pad >= 0, "bad convolution sizes");
  VF_REQUIRE(g.H + 2 * g.pad >= g.k && g.W + 2 * g.pad >= g.k, "kernel larger than the padded input");
  VF_REQUIRE((int64_t)g.Cout * g.k * g.k * g.Cin < ((int64_t)1 << 28), "weight tensor too large for the generic path");
  return 0;
}

int grid_for(int64_t n) { return (int)std::min<int64_t>(vf_cdiv(n, 256), 256 * 32); }

}  // namespace

int vf_internal_gconv_fwd(vf_ctx* ctx, const float* x, const float* w, const float* bias, float* y, int B, int H, int W, int Cin,
                          int Cout, int k, int stride, int pad, int act, float slope) {
  const GConv g = make(B, H, W, Cin, Cout, k, stride, pad);
  if (int rc = check(g)) return rc;
  const int64_t nw = (int64_t)Cout * k * k * Cin;
  const int in_lds = nw <= GC_LDS_FLOATS;
  const int64_t total = (int64_t)B * g.Ho * g.Wo * Cout;
  VfProf prof(ctx, "gconv_fwd", 2.0 * (double)total * k * k * Cin, 4.0 * ((double)B * H * W * Cin + (double)total + (double)nw));
  hipLaunchKernelGGL(k_gconv_fwd, dim3(grid_for(total)), dim3(256), in_lds ? (size_t)nw * 4 : 0, ctx->stream, x, w, bias, y, g, act,
                     slope, in_lds);
  VF_LAUNCH_CHECK();
  return 0;
}

int vf_internal_gconv_bwd_data(vf_ctx* ctx, const float* gy, const float* w, float* gx, int B, int H, int W, int Cin, int Cout,
                               int k, int stride, int pad) {
  const GConv g = make(B, H, W, Cin, Cout, k, stride, pad);
  if (int rc = check(g)) return rc;
  const int64_t nw = (int64_t)Cout * k * k * Cin;
  const int in_lds = nw <= GC_LDS_FLOATS;
  const int64_t total = (int64_t)B * H * W * Cin;
  VfProf prof(ctx, "gconv_bwd_data", 2.0 * (double)B * g.Ho * g.Wo * Cout * k * k * Cin,
              4.0 * ((double)total + (double)B * g.Ho * g.Wo * Cout + (double)nw));
  hipLaunchKernelGGL(k_gconv_bwd_data, dim3(grid_for(total)), dim3(256), in_lds ? (size_t)nw * 4 : 0, ctx->stream, gy, w, gx, g,
                     in_lds);
  VF_LAUNCH_CHECK();
  return 0;
}

int vf_internal_gconv_bwd_weight(vf_ctx* ctx, const float* x, const float* gy, float* gw, float* gb, int B, int H, int W, int Cin,
                                 int Cout, int k, int stride, int pad, float beta) {
  const GConv g = make(B, H, W, Cin, Cout, k, stride, pad);
  if (int rc = check(g)) return rc;
  const int K = k * k * Cin;
  const int nout = Cout * K;
  const int64_t P = (int64_t)B * g.Ho * g.Wo;
  const size_t lds = (size_t)GC_PT * (Cout + K) * 4;
  VF_REQUIRE(lds <= 64 * 1024, "generic weight gradient: Cout + k*k*Cin = %d exceeds the staging tile", Cout + K);
  // enough blocks to fill the chip, but at least 4 staging rounds per block and slabs that fit the workspace
  const int ychunks = (int)vf_cdiv(nout, GC_OUT_PER_BLOCK);
  int64_t nslab = std::max<int64_t>(1, std::min<int64_t>(vf_cdiv(P, 4 * GC_PT), 1024 / ychunks));
  const size_t per_slab = (size_t)nout * 4;
  nslab = std::min<int64_t>(nslab, std::max<int64_t>(1, (int64_t)(vf_ws_avail(ctx) / per_slab)));
  VF_REQUIRE(per_slab <= vf_ws_avail(ctx), "workspace too small for the generic weight gradient's partial sums");
  const int ppb = (int)vf_cdiv(vf_cdiv(P, nslab), GC_PT) * GC_PT;
  nslab = vf_cdiv(P, ppb);
  float* slab = (float*)vf_ws_ptr(ctx);
  {
    VfProf prof(ctx, "gconv_bwd_weight", 2.0 * (double)P * nout, 4.0 * ((double)P * Cout + (double)B * H * W * Cin + (double)nout));
    hipLaunchKernelGGL(k_gconv_bwd_weight, dim3((int)nslab, ychunks), dim3(256), lds, ctx->stream, x, gy, slab, g, ppb);
    VF_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_gconv_slab_sum, dim3((int)vf_cdiv(nout, 256)), dim3(256), 0, ctx->stream, (const float*)slab, (int)nslab,
                       nout, gw, beta);
    VF_LAUNCH_CHECK();
  }
  if (gb) return vf_internal_colsum(ctx, gy, gb, P, Cout, beta);
  return 0;
}
