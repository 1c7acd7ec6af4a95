// vf_conv_thin.hip — the image-side convolutions: 3 input channels (netG's first conv train.lua:89, netD's first conv :183).
//
// K = 16 taps x 3 channels = 48: an MFMA tile would be mostly padding, and the implicit-GEMM kernels' scalar-gather path
// (vf_conv.hip, V = 0) spends its time on per-element address arithmetic — 52 us for 0.4 GFLOP whose cost is really the
// 67 MB output write.  This is a direct convolution instead: a block owns 8 x 8 output pixels of one image and 64 output
// channels; the 18 x 18 pixel input patch is staged once in LDS (one float4 per pixel, zero-filled at the image border), every thread keeps the
// 2 x 48 weights of its two output channels in registers and walks 8 pixels, reading the patch as wave-uniform LDS
// broadcasts; a pixel's 64 channels leave as one 256-byte row (bias + LeakyReLU/ReLU fused), optionally with the three
// bf16 planes of the output beside it for the planes-fed convolution that consumes it (vf_pgemm.hip).
// fp32 FMA chains in (kh, kw, c) order: the parity bars of the implicit-GEMM path (tests/test_gpu_conv_sweep.py).
#include <algorithm>
#include <cstdlib>

#include "vf_common.h"

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int TP = 8;                    // output pixels per tile side
constexpr int PW = 2 * TP + 2;           // patch side (stride 2, pad 1, 4 taps)

template <int CIN>
__global__ __launch_bounds__(256) void k_conv_thin_in(const float* __restrict__ x, const float* __restrict__ w,
                                                      const float* __restrict__ bias, float* __restrict__ y,
                                                      unsigned short* __restrict__ planes, int64_t pstride, int H, int W, int Cout,
                                                      int tiles_x, int tiles_y, float neg, unsigned* __restrict__ bits) {
  static_assert(CIN <= 4, "pixels are padded to one float4 in LDS");
  constexpr int K = 16 * CIN, ROW4 = PW;                 // patch rows of PW float4 pixels (channels CIN..3 unused)
  constexpr int PP = (PW * ROW4 + 255) / 256;            // patch pixels per thread
  __shared__ f32x4 patch[PW * ROW4];
  __shared__ float wl[64 * (K + 1)];
  const int tid = threadIdx.x;
  // a block walks one ROW of tiles (tiles_x of them): the weights reach its registers once, and the next tile's patch is
  // on its way while the current one is computed (with one tile per block the kernel was all prologue: 30 of 55 us)
  const int ty = blockIdx.x % tiles_y, b = blockIdx.x / tiles_y;
  const int n0 = blockIdx.y * 64;
  const int iy0 = 2 * TP * ty - 1;
  const float* xb = x + (int64_t)b * H * W * CIN;
  f32x4 pre[PP];
  auto fetch = [&](int tx) {       // this thread's pixels of tile tx's patch -> registers (zeros outside the image)
    const int ix0 = 2 * TP * tx - 1;
#pragma unroll
    for (int j = 0; j < PP; ++j) {
      const int i = tid + 256 * j;
      const int r = i / ROW4, c = i - r * ROW4;
      const int iy = iy0 + r, ix = ix0 + c;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (i < PW * ROW4 && tx < tiles_x && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) {
        const float* px = xb + ((int64_t)iy * W + ix) * CIN;
#pragma unroll
        for (int e = 0; e < CIN; ++e) v[e] = px[e];
      }
      pre[j] = v;
    }
  };
  auto stash = [&]() {
#pragma unroll
    for (int j = 0; j < PP; ++j) {
      const int i = tid + 256 * j;
      if (i < PW * ROW4) patch[i] = pre[j];
    }
  };
  fetch(0);
  // ---- this thread's two output channels: 2 x K weights in registers, by way of LDS (the 64 x K block of the weight matrix is
  // contiguous in memory: coalesced loads; a thread fetching its own two rows straight from memory touched 32 lines per load)
  for (int i = tid; i < 64 * K; i += 256) wl[(i / K) * (K + 1) + (i % K)] = w[(int64_t)n0 * K + i];
  stash();
  const int n2 = tid & 31, pg = tid >> 5;
  const int n = n0 + 2 * n2;
  __syncthreads();
  float w0[K], w1[K];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    w0[k] = wl[(2 * n2) * (K + 1) + k];
    w1[k] = wl[(2 * n2 + 1) * (K + 1) + k];
  }
  const float b0 = bias ? bias[n] : 0.f, b1 = bias ? bias[n + 1] : 0.f;
  const int Ho = H >> 1, Wo = W >> 1;
  const int oy = TP * ty + pg;
  for (int tx = 0; tx < tiles_x; ++tx) {
    fetch(tx + 1);
#pragma unroll 2
    for (int px = 0; px < TP; ++px) {
      float a0 = b0, a1 = b1;
#pragma unroll
      for (int kh = 0; kh < 4; ++kh) {
        const f32x4* pr = patch + (2 * pg + kh) * ROW4 + 2 * px;      // 4 taps, one float4 each (wave-uniform per half: broadcast)
#pragma unroll
        for (int kw = 0; kw < 4; ++kw) {
          const f32x4 xv = pr[kw];
#pragma unroll
          for (int c = 0; c < CIN; ++c) {
            a0 = fmaf(xv[c], w0[(kh * 4 + kw) * CIN + c], a0);
            a1 = fmaf(xv[c], w1[(kh * 4 + kw) * CIN + c], a1);
          }
        }
      }
      a0 = a0 * (a0 > 0.f ? 1.f : neg);
      a1 = a1 * (a1 > 0.f ? 1.f : neg);
      const int ox = TP * tx + px;
      const int64_t o = (((int64_t)b * Ho + oy) * Wo + ox) * Cout + n;
      f32x2 v = {a0, a1};
      *(f32x2*)(y + o) = v;
      if (bits) {
        // the sign bits a later data-gradient pass masks with (ctx->act_bits_out): lanes 0-31 / 32-63 of a wave hold the 64
        // channels of ONE pixel each (two per lane): word h of the pixel's group = the ballot over channels 2j + h
        const unsigned long long k0 = __ballot(a0 > 0.f), k1 = __ballot(a1 > 0.f);
        if ((tid & 31) == 0) {
          const int sh = tid & 32;
          unsigned* bw = bits + (o - n) / Cout * (Cout >> 5) + 2 * blockIdx.y;      // (o - n) / Cout: the pixel index
          bw[0] = (unsigned)(k0 >> sh);
          bw[1] = (unsigned)(k1 >> sh);
        }
      }
      if (planes) {
        float r0 = a0, r1 = a1;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          const unsigned u0 = __float_as_uint(r0), u1 = __float_as_uint(r1);
          *(unsigned*)(planes + q * pstride + o) = __builtin_amdgcn_perm(u1, u0, 0x07060302u);
          if (q < 2) {
            r0 -= __uint_as_float(u0 & 0xffff0000u);
            r1 -= __uint_as_float(u1 & 0xffff0000u);
          }
        }
      }
    }
    __syncthreads();        // everyone has read this tile's patch
    stash();
    __syncthreads();
  }
}

// ---- thin OUTPUT: the transposed 4x4 stride-2 pass onto <= 4 channels — netG's last full-conv forward (64 -> 3, Tanh;
// train.lua:146) and the data-gradient of netD's first conv (train.lua:183, needed by fGx's netD:updateGradInput, :366).
// N = 3: an MFMA tile is 13/16 padding and K = 64 is two K steps — the implicit GEMM took 40 us for 0.4 GFLOP, plus a col2im
// pass (24 us).  Direct form: a thread owns ONE low-resolution position (b, my, mx) and produces its 2 x 2 output pixels x N
// channels (12 accumulators): the 3 x 3 input neighbourhood meets all 16 filter taps exactly once (9 (pixel) x {1, 2, 4}
// (parity classes) = 16 products per channel), so per input channel a thread does 9 LDS reads and 16 * N FMAs whose weight operand
// is the SAME for every thread — 16 * N consecutive floats of w[c], read from LDS as broadcasts.  One wave per block: 8 x 8 positions,
// the 10 x 10 x 64-channel patch staged in LDS per channel chunk (pixel pitch 68 floats: conflict-free 16-byte reads).
// fp32 FMA chains; written once, coalesced enough (24 contiguous bytes per thread and output row).
template <int N>
__global__ __launch_bounds__(256) void k_deconv_thin_out(const float* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ bias, float* __restrict__ y, int Hi, int Wi, int C,
                                                         int tiles_x, int tiles_y, int act, float slope) {
  // 8 x 8 positions per block, FOUR waves: wave k walks channels [16k, 16k + 16) of every 64-channel chunk for all 64 positions
  // (a lone wave per SIMD sat out every LDS round trip: 28 us; four of them cover each other) and the partial sums meet in LDS,
  // where wave k finishes parity class k (its N channels: bias, activation, store)
  constexpr int TPO = 8, PWO = TPO + 2, CC = 64, PS = CC + 4, CW = CC / 4;
  __shared__ __attribute__((aligned(16))) float patch[PWO * PWO * PS];      // (reused for the partial sums: 4 x 4N x 64 floats)
  static_assert(4 * 4 * N * 64 <= PWO * PWO * PS, "the partial sums fit the patch buffer");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int t = blockIdx.x;
  const int tx = t % tiles_x;
  t /= tiles_x;
  const int tyi = t % tiles_y, b = t / tiles_y;
  const int py = lane >> 3, px = lane & 7;
  const int my = tyi * TPO + py, mx = tx * TPO + px;
  float acc[4][N];
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int n = 0; n < N; ++n) acc[k][n] = 0.f;
  for (int c0 = 0; c0 < C; c0 += CC) {
    __syncthreads();
    // every load of the chunk in flight before the first LDS write (a load -> wait -> write loop cost 13 memory round trips
    // per chunk: 36 us per launch)
    constexpr int TOTX = PWO * PWO * (CC / 4), NX = (TOTX + 255) / 256;
    f32x4 vx[NX];
#pragma unroll
    for (int r = 0; r < NX; ++r) {
      const int idx = tid + 256 * r, pix = idx / (CC / 4), q = idx % (CC / 4);
      const int iy = tyi * TPO - 1 + pix / PWO, ix = tx * TPO - 1 + pix % PWO;
      const bool ok = idx < TOTX && (unsigned)iy < (unsigned)Hi && (unsigned)ix < (unsigned)Wi;
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
      vx[r] = ok ? *(const f32x4*)(x + (((int64_t)b * Hi + iy) * Wi + ix) * C + c0 + 4 * q) : z;
    }
#pragma unroll
    for (int r = 0; r < NX; ++r) {
      const int idx = tid + 256 * r, pix = idx / (CC / 4), q = idx % (CC / 4);
      if (idx < TOTX) *(f32x4*)(patch + pix * PS + 4 * q) = vx[r];
    }
    __syncthreads();
    for (int c = wave * CW; c < wave * CW + CW; c += 4) {
      f32x4 xv[3][3];
#pragma unroll
      for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) xv[dy][dx] = *(const f32x4*)(patch + ((py + dy) * PWO + px + dx) * PS + c);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        // (the weights of a channel as LDS broadcasts instead made the kernel LDS-bound: 28-33 us; a lone wave per SIMD sat
        //  out every SMEM round trip: 44 us — four waves per SIMD cover them)
        const float* __restrict__ wc = w + (int64_t)(c0 + c + e) * 16 * N;      // wave-uniform address: scalar (SMEM) loads
        // output row oy = 2 my + ph takes input rows i = my - 1 + dy with filter row kh = oy + 1 - 2 i:
        //   ph = 0: (dy 1, kh 1), (dy 0, kh 3);   ph = 1: (dy 2, kh 0), (dy 1, kh 2)      (columns alike)
#pragma unroll
        for (int ph = 0; ph < 2; ++ph)
#pragma unroll
          for (int pw = 0; pw < 2; ++pw)
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
              for (int bb = 0; bb < 2; ++bb) {
                const int dy = ph == 0 ? 1 - a : 2 - a, kh = ph == 0 ? 1 + 2 * a : 2 * a;
                const int dx = pw == 0 ? 1 - bb : 2 - bb, kw = pw == 0 ? 1 + 2 * bb : 2 * bb;
                const float xs = xv[dy][dx][e];
#pragma unroll
                for (int n = 0; n < N; ++n) acc[ph * 2 + pw][n] = fmaf(xs, wc[(kh * 4 + kw) * N + n], acc[ph * 2 + pw][n]);
              }
      }
    }
  }
  // the four channel quarters meet: red[wave][class * N + n][position]
  __syncthreads();
  float* red = patch;
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int n = 0; n < N; ++n) red[(wave * 4 * N + k * N + n) * 64 + lane] = acc[k][n];
  __syncthreads();
  {
    const int ph = wave >> 1, pw = wave & 1;      // wave k finishes parity class k
    const int Ho = 2 * Hi, Wo = 2 * Wi;
    float* o = y + (((int64_t)b * Ho + 2 * my + ph) * Wo + 2 * mx + pw) * N;
#pragma unroll
    for (int n = 0; n < N; ++n) {
      float v = 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q) v += red[(q * 4 * N + wave * N + n) * 64 + lane];      // fixed order: deterministic
      o[n] = vf_act_apply(v + (bias ? bias[n] : 0.f), act, slope);
    }
  }
}

}  // namespace

// transposed 4x4 stride-2 pad-1 pass onto N = 3 channels: x [B][Hi][Wi][C] -> y [B][2Hi][2Wi][N], w physical [C][16][N] (a
// full-conv weight [Cin][kH][kW][Cout], or a conv weight [Cout][kH][kW][Cin] read for its data-gradient).  Returns -1 if the
// shape is not this kernel's (the caller keeps its GEMM + col2im path), 0 when launched, > 0 on a launch error.
int vf_internal_deconv_thin_out(vf_ctx* ctx, const float* x, const float* w, const float* bias, float* y, int B, int Hi, int Wi, int C,
                                int N, int act, float slope) {
  static const bool off = getenv("VF_NO_THIN_OUT") != nullptr;
  if (off || ctx->mfma_bf16 == 1) return -1;      // (the bf16-operand mode rounds its operands: the GEMM kernels' business)
  if (N != 3 || C % 64 != 0 || Hi % 8 != 0 || Wi % 8 != 0 || (((uintptr_t)x) & 15) != 0 || (((uintptr_t)w) & 15) != 0) return -1;
  const int tiles_x = Wi / 8, tiles_y = Hi / 8;
  VfProf prof(ctx, "deconv_thin_out", 2.0 * (double)B * Hi * Wi * C * 16 * N, 0.0);
  hipLaunchKernelGGL((k_deconv_thin_out<3>), dim3((unsigned)(B * tiles_y * tiles_x)), dim3(256), 0, ctx->stream, x, w, bias, y, Hi, Wi,
                     C, tiles_x, tiles_y, act, slope);
  VF_LAUNCH_CHECK();
  return 0;
}

// conv forward with 3 input channels, 4x4 stride 2 pad 1; act in {none, LeakyReLU, ReLU}.  Returns -1 if the shape is not this
// kernel's (the caller keeps the implicit-GEMM path), 0 when launched, > 0 on a launch error.
int vf_internal_conv_thin_fwd(vf_ctx* ctx, const float* x, const float* w, const float* bias, float* y, void* y_planes, int B, int H,
                              int W, int Cin, int Cout, int act, float slope) {
  if (ctx->mfma_bf16 == 1) return -1;       // (the bf16-operand mode rounds its operands: that is the GEMM kernels' business)
  if (Cin != 3 || Cout % 64 != 0 || H % (2 * TP) != 0 || W % (2 * TP) != 0) return -1;
  if (!(act == VF_ACT_NONE || act == VF_ACT_LRELU || act == VF_ACT_RELU)) return -1;
  if ((((uintptr_t)y) & 7) != 0) return -1;
  const float neg = act == VF_ACT_LRELU ? slope : (act == VF_ACT_RELU ? 0.f : 1.f);
  const int tiles_x = W / (2 * TP), tiles_y = H / (2 * TP);
  const int64_t out = (int64_t)B * (H / 2) * (W / 2) * Cout;
  VfProf prof(ctx, y_planes ? "conv_thin_in_planes" : "conv_thin_in", 2.0 * (double)out * 16 * Cin, 0.0);
  unsigned* bits = ctx->act_bits_out;      // one-shot (vf_net.hip): also leave the sign bits of the activated output
  ctx->act_bits_out = nullptr;
  ctx->act_bits_written = bits != nullptr;
  hipLaunchKernelGGL((k_conv_thin_in<3>), dim3((unsigned)(B * tiles_y), (unsigned)(Cout / 64)), dim3(256), 0, ctx->stream, x, w,
                     bias, y, (unsigned short*)y_planes, out, H, W, Cout, tiles_x, tiles_y, neg, bits);
  VF_LAUNCH_CHECK();
  return 0;
}

extern "C" int vf_conv2d_fwd(vf_ctx* ctx, const float* x, const float* w, const float* bias, float* y, int B, int H, int W, int Cin,
                             int Cout, int k, int stride, int pad, int act, float slope);
extern "C" int vf_planes_split(vf_ctx* ctx, const float* x, void* planes, int64_t n);

// vf_conv2d_fwd that also leaves the three bf16 planes of y (for a planes-fed consumer): in the epilogue where the kernel
// can (the thin-input layers above), else by a pass over y.
VF_API int vf_conv2d_fwd_planes(vf_ctx* ctx, const float* x, const float* w, const float* bias, float* y, void* y_planes, int B,
                                int H, int W, int Cin, int Cout, int k, int stride, int pad, int act, float slope) {
  VF_REQUIRE(y_planes != nullptr, "vf_conv2d_fwd_planes: y_planes is NULL (use vf_conv2d_fwd)");
  if (k == 4 && stride == 2 && pad == 1) {
    const int rc = vf_internal_conv_thin_fwd(ctx, x, w, bias, y, y_planes, B, H, W, Cin, Cout, act, slope);
    if (rc >= 0) return rc;
  }
  if (int rc = vf_conv2d_fwd(ctx, x, w, bias, y, B, H, W, Cin, Cout, k, stride, pad, act, slope)) return rc;
  const int Ho = (H + 2 * pad - k) / stride + 1, Wo = (W + 2 * pad - k) / stride + 1;
  return vf_planes_split(ctx, y, y_planes, (int64_t)B * Ho * Wo * Cout);
}
