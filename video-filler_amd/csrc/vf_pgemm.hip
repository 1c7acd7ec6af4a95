// vf_pgemm.hip — the 4x4 stride-2 convolution passes as implicit GEMMs whose operands arrive PRE-SPLIT into bf16 planes.
//
// vf_conv.hip's mode-3 kernels split every fp32 operand element into three bf16 planes (x = hi + mid + lo, exact) on its
// way into LDS: 22 vector instructions per staged float4, in every pass that reads the tensor, next to 12 MFMAs per K
// step — the matrix pipe was 30 % busy and the kernel issue-bound.  Here the split happens ONCE, where the tensor is
// produced (BatchNorm apply / BatchNorm backward write the planes beside the fp32 tensor; weights are split once per
// update; anything else by vf_planes_split), in a pass that is HBM-bound and has the vector slots to spare.  The GEMM's K
// step is then: 6 16-byte loads, 6 ds_write_b128, 12 ds_read_b128 and 12 v_mfma_f32_32x32x16_bf16 per wave, a handful
// of address instructions — and bit for bit the same six-term product (hi*hi, hi*mid, mid*hi, hi*lo, lo*hi, mid*mid,
// smallest first, fp32 accumulation) as mode 3.
//
// Planes layout: bf16 [3][elements], plane q at q * plane_stride; activations NHWC per plane, weights row-major
// [n][16 taps][c] per plane (vf_weight_planes also writes the [c][16][n] transpose the data-gradient passes walk).
//
// Reference call sites: nn.SpatialConvolution / nn.SpatialFullConvolution updateOutput, updateGradInput
// (train.lua:89-146,183-196; THNN SpatialConvolutionMM.c, SpatialFullConvolution.c).
#include <algorithm>
#include <cstdlib>

#include "vf_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

#define VF_OOB 0x80000000u
__device__ __forceinline__ __amdgpu_buffer_rsrc_t pg_rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ int pg_xcd_remap(int h, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = h & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (h >> 3);
}

// ------------------------------------------------------------------------------------------------ plane producers
// Exact three-way split by truncation (the same arithmetic as vf_conv.hip's vf_split3): plane q = top 16 bits of the
// running residual; every residual subtraction is exact in fp32, after two steps at most 8 significant bits are left.
__device__ __forceinline__ void pg_split4(f32x4 v, u32x2 (&o)[3]) {
  float r0 = v[0], r1 = v[1], r2 = v[2], r3 = v[3];
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    const unsigned u0 = __float_as_uint(r0), u1 = __float_as_uint(r1), u2 = __float_as_uint(r2), u3 = __float_as_uint(r3);
    o[q][0] = __builtin_amdgcn_perm(u1, u0, 0x07060302u);
    o[q][1] = __builtin_amdgcn_perm(u3, u2, 0x07060302u);
    if (q < 2) {
      r0 -= __uint_as_float(u0 & 0xffff0000u);
      r1 -= __uint_as_float(u1 & 0xffff0000u);
      r2 -= __uint_as_float(u2 & 0xffff0000u);
      r3 -= __uint_as_float(u3 & 0xffff0000u);
    }
  }
}
// The bf16-operand mode (vf_ctx_set_mfma_mode 1; BASELINE configs[4]'s "bf16"): ONE plane, the operand rounded to nearest-even —
// bit for bit the rounding vf_conv.hip's BF = 1 kernels apply inside the GEMM ((u + 0x7FFF + lsb) >> 16), done once here.
// 2 bytes per element instead of 6, one MFMA per product instead of six.
__device__ __forceinline__ unsigned pg_rne16(float v) {
  const unsigned u = __float_as_uint(v);
  return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}
__device__ __forceinline__ u32x2 pg_round4(f32x4 v) {
  u32x2 o;
  o[0] = pg_rne16(v[0]) | (pg_rne16(v[1]) << 16);
  o[1] = pg_rne16(v[2]) | (pg_rne16(v[3]) << 16);
  return o;
}
template <int NPL>
__global__ __launch_bounds__(256) void k_planes_split(const float* __restrict__ x, __bf16* __restrict__ planes, int64_t n4,
                                                      int64_t pstride) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    if constexpr (NPL == 1) {
      *(u32x2*)(planes + 4 * i) = pg_round4(((const f32x4*)x)[i]);
    } else {
      u32x2 o[3];
      pg_split4(((const f32x4*)x)[i], o);
#pragma unroll
      for (int q = 0; q < 3; ++q) *(u32x2*)(planes + q * pstride + 4 * i) = o[q];
    }
  }
}
// weights: w physical [d0][16][d1] fp32 -> native planes [3][d0][16][d1] and transposed planes [3][d1][16][d0].
// One block per (d0 tile of 32, tap, d1 tile of 32): the transpose goes through LDS so that both images are written in
// whole 64-byte rows.
__global__ __launch_bounds__(256) void k_weight_planes(const float* __restrict__ w, __bf16* __restrict__ nat, __bf16* __restrict__ tr,
                                                       int d0, int d1, int64_t pstride, int npl) {
  __shared__ float tile[32][33];
  const int t0 = blockIdx.x * 32, tap = blockIdx.y, u0 = blockIdx.z * 32;      // d0 tile, tap, d1 tile
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int j = ty; j < 32; j += 8) {
    const int a = t0 + j, b = u0 + tx;
    float v = 0.f;
    if (a < d0 && b < d1) v = w[((int64_t)a * 16 + tap) * d1 + b];
    tile[j][tx] = v;
    if (a < d0 && b < d1) {
      float r = v;
      if (npl == 1) {
        ((unsigned short*)nat)[((int64_t)a * 16 + tap) * d1 + b] = (unsigned short)pg_rne16(r);
      } else {
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          const unsigned u = __float_as_uint(r) & 0xffff0000u;
          ((unsigned short*)nat)[q * pstride + ((int64_t)a * 16 + tap) * d1 + b] = (unsigned short)(u >> 16);
          r -= __uint_as_float(u);
        }
      }
    }
  }
  if (!tr) return;
  __syncthreads();
  for (int j = ty; j < 32; j += 8) {
    const int b = u0 + j, a = t0 + tx;
    if (a < d0 && b < d1) {
      float r = tile[tx][j];
      if (npl == 1) {
        ((unsigned short*)tr)[((int64_t)b * 16 + tap) * d0 + a] = (unsigned short)pg_rne16(r);
      } else {
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          const unsigned u = __float_as_uint(r) & 0xffff0000u;
          ((unsigned short*)tr)[q * pstride + ((int64_t)b * 16 + tap) * d0 + a] = (unsigned short)(u >> 16);
          r -= __uint_as_float(u);
        }
      }
    }
  }
}

// every weight of a net in ONE launch (the nets refresh their weight planes once per parameter update: three launches per
// training iteration instead of one per layer).  Descriptor table on the device, mirrored by backend.WPLANES_DESC.
struct VfWpDesc {
  const float* w;        // physical [d0][16][d1]
  void* nat;             // planes [3][d0][16][d1]
  void* tr;              // planes [3][d1][16][d0]
  int d0, d1;
  int gx, gz;            // 32-wide tiles over d0 and d1
  int blk_off;           // first block of this weight
  int pad;
};
static_assert(sizeof(VfWpDesc) == 48, "descriptor layout is shared with the host mirror");
__global__ __launch_bounds__(256) void k_weight_planes_multi(const VfWpDesc* __restrict__ d, int n, int npl) {
  int l = 0;
  while (l + 1 < n && (int)blockIdx.x >= d[l + 1].blk_off) ++l;
  const VfWpDesc L = d[l];
  const int local = (int)blockIdx.x - L.blk_off;
  const int bx = local % L.gx, tap = (local / L.gx) % 16, bz = local / (L.gx * 16);
  const int d0 = L.d0, d1 = L.d1;
  const int64_t pstride = (int64_t)d0 * 16 * d1;
  // one 32 x 32 tile of (d0, d1) at one tap: split on the way into LDS, then both images leave in 8-byte pieces (four
  // consecutive d1 for the native rows, four consecutive d0 for the transposed ones)
  __shared__ unsigned short t[3][32][36];
  const int t0 = bx * 32, u0 = bz * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int j = ty; j < 32; j += 8) {
    const int a = t0 + j, b = u0 + tx;
    float r = (a < d0 && b < d1) ? L.w[((int64_t)a * 16 + tap) * d1 + b] : 0.f;
    if (npl == 1) {
      t[0][j][tx] = (unsigned short)pg_rne16(r);
    } else {
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const unsigned u = __float_as_uint(r) & 0xffff0000u;
        t[q][j][tx] = (unsigned short)(u >> 16);
        r -= __uint_as_float(u);
      }
    }
  }
  __syncthreads();
  const int row = threadIdx.x >> 3, c4 = (threadIdx.x & 7) * 4;
  const bool whole = (d0 % 4 == 0) && (d1 % 4 == 0);
  {      // native: row = d0 index, c4 = d1 offset
    const int a = t0 + row, b = u0 + c4;
    if (a < d0 && b < d1) {
      for (int q = 0; q < npl; ++q) {
        unsigned short* dst = (unsigned short*)L.nat + q * pstride + ((int64_t)a * 16 + tap) * d1 + b;
        if (whole) {
          u32x2 o;
          o[0] = t[q][row][c4] | ((unsigned)t[q][row][c4 + 1] << 16);
          o[1] = t[q][row][c4 + 2] | ((unsigned)t[q][row][c4 + 3] << 16);
          *(u32x2*)dst = o;
        } else {
          for (int e = 0; e < 4 && b + e < d1; ++e) dst[e] = t[q][row][c4 + e];
        }
      }
    }
  }
  {      // transposed: row = d1 index, c4 = d0 offset
    const int b = u0 + row, a = t0 + c4;
    if (a < d0 && b < d1) {
      for (int q = 0; q < npl; ++q) {
        unsigned short* dst = (unsigned short*)L.tr + q * pstride + ((int64_t)b * 16 + tap) * d0 + a;
        if (whole) {
          u32x2 o;
          o[0] = t[q][c4][row] | ((unsigned)t[q][c4 + 1][row] << 16);
          o[1] = t[q][c4 + 2][row] | ((unsigned)t[q][c4 + 3][row] << 16);
          *(u32x2*)dst = o;
        } else {
          for (int e = 0; e < 4 && a + e < d0; ++e) dst[e] = t[q][c4 + e][row];
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ the GEMM
struct PGemm {
  const void* A;         // activation planes [3][pixels][C] bf16
  const void* W;         // weight planes [3][N][16][C] bf16
  unsigned a_bytes, w_bytes;     // extent of the three planes together (buffer descriptors)
  unsigned a_ps, w_ps;           // plane strides in bytes
  const float* bias;
  float* Y;
  float* slab;
  int64_t out_elems;
  int M, lgMh, lgMw;     // GEMM rows = B << (lgMh + lgMw), decoded as (b, my, mx)
  int Hi, Wi, C, N;
  int sy, oy0, sx, ox0;  // window origin (iy0, ix0) = (my*sy + oy0 (+ph), mx*sx + ox0 (+pw)); taps step +1 from there
  int lgTW;              // taps per window row: 4 (lgTW 2; 16 taps) or 2 (lgTW 1; the 4 taps of one output-parity class)
  int kh0, khs, kw0, kws;      // filter tap of window tap (th, tw): (kh0 + th*khs, kw0 + tw*kws); parity: kh0 = 3 - ph, khs = -2
  int parity;
  int outH, outW, osy, ooy0, osx, oox0;
  int gm, gn, gz, ksplit, nchunks;   // nchunks = C / (channels per K step)
  int act;
  float slope;
  const float* dmask;
  const unsigned* dbits;   // the same mask as sign bits (vf_common.h act_bits_out layout), or NULL: read by the epilogue prefetch
  int dact;
  float dslope;
  long long* stamps;     // timing experiments only (VF_PG_STAMPS=<file>, k_pconv_patch_g<., true>): shader-clock stamps per wave and step
  int dbg;               // timing experiments only (VF_PG_DBG; wrong results): 1 = no operand loads after the first step,
                         // 2 = no LDS writes after the first step, 4 = no MFMAs, 8 = no output stores, 32 = no first stage either (k_pconv_patch_g)
  VfBnSt st;
};

// per-channel partial sums of one block's output tile (see vf_conv.hip vf_bn_tile_partials: same layout, same order)
template <int NT, int WAVES_M, int BN>
__device__ __forceinline__ void pg_bn_tile_partials(const VfBnSt& st, float (&s1)[NT], float (&s2)[NT], float* red, int wave_m, int wn,
                                                    int lane, int tid, int n0, int N, int bx, int pz) {
  const int lr = lane & 31;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    s1[nt] += __shfl_xor(s1[nt], 32, 64);
    s2[nt] += __shfl_xor(s2[nt], 32, 64);
    if (lane < 32) {
      red[(wave_m * 2 + 0) * BN + wn + nt * 32 + lr] = s1[nt];
      red[(wave_m * 2 + 1) * BN + wn + nt * 32 + lr] = s2[nt];
    }
  }
  __syncthreads();
  if (tid < BN && n0 + tid < N) {
    double a = 0, b = 0;
#pragma unroll
    for (int w = 0; w < WAVES_M; ++w) {
      a += (double)red[(w * 2 + 0) * BN + tid];
      b += (double)red[(w * 2 + 1) * BN + tid];
    }
    const int g = bx / st.tiles_per_group, local = bx - g * st.tiles_per_group;
    double* o = st.part + ((int64_t)(g * st.rows_per_group + local * st.zpar + pz) * 2) * N;
    o[n0 + tid] = a;
    o[N + n0 + tid] = b;
  }
}

// ---- epilogue shared by the GEMM kernels below: C/D layout col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
// The store loop is instantiated per (derivative mask present, statistics mode) and the (leaky) ReLU is a select of the
// multiplier, so the 16 * MT * NT element bodies are straight-line code (with the activation as a per-element switch and
// the statistics mode as per-element branches the epilogue was a quarter of vf_conv.hip's igemm kernels).
// `red`: LDS the tile loop no longer needs (2 * WAVES_M * BN floats), for the statistics partials.
// pre_d / pre_x (PRE, MT = 1; [column tile][16]): the 16 derivative-mask values / BatchNorm inputs of this lane's accumulator
// elements, fetched by the caller BEFORE its K loop — read here, at the end of the tile, they were a serialised memory phase
// that nothing overlapped (E2's data-gradient: 62 us without the mask, 78 with it)
// pix_of(row): the output pixel (index into [pixels][N]) of row `row` of the block tile, or -1 for a row past the end of the GEMM —
// the row -> pixel map is the caller's (k_pconv / k_pconv_dma: GEMM row m0 + row decoded as (b, my, mx); k_pconv_patch_g: an 8 x 16
// spatial tile), the arithmetic on the element is this function's alone.
template <int MT, int NT, int WAVES_M, int BN, bool PRE = false, class PixOf>
__device__ __forceinline__ void pg_epilogue_at(const PGemm& p, f32x16 (&acc)[MT][NT], float* red, int n0, int wm, int wn, int lane,
                                               int tid, int wave_m, int bx, int ks, int ph, int pw, bool tile_ok,
                                               const float (&pre_d)[NT][16], const float (&pre_x)[NT][16], PixOf pix_of) {
  const int lr = lane & 31, lh = lane >> 5;
  float* out = p.ksplit > 1 ? p.slab + (int64_t)ks * p.out_elems : p.Y;
  const bool fin = p.ksplit == 1;
  const int stm = fin ? p.st.mode : 0;
  float bv[NT], sv[NT], st1[NT], st2[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int n = n0 + wn + nt * 32 + lr;
    bv[nt] = (fin && p.bias && n < p.N) ? p.bias[n] : 0.f;
    sv[nt] = (stm && n < p.N) ? p.st.vec[(stm == 2 ? (bx / p.st.tiles_per_group) * p.N : 0) + n] : 0.f;
    st1[nt] = 0.f;
    st2[nt] = 0.f;
  }
  // y = v * (v > 0 ? 1 : neg): neg = 1 (none), slope (LeakyReLU), 0 (ReLU) — NaN-propagating like `v > 0 ? v : v * slope`
  const float neg = !fin ? 1.f : (p.act == VF_ACT_LRELU ? p.slope : (p.act == VF_ACT_RELU ? 0.f : 1.f));
  const float dneg = p.dact == VF_ACT_LRELU ? p.dslope : (p.dact == VF_ACT_RELU ? 0.f : 1.f);
  const bool smooth_act = fin && (p.act == VF_ACT_TANH || p.act == VF_ACT_SIGMOID);
  auto store_all = [&](auto HAS_DMASK, auto STM) {
    constexpr bool has_dmask = decltype(HAS_DMASK)::value != 0;
    constexpr int sm = decltype(STM)::value;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        const int64_t pix = pix_of(row);
        if (pix < 0) continue;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const int n = n0 + wn + nt * 32 + lr;
          if (n < p.N) {
            float v = acc[mt][nt][r] + bv[nt];
            v = v * (v > 0.f ? 1.f : neg);
            if constexpr (has_dmask) v = v * ((PRE ? pre_d[nt][r] : p.dmask[pix * p.N + n]) > 0.f ? 1.f : dneg);
            if constexpr (sm == 1) {
              const float d = v - sv[nt];
              st1[nt] += d;
              st2[nt] += d * d;
            } else if constexpr (sm == 2) {
              st1[nt] += v;
              st2[nt] += v * ((PRE ? pre_x[nt][r] : p.st.x[pix * p.N + n]) - sv[nt]);
            }
            if (!(p.dbg & 8)) out[pix * p.N + n] = v;
          }
        }
      }
    }
  };
  if (smooth_act) {                 // Tanh / Sigmoid behind a convolution: the image-side layers, not this kernel's business —
#pragma unroll                      // kept for completeness, generic form
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        const int64_t pix = pix_of(row);
        if (pix < 0) continue;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const int n = n0 + wn + nt * 32 + lr;
          if (n < p.N) out[pix * p.N + n] = vf_act_apply(acc[mt][nt][r] + bv[nt], p.act, p.slope);
        }
      }
  } else if (fin && p.dmask) {
    if (stm == 2) store_all(VfIntC<1>{}, VfIntC<2>{});
    else store_all(VfIntC<1>{}, VfIntC<0>{});
  } else {
    if (stm == 1) store_all(VfIntC<0>{}, VfIntC<1>{});
    else if (stm == 2) store_all(VfIntC<0>{}, VfIntC<2>{});
    else store_all(VfIntC<0>{}, VfIntC<0>{});
  }
  if (stm)
    pg_bn_tile_partials<NT, WAVES_M, BN>(p.st, st1, st2, red, wave_m, wn, lane, tid, n0, tile_ok ? p.N : 0, bx,
                                         p.parity ? ((ph << 1) | pw) : 0);
}

// the GEMM-row form: row `row` of the tile is GEMM row m0 + row = (b, my, mx) on the Mh x Mw grid, output pixel (my osy + ooy0, mx osx + oox0)
template <int MT, int NT, int WAVES_M, int BN, bool PRE = false>
__device__ __forceinline__ void pg_epilogue(const PGemm& p, f32x16 (&acc)[MT][NT], float* red, int m0, int n0, int wm, int wn, int lane,
                                            int tid, int wave_m, int bx, int ks, int ooy0, int oox0, int ph, int pw, bool tile_ok,
                                            const float (&pre_d)[16], const float (&pre_x)[16]) {
  const int Mw = 1 << p.lgMw, Mh = 1 << p.lgMh;
  static_assert(!PRE || NT == 1, "the prefetched epilogue operands of the GEMM-row form are one column tile's");
  typedef const float (&pre_t)[NT][16];
  pg_epilogue_at<MT, NT, WAVES_M, BN, PRE>(p, acc, red, n0, wm, wn, lane, tid, wave_m, bx, ks, ph, pw, tile_ok,
                                           reinterpret_cast<pre_t>(pre_d), reinterpret_cast<pre_t>(pre_x),
                                           [&](int row) -> int64_t {
                                             const int m = m0 + row;
                                             if (m >= p.M) return -1;
                                             const int mx = m & (Mw - 1), my = (m >> p.lgMw) & (Mh - 1), b = m >> (p.lgMw + p.lgMh);
                                             return ((int64_t)b * p.outH + (my * p.osy + ooy0)) * p.outW + (mx * p.osx + oox0);
                                           });
}

// BM x BN block tile, 4 waves of WM x WN (32x32 MFMA tiles), one K step = ONE window tap x CH channels, K walked channel
// chunk outer / tap inner (the taps of a chunk touch the same input window, so the re-reads stay in L1/L2).
// CH = 64 wherever C % 64 == 0: a row segment is then a whole 128-byte line per plane.  With 32-channel steps the
// other half of every fetched line was wanted 16 taps later, long after a 32 KB L1 had dropped it: the loads ran at 26
// bytes per clock and CU (an L2 -> L1 fill stream of twice the useful bytes) and the whole kernel at their pace — with
// the MFMAs ablated it took 97 % of its time, with the loads ablated 66 %.
// LDS: 3 planes x (A, B) tiles, rows of CH bf16 with the 16-byte k-octets XOR-swizzled ((row >> 2) & 3 for 64-byte rows,
// (row >> 1) & 7 for 128-byte rows: conflict-free ds_read_b128 fragments and ds_write_b128 pieces).
//   CH = 32: two buffers; per step the loads of step s+1 are issued, the MFMAs of step s run, the pieces go to the other
//            buffer, one barrier.
//   CH = 64: one buffer (48 KB for 64x64: three blocks per CU); loads of step s+1 issued, MFMAs of step s, barrier,
//            pieces to LDS, barrier — half the steps, the same number of barriers per K.
// PAIR: a workgroup of 512 threads = TWO such tiles (threads 0-255 and 256-511, each with its own LDS tile buffers) whose
// steps run in ANTI-PHASE: between two consecutive workgroup barriers one half issues its loads and runs its MFMAs while
// the other waits for its loads and writes them to LDS.  Two independent 256-thread blocks on one CU drift into lockstep
// (both in the load wait, then both in the MFMAs, then both in the LDS writes — a K step of the pair took the SUM of the
// phases: matrix pipe 25 %, TA 33 %, LDS 25 % busy, waves a third of their time in s_waitcnt); the pairing pins the
// complementary schedule.
template <int BM, int BN, int WM, int WN, int NTAPS, int CH, bool PAIR>
__global__ __launch_bounds__(PAIR ? 512 : 256) void k_pconv(const PGemm p) {
  constexpr int MT = WM / 32, NT = WN / 32, WAVES_N = BN / WN;
  static_assert((BM / WM) * WAVES_N == 4, "4 waves per block");
  static_assert(CH == 32 || CH == 64, "K step of 32 or 64 channels");
  constexpr int OC = CH / 8;                                   // 16-byte octets per row
  constexpr int A_CH = BM * OC / 256, B_CH = BN * OC / 256;    // 16-byte pieces per thread and plane
  static_assert(A_CH >= 1 && B_CH >= 1, "tiles of at least 64 rows");
  constexpr int AH_SZ = BM * CH, BH_SZ = BN * CH, PL_SZ = AH_SZ + BH_SZ;      // bf16 elements
  constexpr int BUF_SZ = 3 * PL_SZ;
  constexpr int NBUF = CH == 32 ? 2 : 1;
  __shared__ __attribute__((aligned(16))) __bf16 smem_all[(PAIR ? 2 : 1) * NBUF * BUF_SZ];
  const int sub = PAIR ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8)) : 0;      // which tile of the pair (wave-uniform)
  __bf16* smem = smem_all + sub * (NBUF * BUF_SZ);
  auto sw_off = [](int row, int octet) {
    if constexpr (CH == 32) return row * 32 + ((octet ^ ((row >> 2) & 3)) << 3);
    else return row * 64 + ((octet ^ ((row >> 1) & 7)) << 3);
  };

  const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;
  const int wm = (wave / WAVES_N) * WM, wn = (wave % WAVES_N) * WN;
  const int ntiles = p.gm * p.gn * p.gz;
  int lid = PAIR ? 2 * pg_xcd_remap(blockIdx.x, gridDim.x) + sub : pg_xcd_remap(blockIdx.x, ntiles);
  const bool tile_ok = lid < ntiles;       // (an odd tile count leaves the last pair's second half idle: it keeps the barriers)
  if (!tile_ok) lid = 0;
  int ph = 0, pw = 0;
  if (p.parity) {
    ph = (lid >> 1) & 1;
    pw = lid & 1;
    lid >>= 2;
  }
  const int bx = lid % p.gm, byz = lid / p.gm;
  const int m0 = tile_ok ? bx * BM : p.M, n0 = (byz % p.gn) * BN;      // idle half: every row out of range
  const int ks = byz / p.gn;
  const int cps = (p.nchunks + p.ksplit - 1) / p.ksplit;          // channel chunks per split
  const int ch0 = ks * cps, ch1 = min(p.nchunks, ch0 + cps);
  const int oy0 = p.oy0 + ph, ox0 = p.ox0 + pw;
  const int kh0 = p.parity ? (3 - ph) : p.kh0, kw0 = p.parity ? (3 - pw) : p.kw0;
  const int ooy0 = p.ooy0 + ph, oox0 = p.oox0 + pw;
  const int Mw = 1 << p.lgMw, Mh = 1 << p.lgMh;
  constexpr int TWm = NTAPS == 16 ? 3 : 1, lgTW = NTAPS == 16 ? 2 : 1;

  // ---- per-thread operand pieces: (row, octet) fixed for the whole loop
  unsigned a_byte[A_CH], a_mask[A_CH], w_byte[B_CH];
  int a_lds[A_CH], b_lds[B_CH];
#pragma unroll
  for (int i = 0; i < A_CH; ++i) {
    const int id = tid + 256 * i, row = id / OC, oct = id % OC;
    const int m = m0 + row;
    const int mx = m & (Mw - 1), my = (m >> p.lgMw) & (Mh - 1), b = m >> (p.lgMw + p.lgMh);
    const int iy0 = my * p.sy + oy0, ix0 = mx * p.sx + ox0;
    a_byte[i] = 2u * (unsigned)(((b * p.Hi + iy0) * p.Wi + ix0) * p.C + 8 * oct);      // (wraps for border rows; masked)
    unsigned ym = 0, xm = 0;
#pragma unroll
    for (int t = 0; t <= TWm; ++t) {
      if ((unsigned)(iy0 + t) < (unsigned)p.Hi) ym |= 1u << t;
      if ((unsigned)(ix0 + t) < (unsigned)p.Wi) xm |= 1u << t;
    }
    unsigned mk = 0;
#pragma unroll
    for (int t = 0; t <= TWm; ++t)
      if ((ym >> t) & 1u) mk |= xm << (t << lgTW);
    a_mask[i] = m < p.M ? mk : 0u;
    a_lds[i] = sw_off(row, oct);
  }
#pragma unroll
  for (int i = 0; i < B_CH; ++i) {
    const int id = tid + 256 * i, row = id / OC, oct = id % OC;
    const int n = n0 + row;
    w_byte[i] = n < p.N ? 2u * (unsigned)(n * 16 * p.C + 8 * oct) : VF_OOB;
    b_lds[i] = AH_SZ + sw_off(row, oct);
  }
  const __amdgpu_buffer_rsrc_t rsA = pg_rsrc(p.A, p.a_bytes), rsW = pg_rsrc(p.W, p.w_bytes);
  // wave-uniform byte offsets of window tap (th, tw): th * row + tw * col (+ the filter tap's base for the weights); th and
  // tw are compile-time constants at every use, so each offset is two scalar multiply-adds (a table of 2 x 16 offsets held
  // in SGPRs spilled: 106 SGPRs, 36-68 bytes of scratch per lane)
  int rowA = 2 * p.Wi * p.C, colA = 2 * p.C;
  int w0 = 2 * (kh0 * 4 + kw0) * p.C, rowW = 8 * p.khs * p.C, colW = 2 * p.kws * p.C;

  u32x4 ra[A_CH][3], rb[B_CH][3];
  auto load_step = [&](int ch, auto TAP, bool live) {
    // step index -> window tap.  The 16 taps of the gather form are walked one input-parity class after the other —
    // (th, tw) = (g>>1 + 2*(j>>1), g&1 + 2*(j&1)) for step 4g + j: the four taps of a class read the SAME quarter of the
    // input pixels (each for a different output pixel), so a line is re-read one step after it was fetched instead of
    // two (x shifts) or eight (y shifts) steps later, by which time the 64 tiles that share an XCD's 4 MB L2 had pushed it
    // out: TCC hit rate 64 % and 141 MB of fabric reads per launch for a 25 MB operand in row-major tap order
    constexpr int s_ = decltype(TAP)::value;
    constexpr int th = NTAPS == 16 ? (((s_ >> 3) & 1) + 2 * ((s_ >> 1) & 1)) : (s_ >> 1);
    constexpr int tw = NTAPS == 16 ? (((s_ >> 2) & 1) + 2 * (s_ & 1)) : (s_ & 1);
    constexpr int t = (th << lgTW) | tw;                              // bit of the validity mask
    const unsigned cb = (unsigned)(2 * CH) * (unsigned)ch;            // CH channels * 2 bytes
    const unsigned tA = (unsigned)(th * rowA + tw * colA) + cb, tW = (unsigned)(w0 + th * rowW + tw * colW) + cb;
#pragma unroll
    for (int i = 0; i < A_CH; ++i) {
      const bool ok = live && ((a_mask[i] >> t) & 1u);
      const unsigned off = ok ? a_byte[i] + tA : VF_OOB;
#pragma unroll
      for (int q = 0; q < 3; ++q) ra[i][q] = __builtin_amdgcn_raw_buffer_load_b128(rsA, off, q * p.a_ps, 0);
    }
#pragma unroll
    for (int i = 0; i < B_CH; ++i) {
      const unsigned off = live ? w_byte[i] : VF_OOB;
#pragma unroll
      for (int q = 0; q < 3; ++q) rb[i][q] = __builtin_amdgcn_raw_buffer_load_b128(rsW, off, q * p.w_ps + tW, 0);
    }
  };
  auto store_step = [&](int buf) {
    __bf16* base = smem + buf * BUF_SZ;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
#pragma unroll
      for (int i = 0; i < A_CH; ++i) *(u32x4*)(base + q * PL_SZ + a_lds[i]) = ra[i][q];
#pragma unroll
      for (int i = 0; i < B_CH; ++i) *(u32x4*)(base + q * PL_SZ + b_lds[i]) = rb[i][q];
    }
  };

  f32x16 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;
  const int lr = lane & 31, lh = lane >> 5;

  // fragment reads run one 16-deep group AHEAD of the MFMAs that consume them (two fragment sets): left to itself the
  // compiler issues "reads of group g, wait, six MFMAs of group g, reads of group g+1, wait ..." and every group starts
  // with the LDS latency exposed
  auto compute_step = [&](int buf) {
    const __bf16* base = smem + buf * BUF_SZ;
    constexpr int G = CH / 16;
    bf16x8 a[2][3][MT], b[2][3][NT];
    auto read_frag = [&](int g, int set) {
#pragma unroll
      for (int q = 0; q < 3; ++q) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) a[set][q][mt] = *(const bf16x8*)(base + q * PL_SZ + sw_off(wm + mt * 32 + lr, 2 * g + lh));
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) b[set][q][nt] = *(const bf16x8*)(base + q * PL_SZ + AH_SZ + sw_off(wn + nt * 32 + lr, 2 * g + lh));
      }
    };
    read_frag(0, 0);
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const int cs = g & 1;
      if (g + 1 < G) read_frag(g + 1, cs ^ 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {      // smallest terms first (the order of vf_conv.hip's mode 3)
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[cs][1][mt], b[cs][1][nt], acc[mt][nt], 0, 0, 0);
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[cs][0][mt], b[cs][2][nt], acc[mt][nt], 0, 0, 0);
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[cs][2][mt], b[cs][0][nt], acc[mt][nt], 0, 0, 0);
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[cs][0][mt], b[cs][1][nt], acc[mt][nt], 0, 0, 0);
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[cs][1][mt], b[cs][0][nt], acc[mt][nt], 0, 0, 0);
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[cs][0][mt], b[cs][0][nt], acc[mt][nt], 0, 0, 0);
        }
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  // ---- main loop: NTAPS steps per channel chunk, fully unrolled over the taps
  if (ch0 < ch1) {
    load_step(ch0, VfIntC<0>{}, true);
    store_step(0);
  }
  __syncthreads();
  static_assert(NTAPS % 2 == 0, "the two LDS buffers alternate with the tap parity");
  static_assert(!PAIR || NBUF == 1, "the paired schedule is built on the single-buffered (64-channel) step");
  if (PAIR && sub == 1) __syncthreads();              // the second half runs one phase behind the first
  for (int ch = ch0; ch < ch1; ++ch) {
    const bool more = ch + 1 < ch1;
    // (the five multipliers pass through an empty asm each iteration: left alone, the compiler hoists all 2 x NTAPS tap
    //  offsets out of the loop into SGPRs it does not have)
    asm volatile("" : "+s"(rowA), "+s"(colA), "+s"(w0), "+s"(rowW), "+s"(colW));
    vf_static_for<NTAPS>([&](auto T) {
      constexpr int t = decltype(T)::value;
      constexpr int buf = NBUF == 2 ? (t & 1) : 0;
      if (!(p.dbg & 1)) {
        if constexpr (t + 1 < NTAPS)
          load_step(ch, VfIntC<t + 1>{}, true);
        else
          load_step(ch + 1, VfIntC<0>{}, more);
      }
      if (!(p.dbg & 4)) compute_step(buf);
      if constexpr (NBUF == 1) __syncthreads();       // everyone has read the tile before it is overwritten
      if (!(p.dbg & 2)) store_step(NBUF == 2 ? (buf ^ 1) : 0);
      __syncthreads();
    });
  }
  if (PAIR && sub == 0) __syncthreads();              // (the barrier that pairs with the second half's last one)

  const float none[16] = {};
  pg_epilogue<MT, NT, BM / WM, BN>(p, acc, (float*)smem, m0, n0, wm, wn, lane, tid, wave / WAVES_N, bx, ks, ooy0, oox0, ph, pw, tile_ok,
                                   none, none);
}

// ------------------------------------------------------------------------------------------------ the same GEMM fed by LDS-DMA
// k_pconv stages its operands global -> registers -> ds_write_b128 -> LDS.  Ablated, that kernel without ANY global load
// still took 30 of its 36 us on the 4.3 GFLOP layers (10 us of MFMAs at the pipe's rate): 48 KB of ds_write_b128 per block
// and K step move at ~80 bytes per clock and CU, with a barrier on either side.  Here the tiles go global -> LDS directly
// (`buffer_load_dwordx4 ... lds`: 1 KB = 8 rows x 128 bytes per wave-instruction, no VGPRs, no ds_write, out-of-range
// lanes — padding taps — write zeros: scripts/probe/lds_dma_oob.hip), two LDS stages, ONE barrier per K step:
//     wait for this wave's DMAs of stage s | barrier | issue the DMAs of stage s+1 into the other buffer | MFMAs of stage s
// BM x BN x 64-channel stages, (BM/32) x (BN/32) waves of one 32x32 MFMA tile each (128x64: 8 waves, 144 KB of LDS, one
// block per CU, two waves per SIMD).  The LDS image is lane-linear per DMA instruction, so the XOR swizzle of the k-octets
// sits in the per-lane SOURCE address.  The DMA is issued from inline asm with its own s_waitcnt: to the compiler it is
// no LDS store, so it puts no vmcnt(0) in front of the fragment reads.
__device__ __forceinline__ void pg_dma16(unsigned lds_byte, unsigned voffset, __amdgpu_buffer_rsrc_t rsrc, unsigned soffset) {
  unsigned keep;
  soffset = (unsigned)__builtin_amdgcn_readfirstlane((int)soffset);      // (wave-uniform by construction: pin it to an SGPR)
  lds_byte = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_byte);
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "s"(lds_byte), "v"(voffset), "s"(rsrc), "s"(soffset)
               : "memory");
}

// NPL = planes per operand: 3 (mode 3: the exact split, six product terms) or 1 (mode 1: operands rounded to bf16 by their
// producer, ONE term — a stage is a third of the bytes and a sixth of the MFMAs; 128x64 with two stages: 48 KB, three blocks per CU)
template <int BM, int BN, int NTAPS, int NBUF = 2, int NPL = 3>
__global__ __launch_bounds__((BM / 32) * (BN / 32) * 64) void k_pconv_dma(const PGemm p) {
  constexpr int CH = 64, WAVES_N = BN / 32, WAVES_M = BM / 32, NW = WAVES_M * WAVES_N;
  constexpr int AG = BM / 8 / NW, BG = BN / 8 / NW;                 // 8-row groups (one DMA instruction per plane) per wave
  static_assert(AG * NW * 8 == BM && BG * NW * 8 == BN, "row groups must divide over the waves");
  static_assert(NPL == 1 || NPL == 3, "one rounded plane or the exact three-way split");
  constexpr int AH_SZ = BM * CH, BH_SZ = BN * CH, PL_SZ = AH_SZ + BH_SZ, BUF_SZ = NPL * PL_SZ;      // bf16 elements
  __shared__ __attribute__((aligned(1024))) __bf16 smem[NBUF * BUF_SZ];
  auto sw_off = [](int row, int octet) { return row * 64 + ((octet ^ ((row >> 1) & 7)) << 3); };

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = (wave / WAVES_N) * 32, wn = (wave % WAVES_N) * 32;
  const int ntiles = p.gm * p.gn * p.gz;
  int lid = pg_xcd_remap(blockIdx.x, ntiles);
  int ph = 0, pw = 0;
  if (p.parity) {
    ph = (lid >> 1) & 1;
    pw = lid & 1;
    lid >>= 2;
  }
  const int bx = lid % p.gm, byz = lid / p.gm;
  const int m0 = bx * BM, n0 = (byz % p.gn) * BN;
  const int ks = byz / p.gn;
  const int cps = p.nchunks / p.ksplit;          // channel chunks per split (the host makes it exact)
  const int ch0 = ks * cps, ch1 = ch0 + cps;
  const int oy0 = p.oy0 + ph, ox0 = p.ox0 + pw;
  const int kh0 = p.parity ? (3 - ph) : p.kh0, kw0 = p.parity ? (3 - pw) : p.kw0;
  const int ooy0 = p.ooy0 + ph, oox0 = p.oox0 + pw;
  const int Mw = 1 << p.lgMw, Mh = 1 << p.lgMh;
  constexpr int TWm = NTAPS == 16 ? 3 : 1, lgTW = NTAPS == 16 ? 2 : 1;

  // ---- this lane's share of every stage: row (8 * group + lane / 8), LDS slot lane % 8 = global octet slot ^ swizzle(row)
  unsigned a_byte[AG], a_mask[AG], w_byte[BG];
  unsigned a_lds[AG], b_lds[BG];          // wave-uniform LDS byte offsets of the groups inside a plane
#pragma unroll
  for (int i = 0; i < AG; ++i) {
    const int grp = wave + NW * i, row = 8 * grp + (lane >> 3);
    const int oct = (lane & 7) ^ ((row >> 1) & 7);
    const int m = m0 + row;
    const int mx = m & (Mw - 1), my = (m >> p.lgMw) & (Mh - 1), b = m >> (p.lgMw + p.lgMh);
    const int iy0 = my * p.sy + oy0, ix0 = mx * p.sx + ox0;
    a_byte[i] = 2u * (unsigned)(((b * p.Hi + iy0) * p.Wi + ix0) * p.C + 8 * oct);
    unsigned ym = 0, xm = 0;
#pragma unroll
    for (int t = 0; t <= TWm; ++t) {
      if ((unsigned)(iy0 + t) < (unsigned)p.Hi) ym |= 1u << t;
      if ((unsigned)(ix0 + t) < (unsigned)p.Wi) xm |= 1u << t;
    }
    unsigned mk = 0;
#pragma unroll
    for (int t = 0; t <= TWm; ++t)
      if ((ym >> t) & 1u) mk |= xm << (t << lgTW);
    a_mask[i] = m < p.M ? mk : 0u;
    a_lds[i] = 2u * (unsigned)(grp * 8 * 64);
  }
#pragma unroll
  for (int i = 0; i < BG; ++i) {
    const int grp = wave + NW * i, row = 8 * grp + (lane >> 3);
    const int oct = (lane & 7) ^ ((row >> 1) & 7);
    const int n = n0 + row;
    w_byte[i] = n < p.N ? 2u * (unsigned)(n * 16 * p.C + 8 * oct) : VF_OOB;
    b_lds[i] = 2u * (unsigned)(AH_SZ + grp * 8 * 64);
  }
  const __amdgpu_buffer_rsrc_t rsA = pg_rsrc(p.A, p.a_bytes), rsW = pg_rsrc(p.W, p.w_bytes);
  int rowA = 2 * p.Wi * p.C, colA = 2 * p.C;
  int w0 = 2 * (kh0 * 4 + kw0) * p.C, rowW = 8 * p.khs * p.C, colW = 2 * p.kws * p.C;
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)smem;      // LDS byte address of the buffers

  // one stage = NP pieces (a DMA instruction each: 8 rows x 128 bytes of one plane); piece J of tap TAP's stage
  constexpr int NP = NPL * (AG + BG);
  unsigned szero;                                   // an SGPR zero the compiler cannot fold into the asm's soffset operand
  asm volatile("s_mov_b32 %0, 0" : "=s"(szero));
  auto dma_piece = [&](int ch, auto TAP, auto PIECE, int buf, bool live) {
    constexpr int s_ = decltype(TAP)::value, J = decltype(PIECE)::value;
    constexpr int th = NTAPS == 16 ? (((s_ >> 3) & 1) + 2 * ((s_ >> 1) & 1)) : (s_ >> 1);
    constexpr int tw = NTAPS == 16 ? (((s_ >> 2) & 1) + 2 * (s_ & 1)) : (s_ & 1);
    constexpr int t = (th << lgTW) | tw;
    const unsigned cb = (unsigned)(2 * CH) * (unsigned)ch;
    const unsigned base = lds0 + 2u * (unsigned)(buf * BUF_SZ);
    if constexpr (J < NPL * AG) {
      constexpr int i = J / NPL, q = J % NPL;
      const unsigned tA = (unsigned)(th * rowA + tw * colA) + cb;
      const bool ok = live && ((a_mask[i] >> t) & 1u);
      pg_dma16(base + 2u * (unsigned)(q * PL_SZ) + a_lds[i], ok ? a_byte[i] + tA : VF_OOB, rsA, szero + q * p.a_ps);
    } else {
      constexpr int i = (J - NPL * AG) / NPL, q = (J - NPL * AG) % NPL;
      const unsigned tW = (unsigned)(w0 + th * rowW + tw * colW) + cb;
      pg_dma16(base + 2u * (unsigned)(q * PL_SZ) + b_lds[i], live ? w_byte[i] : VF_OOB, rsW, q * p.w_ps + tW);
    }
  };
  auto dma_step = [&](int ch, auto TAP, int buf, bool live) {
    vf_static_for<NP>([&](auto J) { dma_piece(ch, TAP, J, buf, live); });
  };

  f32x16 acc[1][1];
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[0][0][r] = 0.f;
  const int lr = lane & 31, lh = lane >> 5;
  // MFMAs of the stage in `buf` (fragment reads one k-group ahead).  (Tried and dropped: issuing the next stage's DMA pieces
  // BETWEEN these MFMAs instead of in front of them: -2 % on the gathers, -25 % on the short-K transposed passes — the issue
  // slots are not what the MFMAs wait for: DMA-only 23.4 us, MFMA-only 22.7 us, both 28.8 us, neither 7.7 us on the 4.3 GFLOP pass.)
  auto compute_step = [&](int buf) {
    const __bf16* base = smem + buf * BUF_SZ;
    bf16x8 a[2][NPL], b[2][NPL];
    auto read_frag = [&](int g, int set) {
#pragma unroll
      for (int q = 0; q < NPL; ++q) {
        a[set][q] = *(const bf16x8*)(base + q * PL_SZ + sw_off(wm + lr, 2 * g + lh));
        b[set][q] = *(const bf16x8*)(base + q * PL_SZ + AH_SZ + sw_off(wn + lr, 2 * g + lh));
      }
    };
    read_frag(0, 0);
#pragma unroll
    for (int g = 0; g < CH / 16; ++g) {
      const int cs = g & 1;
      if (g + 1 < CH / 16) read_frag(g + 1, cs ^ 1);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (NPL == 3) {
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[cs][1], b[cs][1], acc[0][0], 0, 0, 0);
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[cs][0], b[cs][2], acc[0][0], 0, 0, 0);
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[cs][2], b[cs][0], acc[0][0], 0, 0, 0);
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[cs][0], b[cs][1], acc[0][0], 0, 0, 0);
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[cs][1], b[cs][0], acc[0][0], 0, 0, 0);
      }
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[cs][0], b[cs][0], acc[0][0], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  // ---- what the epilogue reads besides the accumulators — the derivative mask of the activation below (conv -> LeakyReLU ->
  //      conv, or the BatchNorm's activation) and, for the BatchNorm-backward sums, that BatchNorm's input — fetched NOW, 16
  //      values each per lane, so that they travel beside the first stages instead of after the last MFMA
  float pre_d[16], pre_x[16];
  const bool fin_tile = p.ksplit == 1;
  const bool want_d = fin_tile && p.dmask != nullptr, want_x = fin_tile && p.st.mode == 2;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    pre_d[r] = 1.f;
    pre_x[r] = 0.f;
  }
  if (want_d || want_x) {
    const int n = n0 + wn + lr;
    // (sign bits instead of the fp32 activation where the producer left them: word 2 * (n / 64) + (n & 1) of the pixel, bit
    //  (n / 2) % 32 — the 32 lanes of a half wave share two words)
    const int bw = 2 * (n >> 6) + (n & 1), bb = (n >> 1) & 31, wpp = p.N >> 5;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + wm + (r & 3) + 8 * (r >> 2) + 4 * lh;
      const int mx = m & (Mw - 1), my = (m >> p.lgMw) & (Mh - 1), b = m >> (p.lgMw + p.lgMh);
      const int64_t pix = ((int64_t)b * p.outH + (my * p.osy + ooy0)) * p.outW + (mx * p.osx + oox0);
      if (want_d) pre_d[r] = p.dbits ? (float)((p.dbits[pix * wpp + bw] >> bb) & 1u) : p.dmask[pix * p.N + n];
      if (want_x) pre_x[r] = p.st.x[pix * p.N + n];
    }
  }

  // ---- main loop
  dma_step(ch0, VfIntC<0>{}, 0, ch0 < ch1);
  static_assert(NTAPS % 2 == 0, "the two LDS stages alternate with the tap parity");
  for (int ch = ch0; ch < ch1; ++ch) {
    const bool more = ch + 1 < ch1;
    asm volatile("" : "+s"(rowA), "+s"(colA), "+s"(w0), "+s"(rowW), "+s"(colW));
    vf_static_for<NTAPS>([&](auto T) {
      constexpr int t = decltype(T)::value;
      constexpr int buf = NBUF == 2 ? (t & 1) : 0, nbuf = NBUF == 2 ? (buf ^ 1) : 0;
      auto next_stage = [&]() {
        if (!(p.dbg & 1)) {
          if constexpr (t + 1 < NTAPS)
            dma_step(ch, VfIntC<t + 1>{}, nbuf, true);
          else
            dma_step(ch + 1, VfIntC<0>{}, nbuf, more);
        }
      };
      // this wave's DMAs of the stage about to be read have landed; after the barrier everybody's have, and (two stages)
      // everybody is done reading the other buffer, which the next stage's DMAs overwrite
      asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (NBUF == 2) next_stage();
      __builtin_amdgcn_sched_barrier(0);
      if (!(p.dbg & 4)) compute_step(buf);
      if constexpr (NBUF == 1) {
        // ONE stage (72 KB: two blocks per CU): the next stage can only go in once everybody has read this one — a block
        // does not overlap its own loads and MFMAs, its CU-mate does.  For the transposed passes whose tiles have four K
        // steps, where a lone block per CU spent more time starting up and draining than computing.
        asm volatile("s_barrier" ::: "memory");
        next_stage();
      }
    });
  }
  // the last stage's prefetch (dead: out-of-range zeros) must have landed before LDS is reused, and nobody may still be
  // reading fragments when the epilogue's partial sums go there
  asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
  pg_epilogue<1, 1, WAVES_M, BN, true>(p, acc, (float*)smem, m0, n0, wm, wn, lane, tid, wave / WAVES_N, bx, ks, ooy0, oox0, ph, pw, true,
                                       pre_d, pre_x);
}

// ------------------------------------------------------------------------------------------------ transposed passes from a PATCH
// k_pconv_dma walks a transposed pass (conv data-gradient, full-conv forward) as four independent 2x2-tap GEMMs, one per output-parity
// class, and fills its LDS stages tap by tap: every low-res pixel of the operand travels L2 -> LDS 4 (taps) x 4 (classes) times, the
// weights once per 128-row tile and class.  For the 64-output-channel layers, whose tiles have only 4-8 K steps, that fill stream IS
// the kernel: E2's data-gradient moves 590 MB through L2 in 55 us (10.7 TB/s of the ~17 the L2 can feed to LDS-DMA,
// MI355X_MICROARCH.md "Indexed rows: gather into LDS") for 20 us of MFMAs.  Here a block owns an 8 x 16 region of the low-res grid for
// CPB = 2 or 4 parity classes at once: the region's pixels with a one-pixel halo are staged ONCE per 64-channel chunk as a patch
// ([3 planes][PR x 18 slots][64 ch], 62-69 KB), every (class, tap) A fragment is a ds_read_b128 from that patch at a per-lane slot
// (row (ry, rx) of the tile, tap (th, tw) of class (ph, pw): slot (ry + ph + th) * 18 + rx + pw + tw), and only the weights stream
// through two 24 KB stages, one (class, tap) step of 24 MFMAs per wave each.  Per 512 outputs x 64 channels of K: 69 + 393 KB instead
// of 4 x 288.  N % 64 == 0 (a block serves one 64-column slice), C % 64 == 0, no split-K; 8 waves of one 32 x 32 accumulator per class.
// CPB = 2: the block's classes share ph (its patch has 9 rows); twice the blocks — for the grids that would otherwise leave CUs idle.
// Grids with Wi % 16 == 0, Hi % 8 == 0.  (Tried and removed: 8 x 8 grids as blocks of two whole images with 10-column patches, which
// would have taken the 256-channel layers as well — four channel chunks per block, the patch restaged for each: 4 x 47.0 us against
// 3 x 36.5 + 60.9 us for k_pconv_dma on the same four launches of configs[1].)
template <int CPB>
__global__ __launch_bounds__(512) void k_pconv_patch_tr(const PGemm p) {
  constexpr int PR = CPB == 4 ? 10 : 9, PW = 18, NSLOT = PR * PW, NG = (NSLOT + 7) / 8;      // patch slots, 8-slot DMA groups
  constexpr int AGW = (NG + 7) / 8;                                                         // patch groups per wave
  constexpr int PATCH_PL = NG * 8 * 128, PATCH_BYTES = 3 * PATCH_PL;                         // bytes
  constexpr int W_PL = 64 * 128, WBUF = 3 * W_PL;                                            // one (class, tap) weight stage
  constexpr int NS = 4 * CPB;                                                               // (class, tap) steps per channel chunk
  __shared__ __attribute__((aligned(1024))) unsigned char smem[PATCH_BYTES + 2 * WBUF + 2 * 4 * 64 * 4];
  float* red = (float*)(smem + PATCH_BYTES + 2 * WBUF);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
  const int lr = lane & 31, lh = lane >> 5;
  int lid = pg_xcd_remap(blockIdx.x, gridDim.x);
  const int n0 = 64 * (lid % p.gn);                         // this block's column slice (neighbours in the grid share the patch's lines)
  lid /= p.gn;
  int ph_blk = 0;
  if constexpr (CPB == 2) {
    ph_blk = lid & 1;
    lid >>= 1;
  }
  const int tiles_x = p.Wi >> 4, tiles_y = p.Hi >> 3;
  const int tx = lid % tiles_x, ty = (lid / tiles_x) % tiles_y, b = lid / (tiles_x * tiles_y);
  const int ry0 = 8 * ty, rx0 = 16 * tx;
  const int bx = lid;                                       // row-tile index (batch-major): the statistics' partial row
  // the 16-byte k-octets of a patch slot are XOR-swizzled by its patch COLUMN: a wave's 32 rows are two runs of 16 columns in adjacent
  // patch rows, and with (px >> 1) & 7 every 16-lane group of a ds_read_b128 covers all 64 banks once for each of the nine window
  // offsets (checked exhaustively against MI355X_MICROARCH.md's lane groups; by slot index, the usual choice, the 18-slot pitch makes
  // every group two-way conflicted: 8 cycles per read)
  auto patch_sw = [](int px) { return (px >> 1) & 7; };

  // ---- this lane's share of the patch: slot 8 * group + lane / 8, LDS octet lane % 8 = global octet ^ swizzle(slot's column)
  unsigned a_byte[AGW], a_lds[AGW];
#pragma unroll
  for (int i = 0; i < AGW; ++i) {
    const int grp = wave + 8 * i;
    const int sl = 8 * grp + (lane >> 3);
    const int py = sl / PW, px = sl - py * PW;
    const int il = ry0 - 1 + (CPB == 2 ? ph_blk : 0) + py, jl = rx0 - 1 + px;
    const bool ok = grp < NG && sl < NSLOT && (unsigned)il < (unsigned)p.Hi && (unsigned)jl < (unsigned)p.Wi;
    const int oct = (lane & 7) ^ patch_sw(px);
    a_byte[i] = ok ? 2u * (unsigned)(((b * p.Hi + il) * p.Wi + jl) * p.C + 8 * oct) : VF_OOB;
    a_lds[i] = (unsigned)(grp * 8 * 128);
  }
  // ---- and of every weight stage: row n = 8 * wave + lane / 8 of the 64
  const int wrow = 8 * wave + (lane >> 3);
  const unsigned w_byte = 2u * (unsigned)((n0 + wrow) * 16 * p.C + 8 * ((lane & 7) ^ ((wrow >> 1) & 7)));
  const unsigned w_lds = (unsigned)(wave * 8 * 128);
  const __amdgpu_buffer_rsrc_t rsA = pg_rsrc(p.A, p.a_bytes), rsW = pg_rsrc(p.W, p.w_bytes);
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)smem;
  unsigned szero;
  asm volatile("s_mov_b32 %0, 0" : "=s"(szero));

  auto dma_patch = [&](int ch) {
    const unsigned cb = 128u * (unsigned)ch;
#pragma unroll
    for (int i = 0; i < AGW; ++i) {
      if (wave + 8 * i < NG) {                              // (wave-uniform)
#pragma unroll
        for (int q = 0; q < 3; ++q)
          pg_dma16(lds0 + (unsigned)(q * PATCH_PL) + a_lds[i], a_byte[i] == VF_OOB ? VF_OOB : a_byte[i] + cb, rsA, szero + q * p.a_ps);
      }
    }
  };
  // class `cls` of this block: (ph, pw); its window tap (th, tw) reads filter tap (3 - ph - 2 th, 3 - pw - 2 tw)
  auto dma_w = [&](int ch, auto STEP, int buf) {
    constexpr int s_ = decltype(STEP)::value, cls = s_ >> 2, th = (s_ >> 1) & 1, tw = s_ & 1;
    const int ph = CPB == 4 ? (cls >> 1) : ph_blk, pw = CPB == 4 ? (cls & 1) : cls;
    const int kh = 3 - ph - 2 * th, kw = 3 - pw - 2 * tw;
    const unsigned tW = 2u * (unsigned)((kh * 4 + kw) * p.C) + 128u * (unsigned)ch;
#pragma unroll
    for (int q = 0; q < 3; ++q)
      pg_dma16(lds0 + (unsigned)(PATCH_BYTES + buf * WBUF + q * W_PL) + w_lds, w_byte, rsW, q * p.w_ps + tW);
  };

  f32x16 acc[CPB];
#pragma unroll
  for (int c = 0; c < CPB; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;

  // this lane's A row of the class tile: (ry, rx) -> first slot of its window; B row: wn + lr.  (slot00 / b_off0 / pix0 pass through an
  // empty asm at the top of every chunk: left visible as loop invariants, the 16 unrolled steps' fragment addresses and the four
  // epilogues' element offsets are all hoisted in front of the loop — 256 registers and a kilobyte of scratch)
  const int arow = wm + lr;
  const int apy = arow >> 4, apx = arow & 15;              // tile row -> (ry, rx)
  int slot00 = apy * PW + apx;
  const int brow = wn + lr;
  unsigned b_off0 = (unsigned)(brow * 128);
  const unsigned b_sw = (unsigned)((brow >> 1) & 7);
  auto compute_step = [&](auto STEP, int buf) {
    constexpr int s_ = decltype(STEP)::value, cls = s_ >> 2, th = (s_ >> 1) & 1, tw = s_ & 1;
    const int ph = CPB == 4 ? (cls >> 1) : 0, pw = CPB == 4 ? (cls & 1) : cls;      // (CPB = 2: the patch origin carries ph)
    const int slot = slot00 + (ph + th) * PW + pw + tw;
    const unsigned a_off0 = (unsigned)(slot * 128), a_sw = (unsigned)patch_sw(apx + pw + tw);
    const unsigned char* wb = smem + PATCH_BYTES + buf * WBUF;
    bf16x8 a[2][3], bb[2][3];
    auto read_frag = [&](int g, int set) {
      const unsigned o = (unsigned)(2 * g + lh);
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        a[set][q] = *(const bf16x8*)(smem + q * PATCH_PL + a_off0 + ((o ^ a_sw) << 4));
        bb[set][q] = *(const bf16x8*)(wb + q * W_PL + b_off0 + ((o ^ b_sw) << 4));
      }
    };
    read_frag(0, 0);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int cs = g & 1;
      if (g + 1 < 4) read_frag(g + 1, cs ^ 1);
      __builtin_amdgcn_sched_barrier(0);
      acc[cls] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[cs][1], bb[cs][1], acc[cls], 0, 0, 0);      // smallest terms first
      acc[cls] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[cs][0], bb[cs][2], acc[cls], 0, 0, 0);
      acc[cls] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[cs][2], bb[cs][0], acc[cls], 0, 0, 0);
      acc[cls] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[cs][0], bb[cs][1], acc[cls], 0, 0, 0);
      acc[cls] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[cs][1], bb[cs][0], acc[cls], 0, 0, 0);
      acc[cls] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[cs][0], bb[cs][0], acc[cls], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  // ---- epilogue of one class (C/D layout: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)): bias, (Leaky)ReLU, the
  //      derivative mask, BatchNorm partials — pg_epilogue's arithmetic.  What it reads besides the accumulators (mask, BatchNorm
  //      input) is fetched when the class's first step of the last chunk begins, 16 values each per lane.
  const int n = wn + lr;                // column inside the block's slice; n0 + n in the tensor
  const int N = p.N;
  const int stm = p.st.mode;
  const bool want_d = p.dmask != nullptr, want_x = stm == 2;
  const float bv = p.bias ? p.bias[n0 + n] : 0.f;
  const float sv = stm ? p.st.vec[(stm == 2 ? (bx / p.st.tiles_per_group) * N : 0) + n0 + n] : 0.f;
  const float neg = p.act == VF_ACT_LRELU ? p.slope : (p.act == VF_ACT_RELU ? 0.f : 1.f);
  const float dneg = p.dact == VF_ACT_LRELU ? p.dslope : (p.dact == VF_ACT_RELU ? 0.f : 1.f);
  const unsigned ybytes = (unsigned)(p.out_elems * 4);
  const __amdgpu_buffer_rsrc_t rsY = pg_rsrc(p.Y, ybytes), rsD = pg_rsrc(p.dbits ? (const void*)p.dbits : (const void*)p.dmask, p.dbits ? ybytes / 32 : ybytes),
                               rsX = pg_rsrc(p.st.x, ybytes);
  // output pixel of (row r of this lane, class (ph, pw)), 32-bit (the host checks the extent): row = wm + (r & 3) + 4 lh + 8 (r >> 2),
  // (oy, ox) = (2 (ry0 + row / 16) + ph, 2 (rx0 + row % 16) + pw) = pix0 + (ph outW + pw) + (r >> 3) 2 outW + 2 ((r & 3) + 8 ((r >> 2) & 1))
  unsigned pix0 = (unsigned)(((b * p.outH + 2 * (ry0 + 2 * (wave >> 1))) * p.outW) + 2 * (rx0 + 4 * lh));
  const unsigned ostride = (unsigned)(2 * p.outW);
  float pre_d[16], pre_x[16];
  auto pixel = [&](int r, int ph, int pw) -> unsigned {
    return pix0 + (unsigned)(ph * p.outW + pw) + (unsigned)(r >> 3) * ostride + (unsigned)(2 * ((r & 3) + 8 * ((r >> 2) & 1)));
  };
  auto elem = [&](int r, int ph, int pw) -> unsigned { return pixel(r, ph, pw) * (unsigned)N + (unsigned)(n0 + n); };
  auto prefetch_epi = [&](int ph, int pw) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const unsigned px_ = pixel(r, ph, pw), e = px_ * (unsigned)N + (unsigned)(n0 + n);
      pre_d[r] = 1.f;
      pre_x[r] = 0.f;
      if (want_d) {
        // sign bits: word 2 * (column / 64) + (column & 1) of the pixel's N / 32 words, bit (column / 2) % 32
        if (p.dbits)
          pre_d[r] = (float)((__builtin_amdgcn_raw_buffer_load_b32(rsD, 4u * (px_ * (unsigned)(N >> 5) + (unsigned)(2 * (n0 >> 6) + (n & 1))), 0, 0) >> ((n >> 1) & 31)) & 1u);
        else pre_d[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsD, 4u * e, 0, 0));
      }
      if (want_x) pre_x[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsX, 4u * e, 0, 0));
    }
  };
  auto epilogue = [&](auto CLS, int ph, int pw) {
    constexpr int cls = decltype(CLS)::value;
    float st1[1] = {0.f}, st2[1] = {0.f};
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float v = acc[cls][r] + bv;
      v = v * (v > 0.f ? 1.f : neg);
      if (want_d) v = v * (pre_d[r] > 0.f ? 1.f : dneg);
      if (stm == 1) {
        const float d = v - sv;
        st1[0] += d;
        st2[0] += d * d;
      } else if (stm == 2) {
        st1[0] += v;
        st2[0] += v * (pre_x[r] - sv);
      }
      acc[cls][r] = v;         // the finished value stays in the accumulator's registers: the stores leave after the last step
    }
    if (stm) pg_bn_tile_partials<1, 4, 64>(p.st, st1, st2, red, wave >> 1, wn, lane, tid, n0, N, bx, (ph << 1) | pw);
  };
  // (every step begins with s_waitcnt vmcnt(0) for its DMAs, which would also wait for any store still on its way: 16 stores per lane
  //  and class, issued between the classes, stalled the next step for the whole write latency — in the iteration, with the memory
  //  system busy, more than in a microbenchmark)
  auto store_class = [&](auto CLS, int ph, int pw) {
    constexpr int cls = decltype(CLS)::value;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float v = acc[cls][r];      // (a copy: __builtin_bit_cast applied to the vector-element lvalue itself read element 0 sixteen times)
      if (!(p.dbg & 8)) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsY, 4u * elem(r, ph, pw), 0, 0);
    }
  };

  // ---- main loop: channel chunk outer (the patch is restaged per chunk), (class, tap) steps inner, weights one step ahead
  const int nch = p.nchunks;
  dma_patch(0);
  dma_w(0, VfIntC<0>{}, 0);
  for (int ch = 0; ch < nch; ++ch) {
    const bool last = ch + 1 == nch;
    asm volatile("" : "+v"(slot00), "+v"(b_off0), "+v"(pix0));
    vf_static_for<NS>([&](auto S) {
      constexpr int s_ = decltype(S)::value, buf = s_ & 1, cls = s_ >> 2, tap = s_ & 3;
      const int ph = CPB == 4 ? (cls >> 1) : ph_blk, pw = CPB == 4 ? (cls & 1) : cls;
      // this wave's DMAs (patch, this step's weights) have landed; after the barrier everybody's have, and nobody still reads the
      // other weight stage
      asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      if (!(p.dbg & 1)) {
        if constexpr (s_ + 1 < NS) dma_w(ch, VfIntC<s_ + 1>{}, buf ^ 1);
        else if (!last) dma_w(ch + 1, VfIntC<0>{}, buf ^ 1);
      }
      if (last && tap == 0 && (want_d || want_x)) prefetch_epi(ph, pw);
      __builtin_amdgcn_sched_barrier(0);
      if (!(p.dbg & 4)) compute_step(S, buf);
      if (last && tap == 3) epilogue(VfIntC<cls>{}, ph, pw);
    });
    if (!last) {
      asm volatile("s_barrier" ::: "memory");            // everybody has read the patch: the next chunk's goes in
      dma_patch(ch + 1);
    }
  }
  vf_static_for<CPB>([&](auto C) {
    constexpr int cls = decltype(C)::value;
    store_class(C, CPB == 4 ? (cls >> 1) : ph_blk, CPB == 4 ? (cls & 1) : cls);
  });
}

// ------------------------------------------------------------------------------------------------ gather passes from a PATCH
// k_pconv_dma fills its stages tap by tap: per 64-channel K step a 128 x 64 tile moves 48 KB of A and 24 KB of W from L2 into LDS for
// 0.64 us of matrix-pipe work — 29 TB/s chip-wide at the pipe's rate, against the ~17 TB/s LDS-DMA gathers get out of L2
// (MI355X_MICROARCH.md "Indexed rows: gather into LDS"): the gather (t16) passes, the family's dominant symbol, are FILL-bound near
// 0.58 of their pipe bound and ran at 0.31.  A 4x4 stride-2 window reads each INPUT-PARITY class of the input
// through a 2x2-tap stride-1 window: taps (th, tw) with th = cth + 2a, tw = ctw + 2b read input (2 oy - 1 + th, 2 ox - 1 + tw) =
// sub-image pixel (oy - 1 + cth + a, ox - 1 + ctw + b) of class (cth, ctw) — the order k_pconv_dma already walks its 16 taps in
// (class outer, (a, b) inner).  So per (64-channel chunk, class) a block stages the class's sub-image region ONCE as a patch and
// reads the A fragments of its four taps from it at per-lane slots; only the weights stream, one (tap, 64 ch) stage of 24 KB per
// step: 14.7 + 24.6 KB per step instead of 72.
//   MODE 0: an 8 x 16 tile of an output map with Wo % 16 == 0, Ho % 8 == 0: patch = 9 x 17 sub-image pixels (one halo row and column;
//           pixels outside the image arrive as zeros: out-of-range DMA lanes), slot (r, c) = r * 17 + c.
//   MODE 1 / 2: whole output maps of 8 x 8 / 4 x 4 (two / eight per 128-row tile): the class's sub-image IS Ho x Wo pixels per map
//           and every halo pixel is padding: 128 real slots + two all-zero slots that the lanes of padding taps read instead.
// LDS (147 KB, one block per CU): the patch's hi and mid planes are double-buffered (the next unit's travel beside this unit's four
// steps); the lo plane — read by ONE of the six product terms — is single: its fragments of a unit's LAST step are read one step
// early into 16 registers, so the next unit's lo plane can be fetched during that last step.  (All three planes double-buffered
// are 2.7 KB more than a CU has beside two weight stages.)  16-byte k-octets of a slot are XOR-swizzled by (patch column & 7)
// (MODE 0) / (slot >> 1) & 7 (MODE 1, 2): every ds_read_b128 lane group covers the 64 banks once for all sixteen (class, tap)
// windows (checked exhaustively against MI355X_MICROARCH.md's lane groups; padding lanes read the zero slots at the bank position
// their out-of-image slot would have had).
// Measured on E3's forward pass (4.3 GFLOP, 256 tiles of 16 steps; scripts/bench_pconv.py, VF_PG_DBG ablations, one box): 30.7 us against
// 33.2 for k_pconv_dma; an empty launch of the same shape 9.0, MFMAs alone 25.3, DMAs alone 19.4 (k_pconv_dma: 25.2) — the fill
// stream is 1.6x shorter, the matrix phase (1.0 us per 24-MFMA step and wave pair, 0.7 of it pipe time) now bounds the tile.  Two
// role splits between a SIMD's two waves were built on this kernel, bit-identical, and dropped (profiles/r05_*gather*): waves 4-7
// only fetching and waves 0-3 owning 32 x 64 strips with all 48 MFMAs of their SIMD (33.1 us: one wave per SIMD leaves the pipe idle
// while it issues its own fragment reads, and four waves store the whole tile); both waves computing but issuing their DMAs at
// opposite ends of the step — waves 0-3 the weights in front of their MFMAs, waves 4-7 the next unit's patch planes behind theirs
// (33.0 against 31.4 on one box: no better).  v_mfma_f32_16x16x32_bf16 in place of 32x32x16 (the same cycles; the guide's
// higher-clock shape) moved the tile by 3-5 % in a timing-only build, not enough to give up bit-identity with the other kernels.
// What the step is made of (in-kernel stamps, diagnostic build; profiles/r05_g_gather_patch_stamps_and_fetch_waves.txt): ~180 cycles
// from the barrier to the first MFMA (fragment reads), ~1 530 in which the SIMD's 48 MFMAs (1 536 cycles of pipe time) are issued —
// the older wave of a pair finishes its 24 after ~1 140 and then waits ~1 000 at the barrier for the younger, which gets the pipe's
// leftover slots — ~130 in s_waitcnt, ~270 in the barrier for the wave that arrives last: ~2 360 per step, the pipe 65 % busy.  Built on
// that reading and measured without gain (all bit-identical; the same file): FOUR DEDICATED FETCH WAVES (blocks of 12 waves: waves 8-11
// issue every DMA, the computing waves no vector-memory instruction) — the stamped step is then the same 2 364 cycles with the DMAs as
// without them, i.e. the fill stream costs the computing waves no cycle any more, and the launch takes the same 28.7 us (28.5 without
// fetch waves): what the DMAs cost is CLOCK, not cycles (same cycle count, 47.5 against 41.5 us in the stamped build); fragment reads
// two k-groups ahead; the next group's reads interleaved one by one with the MFMAs (sched_group_barrier: 31.7 against 29.0 us);
// s_setprio for the SIMD's younger wave.  Every variant lands within 3 % of 29 us on E3 — the kernel sits at what the chip's clock
// management gives a loop of this switching activity, and what did move it is what removed bytes (the fill: 72 -> 39 KB per step).
// Same six-term products in the same K order per output element as k_pconv_dma: bit-identical results.  N % 64 == 0, C % 64 == 0,
// split-K over channel chunks as k_pconv_dma.  Semantics: nn.SpatialConvolution forward (train.lua:89-101, 183-193) and
// nn.SpatialFullConvolution's data-gradient (train.lua:134-146).
// ST: diagnostic build (VF_PG_STAMPS): lane 0 of every wave of the first 64 blocks stamps the shader clock at four points of each of its
// first 16 steps — after the barrier, after its DMA issue, after its last MFMA, after its s_waitcnt — into p.stamps
// (scripts/probe/pg_stamp_report.py)
// TR (MODE 1, 2): the TRANSPOSED passes (conv data-gradient, full-conv forward) whose low-resolution grid is a whole 8 x 8 / 4 x 4 map —
// the ones k_pconv_patch_tr's 8 x 16 regions do not cover (E4 / E5 and netD's deeper data-gradients, D2 / D3 forward: 8 launches of
// configs[1]).  A block serves ONE output-parity class (ph, pw) of its 128 low-resolution rows, as k_pconv_dma does (the grid keeps
// its four blocks per row tile); the patch is the low-resolution map itself (128 pixels + the zero slots for the padding taps), staged
// once per 64-channel chunk and read by the class's four taps (th, tw) at (my + ph - 1 + th, mx + pw - 1 + tw): a unit is a chunk,
// four steps each — 12 + 24.6 KB of fill per step instead of 72.  Same planes schedule, same products, same K order as k_pconv_dma
// <., ., 4>: bit-identical results.
template <int MODE, bool ST = false, bool TR = false>
__global__ __launch_bounds__(512) void k_pconv_patch_g(const PGemm p) {
  static_assert(!TR || MODE != 0, "the transposed form serves whole-map tiles (8 x 16 regions of larger maps: k_pconv_patch_tr)");
  constexpr int NS = TR ? 4 : 16;                           // steps per channel chunk
  constexpr int PW = 17;
  constexpr int NSLOT = MODE == 0 ? 9 * PW : 130;           // MODE 1, 2: 128 real slots + the zero slots 128, 129
  constexpr int NG = (NSLOT + 7) / 8, AGW = (NG + 7) / 8;   // 8-slot DMA groups; groups per wave
  constexpr int PL = NG * 8 * 128;                          // one plane image, bytes (a multiple of 1 KB)
  constexpr int W_PL = 64 * 128, WBUF = 3 * W_PL;           // one (tap, 64 ch) weight stage
  constexpr int OFF_LO = 4 * PL, OFF_W = 5 * PL;            // [hi0 | mid0 | hi1 | mid1 | lo | W0 | W1]
  __shared__ __attribute__((aligned(1024))) unsigned char smem[5 * PL + 2 * WBUF];
  float* red = (float*)(smem + OFF_W);                      // (BatchNorm partials of the epilogue: the weight stages are dead by then)

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
  const int lr = lane & 31, lh = lane >> 5;
  const int ntiles = p.gm * p.gn * p.gz;
  int lid = pg_xcd_remap(blockIdx.x, ntiles);
  int ph = 0, pw = 0;                                       // TR: this block's output-parity class
  if constexpr (TR) {
    ph = (lid >> 1) & 1;
    pw = lid & 1;
    lid >>= 2;
  }
  const int bx = lid % p.gm, byz = lid / p.gm;
  const int n0 = (byz % p.gn) * 64;
  const int ks = byz / p.gn;
  const int cps = p.nchunks / p.ksplit;
  const int ch0 = ks * cps;
  const int Wo = 1 << p.lgMw, Ho = 1 << p.lgMh;

  // ---- tile geometry
  int b_img = 0, oy0 = 0, ox0 = 0;                            // MODE 0: image and tile origin on the output map
  if constexpr (MODE == 0) {
    const int tiles_x = Wo >> 4, tiles_y = Ho >> 3;
    const int tx = bx % tiles_x, ty = (bx / tiles_x) % tiles_y;
    b_img = bx / (tiles_x * tiles_y);
    oy0 = 8 * ty;
    ox0 = 16 * tx;
  }
  constexpr int HW = MODE == 1 ? 64 : 16, LW = MODE == 1 ? 3 : 2;      // MODE 1, 2: pixels per map, log2 of its width

  // ---- this lane's share of every patch: slot 8 * group + lane / 8, LDS octet lane % 8 = global octet ^ swizzle(slot).
  //      a_byte: the slot's pixel of class (0, 0); class (cth, ctw) is cls_off(cth, ctw) bytes away; a_ok bit (2 cth + ctw): in the image
  unsigned a_byte[AGW], a_ok[AGW], a_lds[AGW];
#pragma unroll
  for (int i = 0; i < AGW; ++i) {
    const int grp = wave + 8 * i;
    const int sl = 8 * grp + (lane >> 3);
    a_lds[i] = (unsigned)(grp * 8 * 128);
    if constexpr (MODE == 0) {
      const int r = sl / PW, c = sl - r * PW;
      const int y00 = 2 * (oy0 + r) - 1, x00 = 2 * (ox0 + c) - 1;
      const int oct = (lane & 7) ^ (c & 7);
      a_byte[i] = 2u * (unsigned)(((b_img * p.Hi + y00) * p.Wi + x00) * p.C + 8 * oct);
      unsigned ok = 0;
#pragma unroll
      for (int cls = 0; cls < 4; ++cls)
        if ((unsigned)(y00 + (cls >> 1)) < (unsigned)p.Hi && (unsigned)(x00 + (cls & 1)) < (unsigned)p.Wi) ok |= 1u << cls;
      a_ok[i] = (grp < NG && sl < NSLOT) ? ok : 0u;
    } else {
      const int g = sl >> (2 * LW), ii = (sl >> LW) & (Wo - 1), jj = sl & (Wo - 1);
      const int oct = (lane & 7) ^ ((sl >> 1) & 7);
      const int img = bx * (128 / HW) + g;
      // class (cth, ctw) of sub-image pixel (ii, jj) is input pixel (2 ii + 1 - cth, 2 jj + 1 - ctw): always inside the image;
      // TR: the low-resolution pixel (ii, jj) itself
      if constexpr (TR) a_byte[i] = 2u * (unsigned)(((img * p.Hi + ii) * p.Wi + jj) * p.C + 8 * oct);
      else a_byte[i] = 2u * (unsigned)(((img * p.Hi + 2 * ii + 1) * p.Wi + 2 * jj + 1) * p.C + 8 * oct);
      a_ok[i] = (grp < NG && sl < 128) ? 15u : 0u;
    }
  }
  // ---- and of every weight stage: row n = 8 * wave + lane / 8 of the 64
  const int wrow = 8 * wave + (lane >> 3);
  const unsigned w_byte = 2u * (unsigned)((n0 + wrow) * 16 * p.C + 8 * ((lane & 7) ^ ((wrow >> 1) & 7)));
  const unsigned w_lds = (unsigned)(wave * 8 * 128);
  const __amdgpu_buffer_rsrc_t rsA = pg_rsrc(p.A, p.a_bytes), rsW = pg_rsrc(p.W, p.w_bytes);
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)smem;
  unsigned szero;
  asm volatile("s_mov_b32 %0, 0" : "=s"(szero));
  int rowA = 2 * p.Wi * p.C, colA = 2 * p.C;                  // one input row / pixel, bytes

  // plane q (0 hi, 1 mid, 2 lo) of unit (chunk ch, class cls) into patch buffer pb
  auto dma_patch_plane = [&](int ch, int cls, int pb, auto Q, bool live) {
    constexpr int q = decltype(Q)::value;
    const int cth = cls >> 1, ctw = cls & 1;
    const unsigned cb = 128u * (unsigned)ch + (TR ? 0u : (unsigned)(MODE == 0 ? cth * rowA + ctw * colA : -(cth * rowA + ctw * colA)));
    const unsigned dst = lds0 + (unsigned)(q == 2 ? OFF_LO : (2 * pb + q) * PL);
    if (!live) return;                                        // (wave-uniform: the block's last unit has no successor)
#pragma unroll
    for (int i = 0; i < AGW; ++i) {
      if (wave + 8 * i < NG) {                                // (wave-uniform)
        const bool ok = (a_ok[i] >> cls) & 1u;
        pg_dma16(dst + a_lds[i], ok ? a_byte[i] + cb : VF_OOB, rsA, szero + q * p.a_ps);
      }
    }
  };
  // weight stage of step s_ (tap th = cth + 2a, tw = ctw + 2b: filter tap (th, tw)) of chunk ch
  auto dma_w = [&](int ch, auto STEP, int buf, bool live) {
    constexpr int s_ = decltype(STEP)::value;
    constexpr int th = ((s_ >> 3) & 1) + 2 * ((s_ >> 1) & 1), tw = ((s_ >> 2) & 1) + 2 * (s_ & 1);
    // TR: window tap (th, tw) = (s_ >> 1, s_ & 1) of class (ph, pw) meets filter tap (3 - ph - 2 th, 3 - pw - 2 tw)
    const int ftap = TR ? (3 - ph - 2 * (s_ >> 1)) * 4 + 3 - pw - 2 * (s_ & 1) : th * 4 + tw;
    const unsigned tW = 2u * (unsigned)(ftap * p.C) + 128u * (unsigned)ch;
    if (!live) return;
#pragma unroll
    for (int q = 0; q < 3; ++q) pg_dma16(lds0 + (unsigned)(OFF_W + buf * WBUF + q * W_PL) + w_lds, w_byte, rsW, q * p.w_ps + tW);
  };

  f32x16 acc[1][1];
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[0][0][r] = 0.f;

  // ---- this lane's A row of the tile and its first slot; B row wn + lr
  const int arow = wm + lr;
  int slot_base;                                            // MODE 0: slot of tap (a, b) = (0, 0); MODE 1, 2: slot of sub-image pixel (oy - 1, ox - 1)
  unsigned tap_ok = 0xffffu;                                // MODE 1, 2: bit s_ = step s_'s tap lies inside the map
  int a_col = 0;                                            // MODE 0: patch column of tap b = 0
  if constexpr (MODE == 0) {
    a_col = arow & 15;
    slot_base = (arow >> 4) * PW + a_col;
  } else {
    const int oy = (arow >> LW) & (Wo - 1), ox = arow & (Wo - 1);
    slot_base = (arow >> (2 * LW)) * HW + (oy - 1 + ph) * Wo + (ox - 1 + pw);      // (ph = pw = 0 unless TR)
    tap_ok = 0;
#pragma unroll
    for (int s_ = 0; s_ < NS; ++s_) {
      const int di = (TR ? ph : ((s_ >> 3) & 1)) + ((s_ >> 1) & 1) - 1, dj = (TR ? pw : ((s_ >> 2) & 1)) + (s_ & 1) - 1;
      if ((unsigned)(oy + di) < (unsigned)Ho && (unsigned)(ox + dj) < (unsigned)Wo) tap_ok |= 1u << s_;
    }
  }
  const int brow = wn + lr;
  unsigned b_off0 = (unsigned)(brow * 128);
  const unsigned b_sw = (unsigned)((brow >> 1) & 7);

  // A fragment address pieces of step s_: byte offset of the slot inside a plane image, swizzle
  auto a_addr = [&](auto STEP, unsigned& off, unsigned& sw) {
    constexpr int s_ = decltype(STEP)::value;
    constexpr int cth = TR ? 0 : (s_ >> 3) & 1, ctw = TR ? 0 : (s_ >> 2) & 1, a = (s_ >> 1) & 1, b = s_ & 1;
    if constexpr (MODE == 0) {
      off = (unsigned)((slot_base + a * PW + b) * 128);
      sw = (unsigned)((a_col + b) & 7);
    } else {
      const int raw = slot_base + (cth + a) * Wo + ctw + b;
      const bool ok = (tap_ok >> s_) & 1u;
      off = (unsigned)((ok ? raw : (128 | (raw & 1))) * 128);
      sw = (unsigned)((raw >> 1) & 7);
    }
  };

  bf16x8 lo_held[4];                                        // the lo fragments of a unit's last step (read during the step before)
  auto read_lo_ahead = [&](auto STEP) {
    unsigned off, sw;
    a_addr(STEP, off, sw);
#pragma unroll
    for (int g = 0; g < 4; ++g) lo_held[g] = *(const bf16x8*)(smem + OFF_LO + off + ((((unsigned)(2 * g + lh)) ^ sw) << 4));
  };
  auto compute_step = [&](auto STEP, int pb, int wbuf) {
    constexpr int s_ = decltype(STEP)::value;
    constexpr bool held = (s_ & 3) == 3;
    unsigned a_off, a_sw;
    a_addr(STEP, a_off, a_sw);
    const unsigned char* pa = smem + 2 * pb * PL + a_off;
    const unsigned char* pl = smem + OFF_LO + a_off;
    const unsigned char* wb = smem + OFF_W + wbuf * WBUF + b_off0;
    bf16x8 a[2][3], bb[2][3];
    auto read_frag = [&](int g, int set) {
      const unsigned o = (unsigned)(2 * g + lh);
      a[set][0] = *(const bf16x8*)(pa + ((o ^ a_sw) << 4));
      a[set][1] = *(const bf16x8*)(pa + PL + ((o ^ a_sw) << 4));
      if constexpr (!held) a[set][2] = *(const bf16x8*)(pl + ((o ^ a_sw) << 4));
#pragma unroll
      for (int q = 0; q < 3; ++q) bb[set][q] = *(const bf16x8*)(wb + q * W_PL + ((o ^ b_sw) << 4));
    };
    read_frag(0, 0);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int cs = g & 1;
      if (g + 1 < 4) read_frag(g + 1, cs ^ 1);
      __builtin_amdgcn_sched_barrier(0);
      bf16x8 a2;
      if constexpr (held) a2 = lo_held[g];
      else a2 = a[cs][2];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[cs][1], bb[cs][1], acc[0][0], 0, 0, 0);      // smallest terms first
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[cs][0], bb[cs][2], acc[0][0], 0, 0, 0);
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, bb[cs][0], acc[0][0], 0, 0, 0);
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[cs][0], bb[cs][1], acc[0][0], 0, 0, 0);
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[cs][1], bb[cs][0], acc[0][0], 0, 0, 0);
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[cs][0], bb[cs][0], acc[0][0], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  // ---- the output pixel of a tile row, and what the epilogue reads besides the accumulators (derivative mask, BatchNorm input):
  //      fetched now, beside the first stages (k_pconv_dma's pre_d / pre_x)
  auto pix_of = [&](int row) -> int64_t {
    if constexpr (MODE == 0) return ((int64_t)b_img * Ho + oy0 + (row >> 4)) * Wo + ox0 + (row & 15);
    else if constexpr (!TR) return (int64_t)bx * 128 + row;
    else {      // low-resolution pixel (my, mx) of map bx * (128 / HW) + row / HW -> output pixel (2 my + ph, 2 mx + pw)
      const int img = bx * (128 / HW) + (row >> (2 * LW)), my = (row >> LW) & (Wo - 1), mx = row & (Wo - 1);
      return ((int64_t)img * (2 * Ho) + 2 * my + ph) * (2 * Wo) + 2 * mx + pw;
    }
  };
  float pre_d[16], pre_x[16];
  const bool fin_tile = p.ksplit == 1;
  const bool want_d = fin_tile && p.dmask != nullptr, want_x = fin_tile && p.st.mode == 2;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    pre_d[r] = 1.f;
    pre_x[r] = 0.f;
  }
  if (want_d || want_x) {
    const int n = n0 + wn + lr;
    // (sign bits instead of the fp32 activation where the producer left them — k_pconv_dma's layout: word 2 * (n / 64) + (n & 1) of
    //  the pixel, bit (n / 2) % 32)
    const int bw = 2 * (n >> 6) + (n & 1), bb = (n >> 1) & 31, wpp = p.N >> 5;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int64_t pix = pix_of(wm + (r & 3) + 8 * (r >> 2) + 4 * lh);
      if (want_d) pre_d[r] = p.dbits ? (float)((p.dbits[pix * wpp + bw] >> bb) & 1u) : p.dmask[pix * p.N + n];
      if (want_x) pre_x[r] = p.st.x[pix * p.N + n];
    }
  }

  // ---- main loop: unit = (channel chunk, class), four tap steps each; weights one step ahead, the next unit's hi / mid planes during
  //      steps 0 / 1 into the other patch buffer, its lo plane during step 3 (whose lo fragments were read during step 2)
  dma_patch_plane(ch0, 0, 0, VfIntC<0>{}, !(p.dbg & 32));
  dma_patch_plane(ch0, 0, 0, VfIntC<1>{}, !(p.dbg & 32));
  dma_patch_plane(ch0, 0, 0, VfIntC<2>{}, !(p.dbg & 32));
  dma_w(ch0, VfIntC<0>{}, 0, !(p.dbg & 32));
  for (int c = 0; c < cps; ++c) {
    const int ch = ch0 + c;
    asm volatile("" : "+v"(slot_base), "+v"(b_off0), "+s"(rowA), "+s"(colA));
    vf_static_for<NS>([&](auto S) {
      // gather: 4 units (classes) per chunk, the patch buffer's parity follows the class; TR: ONE unit per chunk, it follows the chunk
      constexpr int s_ = decltype(S)::value, cls = s_ >> 2, t = s_ & 3, wbuf = s_ & 1;
      const int pb = TR ? (c & 1) : (cls & 1);
      const bool last_unit = (c + 1 == cps) && (TR || cls == 3);
      const int nch = (TR || cls == 3) ? ch + 1 : ch, ncls = TR ? 0 : ((cls + 1) & 3);      // the next unit
      auto stamp = [&](int slot) {
        if constexpr (ST) {
          if (c == 0 && lane == 0 && blockIdx.x < 64) p.stamps[(((int)blockIdx.x * 8 + wave) * 16 + s_) * 4 + slot] = (long long)__builtin_amdgcn_s_memtime();
        }
      };
      // this wave's DMAs of THIS step's weights have landed — and, at a unit's first step, of its patch — and its LDS reads have
      // returned (the lo fragments read ahead during step 2 in particular, whose plane the DMAs of step 3 overwrite); after the barrier
      // everybody's have.  The hi / mid planes of the NEXT unit, issued behind the weights of steps 0 / 1, are not needed before that
      // unit begins: steps 1 and 2 leave the (at least two) DMAs of the plane issued one step earlier in flight — vmcnt counts in
      // issue order, so the weights in front of them have landed — instead of waiting the whole fill stream out at every step
      if constexpr (t == 1 || t == 2) {
        if (last_unit) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // (no next unit: nothing but weights in flight)
        else asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      }
      stamp(3);          // (of the step that just ended: its slot 3 is written here, one step late — see the dump's reader)
      asm volatile("s_barrier" ::: "memory");
      stamp(0);
      __builtin_amdgcn_sched_barrier(0);
      if (!(p.dbg & 1)) {
        if constexpr (s_ + 1 < NS) dma_w(ch, VfIntC<s_ + 1>{}, wbuf ^ 1, true);
        else dma_w(ch + 1, VfIntC<0>{}, wbuf ^ 1, c + 1 < cps);
        if constexpr (t == 0) dma_patch_plane(nch, ncls, pb ^ 1, VfIntC<0>{}, !last_unit);
        if constexpr (t == 1) dma_patch_plane(nch, ncls, pb ^ 1, VfIntC<1>{}, !last_unit);
        if constexpr (t == 3) dma_patch_plane(nch, ncls, pb ^ 1, VfIntC<2>{}, !last_unit);
      }
      if constexpr (t == 2) read_lo_ahead(VfIntC<s_ + 1>{});
      __builtin_amdgcn_sched_barrier(0);
      stamp(1);
      if (!(p.dbg & 4)) compute_step(S, pb, wbuf);
      stamp(2);
    });
  }
  // every DMA has landed and nobody still reads a weight stage when the epilogue's partial sums go there
  asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
  typedef const float (&pre_t)[1][16];
  pg_epilogue_at<1, 1, 4, 64, true>(p, acc, red, n0, wm, wn, lane, tid, wave >> 1, bx, ks, ph, pw, true, reinterpret_cast<pre_t>(pre_d),
                                    reinterpret_cast<pre_t>(pre_x), pix_of);
}

// ------------------------------------------------------------------------------------------------ weight gradients from planes
// dW[n][tap][c] = sum_p U[p][n] * V[p @ tap][c] (VfPWGrad, vf_common.h): a 128 (n) x 128 (tap, c) tile per block, K = pixels in
// steps of 32.  Both operands are K-MAJOR in memory ([pixel][channel]), which is the layout gfx950's transposing LDS read
// wants: a stage is [3 planes][U 32 x 128 | V 32 x 128] bf16 = 48 KB, moved global -> LDS by DMA (one instruction = 4 pixel
// rows x 256 bytes of one plane; wave w owns rows 4w .. 4w+3 of both tiles in all three planes), fragments come out through
// ds_read_b64_tr_b16 (8 consecutive pixels of one channel per lane: the 32x32x16 operand map).  Rows are unpadded (the DMA
// image is lane-linear), so the four pixel rows one transposing read touches are spread over the banks by an XOR of the
// 64-byte block index with (row & 3), applied in the per-lane SOURCE address.  8 waves of a 32 x 64 accumulator tile.
// NST = 1 (default): ONE stage in LDS (48 KB, three blocks per CU): issue the stage's DMAs, wait, barrier, MFMAs, barrier — a
// block does not overlap its own loads and MFMAs, its CU-mates do; these grids are heavily split (tiles of 8-16 K steps, ten
// rounds of blocks), and what bounds them is the start-up and drain of each tile, which co-resident blocks cover.
// FS = 1 (default since round 5): ONE fragment set per wave — 73 registers instead of 124, so that three blocks per CU fit the register
// file as well as LDS (with two sets it held two: the occupancy this form was built around was never reached); -0.8 % on the iteration.
// NST = 3: three stages (144 KB, one block per CU), DMAs of stage k+2 issued before the MFMAs of stage k, `s_waitcnt vmcnt(6)`
// lets the newest stage stay in flight, one barrier per step: 20 % slower here (VF_PWG_STAGES=3).
// Layers WITHOUT planes (Up == NULL: the bottleneck pair, plain [K][Nu] x [K][16 Cv] fp32 matrices) are staged by the block
// itself — float4 loads, exact three-way split, 8-byte LDS writes into the same swizzled image — so that they share the launch.  Whole tiles only (Nu % 128 == 0, Cv % 64 == 0, P % 32 == 0): the host keeps
// everything else on vf_conv.hip's k_wgrad.
__device__ __forceinline__ bf16x8 pg_tr_frag(const __bf16* tile, int col0, int k0, int lane) {
  // tile: [32][128] bf16, 256-byte rows, 64-byte block b of row k stored at block b ^ (k & 3).  Lane l gets column
  // col0 + l % 32 (col0 a multiple of 32), k = k0 + 8 * (l / 32) .. + 7
  const int grp = lane >> 4, i = lane & 15;
  const int k = k0 + 8 * (grp >> 1) + (i >> 2);
  const int blk = (col0 >> 5) ^ ((i >> 2) & 3);                 // (k & 3) == (i >> 2) & 3: k0 and 8 * (grp >> 1) are multiples of 4
  const __bf16* a = tile + k * 128 + blk * 32 + 16 * (grp & 1) + 4 * (i & 3);
  typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
  const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)a);
  const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(a + 4 * 128));      // rows k + 4: the same (k & 3)
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

template <int NST, int NPL = 3, int FS = 1>
__global__ __launch_bounds__(512, (NST == 1 && FS == 1) ? 6 : 1) void k_pwgrad_group(const VfPWGradGroup G) {
  int l = 0;
  while (l + 1 < G.n && (int)blockIdx.x >= G.blk_off[l + 1]) ++l;
  const VfPWGrad& p = G.d[l];
  const int local = (int)blockIdx.x - G.blk_off[l];
  const int ntiles = p.gx * p.gy * p.gz;
  if (local >= ntiles) return;                       // padding blocks (uniform exit)
  constexpr int BK = 32, TILE = BK * 128, PL_SZ = 2 * TILE, ST_SZ = NPL * PL_SZ;      // bf16 elements
  __shared__ __attribute__((aligned(1024))) __bf16 smem[NST * ST_SZ];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = (wave >> 1) * 32, wn = (wave & 1) * 64;
  const int lid = pg_xcd_remap(local, ntiles);
  const int jx = __builtin_amdgcn_readfirstlane(lid % p.gx), rest = __builtin_amdgcn_readfirstlane(lid / p.gx);
  const int n0 = __builtin_amdgcn_readfirstlane(rest % p.gy) * 128, j0 = jx * 128;
  const int ks = __builtin_amdgcn_readfirstlane(rest / p.gy);
  const int steps = (p.nk + p.ksplit - 1) / p.ksplit;
  const int kt0 = ks * steps, kt1 = min(p.nk, kt0 + steps);

  // ---- this lane's piece of every stage: pixel row k = 4 * wave + lane / 16, 16-byte chunk lane % 16 of the 256-byte row;
  //      the chunk it FETCHES is the one whose swizzled home is that slot
  const int krow = 4 * wave + (lane >> 4);
  const int slot = lane & 15;
  const int chunk = (((slot >> 2) ^ (krow & 3)) << 2) | (slot & 3);            // logical 8-channel chunk of the tile row
  const unsigned u_off = 2u * (unsigned)(n0 + 8 * chunk);                      // + pixel * Nu * 2
  const int col = j0 + 8 * chunk;                                              // (tap, c) column of the V tile
  const int tap = col / p.Cv, cch = col - tap * p.Cv;
  const int dy = (tap >> 2) - 1, dx = (tap & 3) - 1;                           // stride 2, pad 1
  const int Mw = 1 << p.lgMw, Mh = 1 << p.lgMh;
  const __amdgpu_buffer_rsrc_t rsU = pg_rsrc(p.Up, (unsigned)NPL * p.u_ps), rsV = pg_rsrc(p.Vp, (unsigned)NPL * p.v_ps);
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)smem;
  const unsigned row_lds = 2u * (unsigned)(4 * wave * 128);                    // this wave's four rows inside a tile (bytes)
  unsigned szero;
  asm volatile("s_mov_b32 %0, 0" : "=s"(szero));

  auto dma_stage = [&](int kt, int st, bool live) {
    const int pix = kt * BK + krow;
    const int mx = pix & (Mw - 1), my = (pix >> p.lgMw) & (Mh - 1), b = pix >> (p.lgMw + p.lgMh);
    const int iy = 2 * my + dy, ix = 2 * mx + dx;
    const bool okv = live && (unsigned)iy < (unsigned)p.Hv && (unsigned)ix < (unsigned)p.Wv;
    const unsigned uo = live ? u_off + 2u * (unsigned)(pix * p.Nu) : VF_OOB;
    const unsigned vo = okv ? 2u * (unsigned)(((b * p.Hv + iy) * p.Wv + ix) * p.Cv + cch) : VF_OOB;
    const unsigned base = lds0 + 2u * (unsigned)(st * ST_SZ) + row_lds;
#pragma unroll
    for (int q = 0; q < NPL; ++q) {
      pg_dma16(base + 2u * (unsigned)(q * PL_SZ), uo, rsU, szero + q * p.u_ps);
      pg_dma16(base + 2u * (unsigned)(q * PL_SZ + TILE), vo, rsV, szero + q * p.v_ps);
    }
  };

  // fp32-fed layers: 32 rows x 128 columns of each operand per stage, 2 float4 per thread and operand; every float4 becomes
  // three 8-byte plane pieces written to the same swizzled K-major image the DMA form builds
  const bool fed32 = p.Up == nullptr;
  const int Ncols = 16 * p.Cv;
  auto stage_fp32 = [&](int kt) {
    const int c4 = (tid & 31) * 4;                     // 4 consecutive columns of the 128-wide tiles
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int k = (tid >> 5) + 16 * j;               // row of the stage
      const int pix = kt * BK + k;
      f32x4 u = {0.f, 0.f, 0.f, 0.f}, v = {0.f, 0.f, 0.f, 0.f};
      if (pix < p.P) {
        if (n0 + c4 < p.Nu) u = *(const f32x4*)(p.Uf + (int64_t)pix * p.Nu + n0 + c4);      // (Nu % 4 == 0: a float4 is in or out)
        v = *(const f32x4*)(p.Vf + (int64_t)pix * Ncols + j0 + c4);
      }
      const int off = k * 128 + (((c4 >> 5) ^ (k & 3)) << 5) + (c4 & 31);      // bf16 elements inside a tile
      if constexpr (NPL == 1) {
        *(u32x2*)(smem + off) = pg_round4(u);
        *(u32x2*)(smem + TILE + off) = pg_round4(v);
      } else {
        u32x2 up[3], vp[3];
        pg_split4(u, up);
        pg_split4(v, vp);
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          *(u32x2*)(smem + q * PL_SZ + off) = up[q];
          *(u32x2*)(smem + q * PL_SZ + TILE + off) = vp[q];
        }
      }
    }
  };

  f32x16 acc[2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
  auto compute_stage = [&](int st) {
    const __bf16* base = smem + st * ST_SZ;
    if constexpr (FS == 1) {
      // ONE fragment set: a k-group's fragments are read, then multiplied — 73 registers instead of 124, so that the three blocks per CU
      // the 48 KB stage was sized for really are resident (six waves per SIMD; with two sets the register file held two blocks).  The wave
      // no longer reads group g + 1 under the MFMAs of group g: the other five waves of its SIMD are what covers that now.  Same MFMAs in
      // the same order: bit-identical.  Same-box A/B of the iteration, eight interleaved runs each: 2.7013 -> 2.6795 ms (median, -0.8 %)
#pragma unroll
      for (int g = 0; g < BK / 16; ++g) {
        bf16x8 a1[NPL], b1[NPL][2];
#pragma unroll
        for (int q = 0; q < NPL; ++q) {
          a1[q] = pg_tr_frag(base + q * PL_SZ, wm, 16 * g, lane);
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) b1[q][nt] = pg_tr_frag(base + q * PL_SZ + TILE, wn + 32 * nt, 16 * g, lane);
        }
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {      // smallest terms first (the order of vf_conv.hip's mode 3)
          if constexpr (NPL == 3) {
            acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[1], b1[1][nt], acc[nt], 0, 0, 0);
            acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[0], b1[2][nt], acc[nt], 0, 0, 0);
            acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[2], b1[0][nt], acc[nt], 0, 0, 0);
            acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[0], b1[1][nt], acc[nt], 0, 0, 0);
            acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[1], b1[0][nt], acc[nt], 0, 0, 0);
          }
          acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[0], b1[0][nt], acc[nt], 0, 0, 0);
        }
      }
    } else {
    bf16x8 a[2][NPL], b[2][NPL][2];
    auto read_frag = [&](int g, int set) {
#pragma unroll
      for (int q = 0; q < NPL; ++q) {
        a[set][q] = pg_tr_frag(base + q * PL_SZ, wm, 16 * g, lane);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) b[set][q][nt] = pg_tr_frag(base + q * PL_SZ + TILE, wn + 32 * nt, 16 * g, lane);
      }
    };
    read_frag(0, 0);
#pragma unroll
    for (int g = 0; g < BK / 16; ++g) {
      const int cs = g & 1;
      if (g + 1 < BK / 16) read_frag(g + 1, cs ^ 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {      // smallest terms first (the order of vf_conv.hip's mode 3)
        if constexpr (NPL == 3) {
          acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[cs][1], b[cs][1][nt], acc[nt], 0, 0, 0);
          acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[cs][0], b[cs][2][nt], acc[nt], 0, 0, 0);
          acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[cs][2], b[cs][0][nt], acc[nt], 0, 0, 0);
          acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[cs][0], b[cs][1][nt], acc[nt], 0, 0, 0);
          acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[cs][1], b[cs][0][nt], acc[nt], 0, 0, 0);
        }
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[cs][0], b[cs][0][nt], acc[nt], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    }
  };

  if (fed32) {
    for (int kt = kt0; kt < kt1; ++kt) {
      stage_fp32(kt);
      __syncthreads();
      compute_stage(0);
      __syncthreads();
    }
  } else if constexpr (NST == 3) {
    // ---- K loop: stages kt and kt + 1 are in flight when stage kt is waited for; every wave issues 6 DMAs per stage, live or not
    dma_stage(kt0, 0, kt0 < kt1);
    dma_stage(kt0 + 1, 1, kt0 + 1 < kt1);
    int st = 0;
    for (int kt = kt0; kt < kt1; ++kt) {
      // the 6 newest DMAs of this wave (stage kt + 1) may stay outstanding; after the barrier everybody's stage kt has landed
      // and nobody still reads the buffer stage kt + 2 goes to (it held stage kt - 1)
      if constexpr (NPL == 3) asm volatile("s_waitcnt vmcnt(6)\n\ts_barrier" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(2)\n\ts_barrier" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      const int st2 = st >= 1 ? st - 1 : 2;                        // (st + 2) % 3
      dma_stage(kt + 2, st2, kt + 2 < kt1);
      __builtin_amdgcn_sched_barrier(0);
      compute_stage(st);
      st = st == 2 ? 0 : st + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // the dead prefetches (zeros) have landed before the block ends
  } else {
    // ---- ONE stage (48 KB: three blocks per CU): a block does not overlap its own DMAs and MFMAs, its CU-mates do
    for (int kt = kt0; kt < kt1; ++kt) {
      dma_stage(kt, 0, true);
      asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      compute_stage(0);
      asm volatile("s_barrier" ::: "memory");
    }
  }

  // ---- epilogue: C/D layout col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
  const int lr = lane & 31, lh = lane >> 5;
  const int64_t total = (int64_t)p.Nu * Ncols;
  float* out = p.out + (p.ksplit > 1 ? (int64_t)ks * total : 0);
  const bool acc_old = p.ksplit == 1 && p.beta != 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int n = n0 + wm + (r & 3) + 8 * (r >> 2) + 4 * lh;
    if (n >= p.Nu) continue;                          // (only the fp32-fed layers have a ragged last row tile)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const int64_t idx = (int64_t)n * Ncols + j0 + wn + nt * 32 + lr;
      float v = acc[nt][r];
      if (acc_old) v += p.beta * out[idx];
      out[idx] = v;
    }
  }
}

int vf_internal_pwgrad_group(vf_ctx* ctx, const VfPWGradGroup& G, int blocks, const char* name, double flops) {
  static const int env_nst = getenv("VF_PWG_STAGES") ? atoi(getenv("VF_PWG_STAGES")) : 1;   // 1: 48 KB, three blocks per CU (measured 34.7 vs 42.9 us per 4.3 GFLOP layer)
  static const int env_fs = getenv("VF_PWG_FS") ? atoi(getenv("VF_PWG_FS")) : 1;            // fragment sets per wave: 1 (three blocks per CU fit the register file too), 2 (round 2-4)
  if (ctx->mfma_bf16 == 1) {      // one rounded plane per operand: 16 KB stages
    VF_LAUNCH_TIMED(ctx, "pwgrad_group_128x128x32_bf16", flops, 0.0, (k_pwgrad_group<1, 1, 2>), dim3((unsigned)blocks), dim3(512), G);
    VF_LAUNCH_CHECK();
    return 0;
  }
  if (env_nst == 1 && env_fs != 2) VF_LAUNCH_TIMED(ctx, name, flops, 0.0, (k_pwgrad_group<1, 3, 1>), dim3((unsigned)blocks), dim3(512), G);
  else if (env_nst == 1) VF_LAUNCH_TIMED(ctx, name, flops, 0.0, (k_pwgrad_group<1, 3, 2>), dim3((unsigned)blocks), dim3(512), G);
  else VF_LAUNCH_TIMED(ctx, name, flops, 0.0, (k_pwgrad_group<3, 3, 2>), dim3((unsigned)blocks), dim3(512), G);
  VF_LAUNCH_CHECK();
  return 0;
}

// ================================================================================================ host
// which kernel serves a planes pass is a tiling decision; the two patch-fed forms can be switched per process (A/B runs, and the tests
// that compare them with k_pconv_dma bit for bit): defaults from VF_PG_GPATCH / VF_PG_PATCH, vf_pconv_set_routing overrides
static int g_gather_patch = getenv("VF_PG_GPATCH") ? atoi(getenv("VF_PG_GPATCH")) : 1;     // 0 off, 1 on
static int g_scatter_patch = getenv("VF_PG_PATCH") ? atoi(getenv("VF_PG_PATCH")) : 1;      // 0 off, 1 auto, 2 / 4 classes per block
VF_API int vf_pconv_set_routing(int gather_patch, int scatter_patch) {
  if (gather_patch >= 0) g_gather_patch = gather_patch;
  if (scatter_patch >= 0) g_scatter_patch = scatter_patch;
  return 0;
}

static inline bool pg_aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

// shared with vf_conv.hip: the split-K combine (plain and with BatchNorm statistics)
int vf_internal_slab_reduce(vf_ctx* ctx, const float* slab, float* dst, const float* bias, int64_t total, int N, int ksplit, int act,
                            float slope, const float* dmask, int dact, float dslope, const VfBnSt* st, int st_groups);
bool vf_internal_slab_st_ok(int64_t total, int N, int groups, int rows_cap, int* blocks_per_group);

// what the shapes must satisfy for the planes GEMM (callers fall back to vf_conv.hip's kernels otherwise)
static bool pg_shape_ok(int B, int Hl, int Wl, int C, int N) {
  return C % 32 == 0 && N >= 32 && N % 4 == 0 && (int64_t)B * Hl * Wl > 64 && vf_is_pow2(Hl) && vf_is_pow2(Wl);
}

static int launch_pconv(vf_ctx* ctx, PGemm& g, int ntaps, const char* what) {
  const int zpar = g.parity ? 4 : 1;
  static const int env_ch = getenv("VF_PG_CH") ? atoi(getenv("VF_PG_CH")) : 0;
  const int ch = (g.C % 64 == 0 && env_ch != 32) ? 64 : 32;      // channels per K step: whole 128-byte lines where possible
  g.nchunks = g.C / ch;
  // tile: 128x64 when that still fills the chip twice over, else 64x64; split-K over channel chunks for the small grids
  struct Tile { int bm, bn; };
  static const int env_tile = getenv("VF_PG_TILE") ? atoi(getenv("VF_PG_TILE")) : 0;
  static const int env_dma0 = getenv("VF_PG_DMA") ? atoi(getenv("VF_PG_DMA")) : 1;
  const Tile cand[2] = {{128, 64}, {64, 64}};
  int pick = 1;
  if (env_tile == 128) pick = 0;
  else if (env_tile == 0 && vf_cdiv(g.M, 128) * vf_cdiv(g.N, 64) * zpar >= 1024) pick = 0;
  // the LDS-DMA kernel: 64-channel stages, whole tiles (its row decode has no ragged edge), 128x64 tiles of 8 waves by default
  bool use_dma = env_dma0 && ch == 64 && g.N % 64 == 0;
  if (use_dma) {
    if (env_tile == 0) pick = (g.M % 128 == 0) ? 0 : 1;
    if (g.M % cand[pick].bm != 0) use_dma = false;
  }
  const Tile t = cand[pick];
  const int gm = (int)vf_cdiv(g.M, t.bm), gn = (int)vf_cdiv(g.N, t.bn);
  const int64_t blocks = (int64_t)gm * gn * zpar;
  int ksplit = 1;
  static const int env_split = getenv("VF_PG_SPLIT_BLOCKS") ? atoi(getenv("VF_PG_SPLIT_BLOCKS")) : 340;   // a grid of 256 tiles (one per CU) is left whole: measured 27.6 vs 39.2 us (E4 data-gradient) against splitting it in two
  if (blocks < env_split * 3 / 4 && g.nchunks >= 2) {
    ksplit = (int)std::min<int64_t>(g.nchunks, vf_cdiv(env_split, blocks));
    const size_t slab_bytes = (size_t)g.out_elems * sizeof(float);
    while (ksplit > 1 && (size_t)ksplit * slab_bytes > vf_ws_avail(ctx)) --ksplit;
    while (ksplit > 1 && g.nchunks % ksplit != 0) --ksplit;      // equal K ranges: every tile runs the same number of steps
  }
  g.ksplit = ksplit;
  g.slab = ksplit > 1 ? (float*)vf_ws_ptr(ctx) : nullptr;
  g.gm = gm; g.gn = gn; g.gz = zpar * ksplit;
  static const int env_dbg = getenv("VF_PG_DBG") ? atoi(getenv("VF_PG_DBG")) : 0;
  g.dbg = env_dbg;
  // ---- BatchNorm statistics attachment (vf_bn_fuse_next_*): same contract as vf_conv.hip's launch_igemm
  g.st.mode = 0;
  bool slab_st = false;
  int st_groups = 1;
  if (ctx->bnf.mode) {
    const int groups = ctx->bnf_groups;
    VfBnSt st = ctx->bnf;
    bool fused = false;
    int bpg = 0;
    if (ksplit == 1) {
      if (g.M % groups == 0 && (g.M / groups) % t.bm == 0 && (int64_t)(gm / groups) * zpar <= ctx->bnf_rows_cap) {
        st.tiles_per_group = gm / groups;
        st.zpar = zpar;
        st.rows_per_group = (gm / groups) * zpar;
        g.st = st;
        fused = true;
      }
    } else if (vf_internal_slab_st_ok(g.out_elems, g.N, groups, ctx->bnf_rows_cap, &bpg)) {
      st.tiles_per_group = bpg;
      st.zpar = 1;
      st.rows_per_group = bpg;
      g.st = st;
      slab_st = true;
      st_groups = groups;
      fused = true;
    }
    if (fused) {
      ctx->bnf_result_rows = st.rows_per_group;
      if (st.mode == 2) {
        g.dmask = ctx->bnf_yact;
        g.dact = ctx->bnf_act;
        g.dslope = ctx->bnf_slope;
      }
    }
    ctx->bnf.mode = 0;
  }
  const bool one_plane = ctx->mfma_bf16 == 1;
  VF_REQUIRE(!one_plane || use_dma, "vf_pconv: in the bf16-operand mode the planes path serves whole 64-channel / 64-row tiles only "
             "(vf_pconv_supported_in_mode)");
  if (use_dma && one_plane) {
    // one rounded plane per operand: 128x64 stages of 24 KB (two stages: 48 KB, three blocks per CU), 64x64 of 16 KB
    const unsigned nt = (unsigned)(gm * gn * zpar * ksplit);
    char dname[64];
    snprintf(dname, sizeof(dname), "pconv_dma_%dx%dx64_%s_bf16", t.bm, t.bn, ntaps == 16 ? "t16" : "t4");
    const double dfl = 2.0 * (double)g.M * g.N * (double)ntaps * g.C * zpar;
    // algorithmic bytes: operand planes and weight planes once, the output (or its split-K slabs), and what the epilogue reads
    // (derivative mask, BatchNorm input) — bench.py prices a kernel against the LONGER of its two floors
    const double dby = (double)g.a_bytes + (double)g.w_bytes + 4.0 * (double)g.out_elems * (ksplit > 1 ? ksplit : 1) +
                       (g.dmask ? (g.dbits ? 0.125 : 4.0) * (double)g.out_elems : 0.0) + (g.st.mode == 2 ? 4.0 * (double)g.out_elems : 0.0);
    if (t.bm == 128) {
      if (ntaps == 16) VF_LAUNCH_TIMED(ctx, dname, dfl, dby, (k_pconv_dma<128, 64, 16, 2, 1>), dim3(nt), dim3(512), g);
      else VF_LAUNCH_TIMED(ctx, dname, dfl, dby, (k_pconv_dma<128, 64, 4, 2, 1>), dim3(nt), dim3(512), g);
    } else {
      if (ntaps == 16) VF_LAUNCH_TIMED(ctx, dname, dfl, dby, (k_pconv_dma<64, 64, 16, 2, 1>), dim3(nt), dim3(256), g);
      else VF_LAUNCH_TIMED(ctx, dname, dfl, dby, (k_pconv_dma<64, 64, 4, 2, 1>), dim3(nt), dim3(256), g);
    }
    VF_LAUNCH_CHECK();
    if (ksplit > 1) {
      VfProf prof(ctx, slab_st ? "slab_reduce_pconv_bnstats" : "slab_reduce_pconv", 0.0, 4.0 * (double)g.out_elems * (ksplit + 1));
      return vf_internal_slab_reduce(ctx, g.slab, g.Y, g.bias, g.out_elems, g.N, ksplit, g.act, g.slope, g.dmask, g.dact, g.dslope,
                                     slab_st ? &g.st : nullptr, st_groups);
    }
    return 0;
  }
  static const int env_dma = getenv("VF_PG_DMA") ? atoi(getenv("VF_PG_DMA")) : 1;
  if (use_dma) {
    const unsigned nt = (unsigned)(gm * gn * zpar * ksplit);
    char dname[64];
    snprintf(dname, sizeof(dname), "pconv_dma_%dx%dx64_%s", t.bm, t.bn, ntaps == 16 ? "t16" : "t4");
    const double dfl = 2.0 * (double)g.M * g.N * (double)ntaps * g.C * zpar;
    // algorithmic bytes: operand planes and weight planes once, the output (or its split-K slabs), and what the epilogue reads
    // (derivative mask, BatchNorm input) — bench.py prices a kernel against the LONGER of its two floors
    const double dby = (double)g.a_bytes + (double)g.w_bytes + 4.0 * (double)g.out_elems * (ksplit > 1 ? ksplit : 1) +
                       (g.dmask ? (g.dbits ? 0.125 : 4.0) * (double)g.out_elems : 0.0) + (g.st.mode == 2 ? 4.0 * (double)g.out_elems : 0.0);
    // a grid of at least two tiles per CU runs the single-stage variant, two blocks per CU (measured, scripts/bench_pconv.py:
    // E2 transposed pass 79 -> 67 us, E2 gather 68 -> 62, E3 transposed 31.2 -> 28.8, netD's first layer at 2B 58 -> 53;
    // with one tile per CU the second stage is what is needed instead: E3 gather 29.7 vs 39.8 single-stage)
    // transposed passes on grids of at least 8 x 16: the patch kernel (k_pconv_patch_tr).  VF_PG_PATCH: 0 off, 2 / 4 = that many parity
    // classes per block, 1 (default) = 4 where that still gives two rounds of blocks, else 2
    const int env_patch = g_scatter_patch;
    if (env_patch && ntaps == 4 && g.parity && g.N % 64 == 0 && ksplit == 1 && t.bm == 128 && g.Wi % 16 == 0 && g.Hi % 8 == 0 &&
        (g.act == VF_ACT_NONE || g.act == VF_ACT_LRELU || g.act == VF_ACT_RELU) &&
        g.out_elems * 4 < ((int64_t)1 << 31)) {       // (its epilogue addresses the output through 32-bit buffer offsets)
      const unsigned tiles = (unsigned)(g.M / 128), slices = (unsigned)(g.N / 64);
      const int cpb = env_patch == 2 || env_patch == 4 ? env_patch : (tiles * slices >= 512 ? 4 : 2);
      snprintf(dname, sizeof(dname), "pconv_patch_128x64_t4_c%d", cpb);
      const dim3 pgrid(tiles * slices * (cpb == 2 ? 2u : 1u));
      if (cpb == 4) VF_LAUNCH_TIMED(ctx, dname, dfl, dby, k_pconv_patch_tr<4>, pgrid, dim3(512), g);
      else VF_LAUNCH_TIMED(ctx, dname, dfl, dby, k_pconv_patch_tr<2>, pgrid, dim3(512), g);
      VF_LAUNCH_CHECK();
      return 0;
    }
    // gather passes (conv forward, full-conv data-gradient) on whole 128-row tiles: the patch kernel (k_pconv_patch_g).  VF_PG_GPATCH=0
    // keeps k_pconv_dma's tap-by-tap stages
    const int env_gpatch = g_gather_patch;
    if (env_gpatch && ntaps == 16 && !g.parity && g.N % 64 == 0 && t.bm == 128 && g.sy == 2 && g.sx == 2) {
      const int Ho = 1 << g.lgMh, Wo = 1 << g.lgMw;
      const int mode = (Wo % 16 == 0 && Ho % 8 == 0) ? 0 : (Ho == 8 && Wo == 8) ? 1 : (Ho == 4 && Wo == 4) ? 2 : -1;
      if (mode >= 0) {
        snprintf(dname, sizeof(dname), "pconv_patchg_128x64_t16_m%d", mode);
        // timing experiments only (VF_PG_STAMPS=<file>): the stamped build of the 8 x 16-tile form; every 16th launch is synchronised and dumped
        static const char* stamp_file = getenv("VF_PG_STAMPS");
        if (stamp_file && mode == 0 && ksplit == 1) {
          static long long* stamp_buf = nullptr;
          static unsigned stamp_count = 0;
          constexpr size_t NST = (size_t)64 * 8 * 16 * 4;
          if (!stamp_buf) VF_CHECK_HIP(hipHostMalloc((void**)&stamp_buf, NST * sizeof(long long), hipHostMallocDefault));
          g.stamps = stamp_buf;
          const bool dump = (++stamp_count % 16) == 0;
          if (dump) {
            VF_CHECK_HIP(hipStreamSynchronize(ctx->stream));
            memset(stamp_buf, 0, NST * sizeof(long long));
          }
          hipLaunchKernelGGL((k_pconv_patch_g<0, true>), dim3(nt), dim3(512), 0, ctx->stream, g);
          VF_LAUNCH_CHECK();
          if (dump) {
            VF_CHECK_HIP(hipStreamSynchronize(ctx->stream));
            if (FILE* f = fopen(stamp_file, "ab")) {
              const int hdr[4] = {(int)nt, g.M, g.N, g.C};
              fwrite(hdr, sizeof(hdr), 1, f);
              fwrite(stamp_buf, sizeof(long long), NST, f);
              fclose(f);
            }
          }
          return 0;
        }
        if (mode == 0) VF_LAUNCH_TIMED(ctx, dname, dfl, dby, k_pconv_patch_g<0>, dim3(nt), dim3(512), g);
        else if (mode == 1) VF_LAUNCH_TIMED(ctx, dname, dfl, dby, k_pconv_patch_g<1>, dim3(nt), dim3(512), g);
        else VF_LAUNCH_TIMED(ctx, dname, dfl, dby, k_pconv_patch_g<2>, dim3(nt), dim3(512), g);
        VF_LAUNCH_CHECK();
        if (ksplit > 1) {
          VfProf prof(ctx, slab_st ? "slab_reduce_pconv_bnstats" : "slab_reduce_pconv", 0.0, 4.0 * (double)g.out_elems * (ksplit + 1));
          return vf_internal_slab_reduce(ctx, g.slab, g.Y, g.bias, g.out_elems, g.N, ksplit, g.act, g.slope, g.dmask, g.dact, g.dslope,
                                         slab_st ? &g.st : nullptr, st_groups);
        }
        return 0;
      }
    }
    // transposed passes whose low-resolution grid is a whole 8 x 8 / 4 x 4 map (what k_pconv_patch_tr's 8 x 16 regions do not cover): the
    // patch kernel's TR form, one output-parity class per block as here
    // (grids of one tile per CU only: from two rounds of blocks the single-stage k_pconv_dma, two blocks per CU, is faster —
    //  netD's third conv at 2B, 512 blocks: 45.6 us against 46.5, with its derivative mask 51.9 against 59.7)
    if (env_patch && ntaps == 4 && g.parity && g.N % 64 == 0 && t.bm == 128 && nt < 512 &&
        ((g.Hi == 8 && g.Wi == 8) || (g.Hi == 4 && g.Wi == 4))) {
      const int mode = g.Hi == 8 ? 1 : 2;
      snprintf(dname, sizeof(dname), "pconv_patchg_128x64_t4_m%d", mode);
      if (mode == 1) VF_LAUNCH_TIMED(ctx, dname, dfl, dby, (k_pconv_patch_g<1, false, true>), dim3(nt), dim3(512), g);
      else VF_LAUNCH_TIMED(ctx, dname, dfl, dby, (k_pconv_patch_g<2, false, true>), dim3(nt), dim3(512), g);
      VF_LAUNCH_CHECK();
      if (ksplit > 1) {
        VfProf prof(ctx, slab_st ? "slab_reduce_pconv_bnstats" : "slab_reduce_pconv", 0.0, 4.0 * (double)g.out_elems * (ksplit + 1));
        return vf_internal_slab_reduce(ctx, g.slab, g.Y, g.bias, g.out_elems, g.N, ksplit, g.act, g.slope, g.dmask, g.dact, g.dslope,
                                       slab_st ? &g.st : nullptr, st_groups);
      }
      return 0;
    }
    static const int env_nbuf = getenv("VF_PG_NBUF") ? atoi(getenv("VF_PG_NBUF")) : 0;
    const bool one_stage = env_nbuf ? env_nbuf == 1 : (t.bm == 128 && nt >= 512);
    if (one_stage) snprintf(dname, sizeof(dname), "pconv_dma_%dx%dx64_%s_1stage", t.bm, t.bn, ntaps == 16 ? "t16" : "t4");
    if (t.bm == 128) {
      if (ntaps == 16) {
        if (one_stage) VF_LAUNCH_TIMED(ctx, dname, dfl, dby, (k_pconv_dma<128, 64, 16, 1>), dim3(nt), dim3(512), g);
        else VF_LAUNCH_TIMED(ctx, dname, dfl, dby, (k_pconv_dma<128, 64, 16, 2>), dim3(nt), dim3(512), g);
      } else {
        if (one_stage) VF_LAUNCH_TIMED(ctx, dname, dfl, dby, (k_pconv_dma<128, 64, 4, 1>), dim3(nt), dim3(512), g);
        else VF_LAUNCH_TIMED(ctx, dname, dfl, dby, (k_pconv_dma<128, 64, 4, 2>), dim3(nt), dim3(512), g);
      }
    } else {
      if (ntaps == 16) {
        if (one_stage) VF_LAUNCH_TIMED(ctx, dname, dfl, dby, (k_pconv_dma<64, 64, 16, 1>), dim3(nt), dim3(256), g);
        else VF_LAUNCH_TIMED(ctx, dname, dfl, dby, (k_pconv_dma<64, 64, 16, 2>), dim3(nt), dim3(256), g);
      } else {
        if (one_stage) VF_LAUNCH_TIMED(ctx, dname, dfl, dby, (k_pconv_dma<64, 64, 4, 1>), dim3(nt), dim3(256), g);
        else VF_LAUNCH_TIMED(ctx, dname, dfl, dby, (k_pconv_dma<64, 64, 4, 2>), dim3(nt), dim3(256), g);
      }
    }
    VF_LAUNCH_CHECK();
    if (ksplit > 1) {
      VfProf prof(ctx, slab_st ? "slab_reduce_pconv_bnstats" : "slab_reduce_pconv", 0.0, 4.0 * (double)g.out_elems * (ksplit + 1));
      return vf_internal_slab_reduce(ctx, g.slab, g.Y, g.bias, g.out_elems, g.N, ksplit, g.act, g.slope, g.dmask, g.dact, g.dslope,
                                     slab_st ? &g.st : nullptr, st_groups);
    }
    return 0;
  }
  static const int env_pair = getenv("VF_PG_PAIR") ? atoi(getenv("VF_PG_PAIR")) : 0;
  const bool pair = env_pair && ch == 64 && t.bm == 64;
  const unsigned ntiles = (unsigned)(gm * gn * zpar * ksplit);
  dim3 grid(pair ? (ntiles + 1) / 2 : ntiles), block(256);
  char name[64];
  snprintf(name, sizeof(name), "pconv_%dx%dx%d_%s%s", t.bm, t.bn, ch, ntaps == 16 ? "t16" : "t4", pair ? "_pair" : "");
  const double fl = 2.0 * (double)g.M * g.N * (double)ntaps * g.C * zpar;
#define PG_LAUNCH(BM_, BN_, WM_, WN_)                                                                                      \
  do {                                                                                                                     \
    if (pair) {                                                                                                            \
      if (ntaps == 16) VF_LAUNCH_TIMED(ctx, name, fl, 0.0, (k_pconv<BM_, BN_, WM_, WN_, 16, 64, true>), grid, dim3(512), g); \
      else VF_LAUNCH_TIMED(ctx, name, fl, 0.0, (k_pconv<BM_, BN_, WM_, WN_, 4, 64, true>), grid, dim3(512), g);              \
    } else if (ch == 64) {                                                                                                 \
      if (ntaps == 16) VF_LAUNCH_TIMED(ctx, name, fl, 0.0, (k_pconv<BM_, BN_, WM_, WN_, 16, 64, false>), grid, block, g);    \
      else VF_LAUNCH_TIMED(ctx, name, fl, 0.0, (k_pconv<BM_, BN_, WM_, WN_, 4, 64, false>), grid, block, g);                 \
    } else {                                                                                                               \
      if (ntaps == 16) VF_LAUNCH_TIMED(ctx, name, fl, 0.0, (k_pconv<BM_, BN_, WM_, WN_, 16, 32, false>), grid, block, g);    \
      else VF_LAUNCH_TIMED(ctx, name, fl, 0.0, (k_pconv<BM_, BN_, WM_, WN_, 4, 32, false>), grid, block, g);                 \
    }                                                                                                                      \
  } while (0)
  if (t.bm == 128) PG_LAUNCH(128, 64, 64, 32);
  else PG_LAUNCH(64, 64, 32, 32);
#undef PG_LAUNCH
  VF_LAUNCH_CHECK();
  if (ksplit > 1) {
    VfProf prof(ctx, slab_st ? "slab_reduce_pconv_bnstats" : "slab_reduce_pconv", 0.0, 4.0 * (double)g.out_elems * (ksplit + 1));
    return vf_internal_slab_reduce(ctx, g.slab, g.Y, g.bias, g.out_elems, g.N, ksplit, g.act, g.slope, g.dmask, g.dact, g.dslope,
                                   slab_st ? &g.st : nullptr, st_groups);
  }
  return 0;
}

static int pg_fill_common(vf_ctx* ctx, PGemm& g, const void* a, int64_t a_elems, const void* w, int64_t w_elems, const float* bias, float* y) {
  memset(&g, 0, sizeof(g));
  g.A = a; g.W = w; g.bias = bias; g.Y = y;
  const int npl = ctx->mfma_bf16 == 1 ? 1 : 3;
  VF_REQUIRE(ctx->mfma_bf16 == 3 || ctx->mfma_bf16 == 1, "vf_pconv: the planes path serves product modes 3 (exact split) and 1 (bf16 operands)");
  VF_REQUIRE(a_elems * 2 * npl < ((int64_t)1 << 31) && w_elems * 2 * npl < ((int64_t)1 << 31), "planes exceed the 2 GiB buffer-descriptor range");
  g.a_ps = (unsigned)(a_elems * 2);
  g.w_ps = (unsigned)(w_elems * 2);
  g.a_bytes = npl * g.a_ps;
  g.w_bytes = npl * g.w_ps;
  return 0;
}

// conv-like pass: Y[b,oy,ox,n] = sum_{kh,kw,c} A[b, 2oy-1+kh, 2ox-1+kw, c] * Wp[n][kh][kw][c]      (4x4, stride 2, pad 1)
static int pconv_like_fwd(vf_ctx* ctx, const void* ap, const void* wp, const float* bias, float* Y, int B, int Hi, int Wi, int C,
                          int N, int act, float slope) {
  const int Ho = Hi / 2, Wo = Wi / 2;
  PGemm g;
  if (int rc = pg_fill_common(ctx, g, ap, (int64_t)B * Hi * Wi * C, wp, (int64_t)N * 16 * C, bias, Y)) return rc;
  g.lgMh = vf_ilog2(Ho); g.lgMw = vf_ilog2(Wo);
  g.M = B * Ho * Wo;
  g.Hi = Hi; g.Wi = Wi; g.C = C; g.N = N;
  g.sy = 2; g.oy0 = -1; g.sx = 2; g.ox0 = -1;
  g.lgTW = 2;
  g.kh0 = 0; g.khs = 1; g.kw0 = 0; g.kws = 1;
  g.outH = Ho; g.outW = Wo; g.osy = 1; g.osx = 1;
  g.out_elems = (int64_t)g.M * N;
  g.act = act; g.slope = slope;
  return launch_pconv(ctx, g, 16, "fwd");
}
// transposed pass: Y[b,oh,ow,n] = sum_{kh,kw,c : oh = 2i-1+kh, ow = 2j-1+kw} A[b,i,j,c] * Wp[n][kh][kw][c]; per output parity
// (ph, pw) a 2x2-tap GEMM over the low-res grid: window rows i = my + ph - 1 + th, filter row kh = 3 - ph - 2*th
static int pconv_like_tr(vf_ctx* ctx, const void* ap, const void* wp, const float* bias, float* Y, int B, int Hi, int Wi, int C, int N,
                         int act, float slope, const float* dmask = nullptr, int dact = 0, float dslope = 0.f) {
  PGemm g;
  if (int rc = pg_fill_common(ctx, g, ap, (int64_t)B * Hi * Wi * C, wp, (int64_t)N * 16 * C, bias, Y)) return rc;
  g.lgMh = vf_ilog2(Hi); g.lgMw = vf_ilog2(Wi);
  g.M = B * Hi * Wi;
  g.Hi = Hi; g.Wi = Wi; g.C = C; g.N = N;
  g.sy = 1; g.oy0 = -1; g.sx = 1; g.ox0 = -1;      // (+ph, +pw in the kernel)
  g.lgTW = 1;
  g.khs = -2; g.kws = -2;                          // kh0 = 3 - ph, kw0 = 3 - pw in the kernel
  g.parity = 1;
  g.outH = 2 * Hi; g.outW = 2 * Wi; g.osy = 2; g.osx = 2;
  g.out_elems = (int64_t)B * g.outH * g.outW * N;
  g.act = act; g.slope = slope;
  g.dmask = dmask; g.dact = dact; g.dslope = dslope;
  // one-shot (vf_net.hip): the same mask as sign bits, where the tensor's producer left them and the channel count is whole groups
  g.dbits = (dmask && N % 64 == 0) ? ctx->dmask_bits : nullptr;
  ctx->dmask_bits = nullptr;
  return launch_pconv(ctx, g, 4, "tr");
}

// ------------------------------------------------------------------------------------------------ C ABI
VF_API int vf_planes_split(vf_ctx* ctx, const float* x, void* planes, int64_t n) {
  VF_REQUIRE(n % 4 == 0 && pg_aligned16(x) && ((uintptr_t)planes & 7) == 0, "vf_planes_split: n %% 4 == 0 and aligned buffers");
  const int64_t n4 = n / 4;
  const int nb = (int)std::max<int64_t>(1, std::min<int64_t>(vf_cdiv(n4, 256 * 2), 4096));
  if (ctx->mfma_bf16 == 1) {
    VfProf prof(ctx, "planes_round_bf16", 0.0, 6.0 * (double)n);
    hipLaunchKernelGGL(k_planes_split<1>, dim3(nb), dim3(256), 0, ctx->stream, x, (__bf16*)planes, n4, n);
    VF_LAUNCH_CHECK();
    return 0;
  }
  VfProf prof(ctx, "planes_split", 0.0, 10.0 * (double)n);
  hipLaunchKernelGGL(k_planes_split<3>, dim3(nb), dim3(256), 0, ctx->stream, x, (__bf16*)planes, n4, n);
  VF_LAUNCH_CHECK();
  return 0;
}
VF_API int vf_weight_planes(vf_ctx* ctx, const float* w, void* planes_native, void* planes_transposed, int d0, int d1) {
  VF_REQUIRE(w && planes_native && d0 > 0 && d1 > 0, "vf_weight_planes: bad arguments");
  VfProf prof(ctx, "weight_planes", 0.0, (planes_transposed ? 16.0 : 10.0) * (double)d0 * 16 * d1);
  hipLaunchKernelGGL(k_weight_planes, dim3((unsigned)vf_cdiv(d0, 32), 16, (unsigned)vf_cdiv(d1, 32)), dim3(256), 0, ctx->stream, w,
                     (__bf16*)planes_native, (__bf16*)planes_transposed, d0, d1, (int64_t)d0 * 16 * d1, ctx->mfma_bf16 == 1 ? 1 : 3);
  VF_LAUNCH_CHECK();
  return 0;
}
VF_API int vf_weight_planes_multi(vf_ctx* ctx, const void* desc_dev, int n, int blocks) {
  VF_REQUIRE(desc_dev != nullptr && n > 0 && blocks > 0, "vf_weight_planes_multi: empty table");
  VfProf prof(ctx, "weight_planes_multi", 0.0, 0.0);
  hipLaunchKernelGGL(k_weight_planes_multi, dim3(blocks), dim3(256), 0, ctx->stream, (const VfWpDesc*)desc_dev, n,
                     ctx->mfma_bf16 == 1 ? 1 : 3);
  VF_LAUNCH_CHECK();
  return 0;
}
VF_API int vf_pconv_supported(int B, int H, int W, int Cin, int Cout, int k, int stride, int pad, int transposed) {
  if (k != 4 || stride != 2 || pad != 1) return 0;
  // conv-like passes gather on the H x W grid with channels Cin and produce Cout; transposed ones walk the low-res grid
  return pg_shape_ok(B, transposed ? H : H / 2, transposed ? W : W / 2, Cin, Cout) && H >= 2 && W >= 2 ? 1 : 0;
}
// the same question for a product mode: 3 = vf_pconv_supported; 1 (operands rounded to bf16, ONE plane [1][n]) is served by the
// LDS-DMA kernel alone: whole 64-channel K steps and whole 64 x 64 tiles; other modes: never
VF_API int vf_pconv_supported_in_mode(int mfma_mode, int B, int H, int W, int Cin, int Cout, int k, int stride, int pad, int transposed) {
  if (mfma_mode == 3) return vf_pconv_supported(B, H, W, Cin, Cout, k, stride, pad, transposed);
  if (mfma_mode != 1 || !vf_pconv_supported(B, H, W, Cin, Cout, k, stride, pad, transposed)) return 0;
  const int64_t M = (int64_t)B * (transposed ? H : H / 2) * (transposed ? W : W / 2);
  return (Cin % 64 == 0 && Cout % 64 == 0 && M % 64 == 0) ? 1 : 0;
}
/* conv forward / full-conv data-gradient: gather planes `ap` [B][H][W][Cin], weight planes `wp` [Cout][16][Cin] */
VF_API int vf_pconv_gather(vf_ctx* ctx, const void* ap, const void* wp, const float* bias, float* y, int B, int H, int W, int Cin,
                           int Cout, int act, float slope) {
  VF_REQUIRE(vf_pconv_supported_in_mode(ctx->mfma_bf16, B, H, W, Cin, Cout, 4, 2, 1, 0), "vf_pconv_gather: unsupported shape B=%d %dx%d %d->%d (product mode %d)", B, H, W, Cin, Cout, ctx->mfma_bf16);
  return pconv_like_fwd(ctx, ap, wp, bias, y, B, H, W, Cin, Cout, act, slope);
}
/* conv data-gradient / full-conv forward: low-res planes `ap` [B][H][W][Cin] -> y [B][2H][2W][Cout], weight planes [Cout][16][Cin] */
VF_API int vf_pconv_scatter(vf_ctx* ctx, const void* ap, const void* wp, const float* bias, float* y, int B, int H, int W, int Cin,
                            int Cout, int act, float slope, const float* dmask, int dact, float dslope) {
  VF_REQUIRE(vf_pconv_supported_in_mode(ctx->mfma_bf16, B, H, W, Cin, Cout, 4, 2, 1, 1), "vf_pconv_scatter: unsupported shape B=%d %dx%d %d->%d (product mode %d)", B, H, W, Cin, Cout, ctx->mfma_bf16);
  VF_REQUIRE(!(dmask && bias), "vf_pconv_scatter: the activation-backward epilogue is for data-gradient passes (no bias)");
  return pconv_like_tr(ctx, ap, wp, bias, y, B, H, W, Cin, Cout, act, slope, dmask, dact, dslope);
}
