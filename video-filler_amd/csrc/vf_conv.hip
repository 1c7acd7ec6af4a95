// vf_conv.hip — 4x4 convolution / transposed convolution, forward + data-grad + weight-grad, for gfx950.
//
// Every pass is an implicit GEMM on the matrix cores.  Three product modes (vf_ctx_set_mfma_mode, template parameter MODE):
//   3 (default)  fp32 operands split exactly into three bf16 planes on their way into LDS (truncation by v_perm_b32), six
//                cross terms on v_mfma_f32_32x32x16_bf16, fp32 accumulators: fp32-grade results at 1.1-1.4x mode 0's speed;
//   0            v_mfma_f32_32x32x2_f32 on the fp32 operands (bit for bit a k-ordered fmaf chain);
//   1            operands rounded to bf16, one term (opt-in).
// Activations are NHWC, weights are the reference tensors in channels-last (see include/vf_hip.h), which makes a layer's
// three passes need only two kernels:
//
//   k_igemm  C[m][n] = sum_{tap,c} A(m,tap,c) * Wt(n,tap,c)
//            - conv forward            (reference: THNN SpatialConvolutionMM_updateOutput, train.lua:89)
//            - conv data-grad          (transposed conv, 4 output-parity classes, each a 2x2-tap GEMM)
//            - full-conv forward       (= conv data-grad; THNN SpatialFullConvolution_updateOutput, train.lua:134)
//            - full-conv data-grad     (= conv forward without bias)
//            - the 4x4 -> 1x1 bottleneck conv and 1x1 -> 4x4 full-conv (plain GEMMs, weight-bandwidth bound)
//   k_wgrad / k_wgrad_group  dW[n][tap][c] = sum_p U(p,n) * V(p,tap,c), split over pixels p
//            - conv / full-conv accGradParameters (the two differ only in which tensor is x and which is gy);
//              the group form runs the weight gradients of a whole backward walk as one launch (vf_wgrad_group_*)
//
// Since round 2 the forward / data-grad passes of layers with >= 1024 GEMM rows and >= 32 channels are served by the
// planes-fed kernels of vf_pgemm.hip (operands split once by their producer, LDS-DMA staging), and the 3-channel
// image-side forward by vf_conv_thin.hip; k_igemm keeps the small-row layers (4x4 maps, bottleneck), modes 0 / 1, the
// thin-output passes (V = 0 scalar gather + col2im) and the callers that bring no planes.
//
// k_igemm tiles: 256 threads = 4 waves of 32x32 MFMA tiles (64x64, 64x128 or 128x128 block tiles), BK = 16 (modes 0 / 1)
// or 32 channels per step (mode 3), LDS double-buffered, register-staged global->LDS copies through BUFFER loads with
// hardware range checking (an out-of-range offset returns zeros: padding taps, ragged tile edges and split-K tails need
// neither a branch nor a select), so every load of step k+1 stays in flight across the MFMAs of step k.  Grid.z carries
// output parity and split-K; blocks are remapped so that the tiles sharing an operand panel land on one XCD's L2.
// Split-K partial slabs are combined by k_slab_reduce* in a fixed order (deterministic; no float atomics); that pass — or
// the GEMM epilogue when there is no split — also leaves the per-tile BatchNorm partial sums of the output when the next
// module is a BatchNorm (vf_bn_fuse_next_fwd / _bwd: the separate statistics pass over the tensor disappears).
#include <algorithm>
#include <vector>
#include <cstdlib>

#include "vf_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

// Buffer loads with hardware range checking: an offset past num_records returns 0, so padding taps, ragged
// tile edges and split-K tails need neither a branch nor a select — the loads stay in flight across the MFMAs.
#define VF_OOB 0x80000000u   // byte offset that is always out of range (operands are < 2 GiB)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t vf_rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ f32x4 vf_bload4(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 0));
}
__device__ __forceinline__ float vf_bload1(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, byte_off, 0, 0));
}

// bf16 MFMA operand (32 rows x 16 k, 8 consecutive k per lane) out of a K-MAJOR bf16 tile [k][LD] in LDS, with gfx950's
// transposing read: per 16-lane group ds_read_b64_tr_b16 takes a 4(k) x 16(column) block and hands lane i column i's four
// k values, so tiles whose pieces arrive as "4 consecutive columns at one k" are stored with one 8-byte write and need
// no transposing pass.  Lane l ends up with column col0 + l%32 and k = k0 + 8*(l/32) .. +7, the 32x32x16 operand map.
// LD (bf16 elements) = columns + 32 keeps the four k rows of a block on disjoint bank quarters.  EXEC must be full.
template <int LD>
__device__ __forceinline__ bf16x8 vf_tr_frag(const __bf16* tile, int col0, int k0, int lane) {
  const int grp = lane >> 4, i = lane & 15;
  const __bf16* a = tile + (k0 + 8 * (grp >> 1) + (i >> 2)) * LD + col0 + 16 * (grp & 1) + 4 * (i & 3);
  typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
  const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)a);
  const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(a + 4 * LD));
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

// Exact three-way split of four fp32 values into bf16 planes by TRUNCATION (mode 3): plane q holds the top 16 bits of the
// running residual, the residual loses exactly those bits (v - float(top16(v)) is exact), and after two steps at most 8
// significant bits are left, so hi + mid + lo == v bit for bit.  No conversion instruction is needed: a v_perm_b32
// packs the top halves of two residuals into one bf16x2 word (6 perms + 8 ands + 8 subs per four elements; the
// round-to-nearest form through v_cvt_pk_bf16_f32 cost 30).
struct VfPlanes3 { uint2 p[3]; };
__device__ __forceinline__ VfPlanes3 vf_split3(f32x4 v) {
  VfPlanes3 o;
  float r0 = v[0], r1 = v[1], r2 = v[2], r3 = v[3];
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    const unsigned u0 = __float_as_uint(r0), u1 = __float_as_uint(r1), u2 = __float_as_uint(r2), u3 = __float_as_uint(r3);
    o.p[q].x = __builtin_amdgcn_perm(u1, u0, 0x07060302u);      // (u1 & 0xffff0000) | (u0 >> 16)
    o.p[q].y = __builtin_amdgcn_perm(u3, u2, 0x07060302u);
    if (q < 2) {
      r0 -= __uint_as_float(u0 & 0xffff0000u);
      r1 -= __uint_as_float(u1 & 0xffff0000u);
      r2 -= __uint_as_float(u2 & 0xffff0000u);
      r3 -= __uint_as_float(u3 & 0xffff0000u);
    }
  }
  return o;
}

// XCD-aware block order.  The dispatcher deals consecutive workgroup ids round-robin over the 8 XCDs (each with its
// own 4 MiB L2), so neighbouring tiles — which share input halos (k_igemm) or the whole gathered operand (k_wgrad's
// column tiles) — land on different L2s and every one of them fetches the shared rows from the fabric again
// (measured with FETCH_SIZE: 3-5x the algorithmic bytes).  This bijective remap gives each XCD one contiguous run of
// logical tile ids instead; it only ever changes speed, never results.
__device__ __forceinline__ int vf_xcd_remap(int h, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = h & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (h >> 3);
}

// ------------------------------------------------------------------------------------------------
struct IGemm {
  const float* A;     // gathered activations [B][Hi][Wi][C]
  const float* Wt;    // weights
  const float* bias;  // [N] or null
  float* Y;           // output [B][outH][outW][N]
  float* slab;        // split-K partials, ksplit x (output-sized) ; null when ksplit == 1
  int64_t out_elems;  // B*outH*outW*N
  unsigned a_bytes, w_bytes;  // operand sizes for the buffer descriptors (< 2 GiB each)
  int M, lgMh, lgMw;  // GEMM rows = B << (lgMh+lgMw), decoded as (b, my, mx)
  int Hi, Wi, C;
  int N;
  int TH, TW;                      // taps walked by the K loop
  int sy, ty, oy0, sx, tx, ox0;    // iy = my*sy + th*ty + oy0 (+ph in parity mode)
  int kh0, khs, kw0, kws;          // filter tap (kh, kw) = (kh0 + th*khs, kw0 + tw*kws)
  int wsN, wsC, wsTap;             // weight offset = n*wsN + (kh*4+kw)*wsTap + c*wsC   (all tensors < 2^31 elements)
  int outH, outW, osy, ooy0, osx, oox0;  // output pixel = (my*osy + ooy0, mx*osx + oox0)
  int parity;                      // 1: z&3 = (ph<<1)|pw shifts oy0/ox0/ooy0/oox0 and selects kh0/kw0
  int gm, gn, gz;                  // logical grid (the launch is 1-D, remapped per XCD)
  int ksplit, nk, nq;              // K steps (of 32), number of splits, 16-wide chunks on the vector path
  int klin;                        // K order of the vector path: 1 = tap outer, channel chunk inner (see next_chunk)
  int dbg;                         // ablation switch (timing experiments only; wrong results): 1 = no operand reload
  long long* stamps;               // timing experiments only: 8 stamp slots per block (VF_IGEMM_STAMPS=<file>)
  int act;
  float slope;
  const float* dmask;              // optional: out = act'(dmask[same index]) * out — the backward of the in-place activation
  int dact;                        // that produced this pass's output tensor's forward twin (nn.LeakyReLU:updateGradInput)
  float dslope;
  VfBnSt st;                       // BatchNorm statistics of the output as a by-product of the epilogue (mode 0: none)
};

// Per-channel partial sums of one block's output tile -> one partial row (see VfBnSt).  Lane l of a wave holds column
// l % 32 of its 32-wide fragments and 16 rows per fragment: the per-lane sums over those rows are combined across the two
// lane halves with a shuffle, across the waves stacked in M through LDS in a fixed order (deterministic), and the block
// writes doubles.  `red` = 2 * WAVES_M * BN floats of LDS that nothing else uses any more.
template <int NT, int WAVES_M, int BN>
__device__ __forceinline__ void vf_bn_tile_partials(const VfBnSt& st, float (&s1)[NT], float (&s2)[NT], float* red, int wave_m,
                                                    int wn, int lane, int tid, int n0, int N, int bx, int pz) {
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

// V = 2: 16-byte loads for A and B (C % 16 == 0);  V = 1: 16-byte A, scalar B (k-major B with N % 4 != 0);
// V = 0: scalar loads with a flattened (tap, c) K index (first/last layers: C = 3, 12, 27 ...).
// BF = 1: bf16-operand mode (opt-in, vf_ctx_set_mfma_mode): the fp32 pieces are rounded to bf16 (RNE) on their way
// into LDS (row-major [row][k], 80-byte rows: conflict-free ds_read_b128; k-major operands: vf_tr_frag) and the
// products run on v_mfma_f32_32x32x16_bf16 with fp32 accumulation.
// BF = 3: fp32-grade products on the bf16 pipe: every operand element is split EXACTLY into three bf16 planes
// (x = hi + mid + lo: 3 x 8 = the 24 significand bits of an fp32) and the six largest cross terms are accumulated —
// what is dropped (mid*lo, lo*mid, lo*lo) is below 2^-24 of the product, the size of one fp32 rounding.
// Everything outside the LDS tile (addresses, loads, epilogue, split-K) is shared with the fp32 path.
// KLIN: K walked in memory order (tap outer, channel chunk inner) — the 1x1-output bottleneck layers (see next_chunk);
// a template parameter so that no other instantiation carries its counters.
// -DVF_IGEMM_SPY (timing experiments; scripts/probe/spy_report.py): thread 0 of block 7 stamps the shader clock between
// the phases of each of its first 32 K steps of the single-buffered mode-3 loop, into rows 4096.. of the VF_IGEMM_STAMPS
// buffer.  Compiled out otherwise.
#ifdef VF_IGEMM_SPY
#define VF_SPY(slot)                                                                                             \
  do {                                                                                                           \
    if (p.stamps && blockIdx.x == 7 && tid == 0 && kt - kt0 < 32)                                                \
      p.stamps[8 * 4096 + (kt - kt0) * 8 + (slot)] = (long long)__builtin_readcyclecounter();                    \
  } while (0)
#else
#define VF_SPY(slot) do { } while (0)
#endif
template <int I> struct VfIC { static constexpr int value = I; };
// DB (mode 3, V = 2 only): double-buffered LDS with the split and the LDS writes of step k+1 issued between the MFMAs of
// step k and the global loads running two steps ahead in a second register set — for launches whose grid leaves at
// most two blocks per CU anyway (the 66 KB this needs costs no occupancy there; see launch_igemm).
template <int BM, int BN, int WM, int WN, bool BKM, int V, int BF = 0, bool KLIN = false, bool DB = false>
__global__ __launch_bounds__(256) void k_igemm(const IGemm p) {
  // One K step = 32 = two 16-wide chunks; each chunk has its own (tap, c0), so any C % 16 == 0 vectorises.
  constexpr int BK = 32, LDA = BK + 4;
  constexpr int LDB = BKM ? (BN + 4) : BK + 4;
  constexpr int MT = WM / 32, NT = WN / 32;
  constexpr int WAVES_N = BN / WN;
  static_assert((BM / WM) * WAVES_N == 4, "4 waves per block");
  constexpr int A_CH = (BM * 8 + 255) / 256;
  constexpr int B_CH = (BN * 8 + 255) / 256;
  constexpr int A_SZ = BM * LDA;
  constexpr int B_SZ = BKM ? BK * LDB : BN * LDB;
  // Row-major bf16 tiles (BF): 64-byte rows, NO padding, the four 16-byte k-octets of a row XOR-swizzled by
  // (row >> 2) & 3.  ds_read_b128 serves {0-3,12-15,20-27}-style 16-lane groups on a 256-byte bank row: the four
  // row-quads of a group get the four different swizzles, so their octets land on 16 distinct slots; ds_write_b64
  // serves 16 contiguous lanes (= two whole rows) on a 128-byte window, which two unpadded 64-byte rows fill exactly.
  // (The padded linear layout — 80-byte rows — kept the reads conflict-free but made every plane write 2-way:
  //  SQ_LDS_BANK_CONFLICT was a third of the LDS pipe's active cycles.)
  constexpr int LDH = BK;                              // bf16 elements per LDS row (BF)
  auto frag_off = [](int row, int octet) { return row * LDH + ((octet ^ ((row >> 2) & 3)) << 3); };        // 8 k of `row`
  auto piece_off = [](int row, int kq) { return row * LDH + (((kq >> 1) ^ ((row >> 2) & 3)) << 3) + ((kq & 1) << 2); };
  constexpr int LDN = BN + 32;                         // k-major bf16 B tile (BF && BKM): [k][LDN], transposing reads
  constexpr int AH_SZ = BM * LDH, BH_SZ = BKM ? BK * LDN : BN * LDH;    // bf16 elements per buffer (BF)
  constexpr int NP = BF > 0 ? BF : 1;                  // bf16 planes per operand
  static_assert(!DB || (BF == 3 && V == 2), "the double-buffered schedule is built for mode 3's vector path");
  constexpr int NBUF = (BF == 3 && !DB) ? 1 : 2;       // three planes: single-buffered LDS (capacity), loads still run ahead
  constexpr int PL_SZ = AH_SZ + BH_SZ;                 // one plane of (A, B), bf16 elements
  constexpr int SMEM_F = BF ? (NBUF * NP * PL_SZ + 1) / 2 : 2 * (A_SZ + B_SZ);
  __shared__ __attribute__((aligned(16))) float smem[SMEM_F];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = (wave / WAVES_N) * WM, wn = (wave % WAVES_N) * WN;
  if (p.stamps && tid == 0) { p.stamps[8 * blockIdx.x + 0] = wall_clock64(); p.stamps[8 * blockIdx.x + 4] = (long long)__builtin_readcyclecounter(); }
  // logical tile id: output parity fastest (the 4 parity classes of a transposed pass read the same input rows),
  // then M tiles (halo sharing), then N tiles, then split-K
  int lid = vf_xcd_remap(blockIdx.x, p.gm * p.gn * p.gz);
  int ph = 0, pw = 0;
  if (p.parity) {
    ph = (lid >> 1) & 1;
    pw = lid & 1;
    lid >>= 2;
  }
  const int bx = lid % p.gm, byz = lid / p.gm;
  const int m0 = bx * BM, n0 = (byz % p.gn) * BN;
  const int z = byz / p.gn;
  const int ks = z;
  const int steps = (p.nk + p.ksplit - 1) / p.ksplit;
  const int kt0 = ks * steps;
  const int kt1 = min(p.nk, kt0 + steps);
  const int oy0 = p.oy0 + ph, ox0 = p.ox0 + pw;
  const int kh0 = p.parity ? (1 - ph) : p.kh0, kw0 = p.parity ? (1 - pw) : p.kw0;
  const int ooy0 = p.ooy0 + ph, oox0 = p.oox0 + pw;
  const int Mw = 1 << p.lgMw, Mh = 1 << p.lgMh;
  const int Ktot = p.TH * p.TW * p.C;

  // ---- per-thread A rows (fixed for the whole K loop)
  int a_iy0[A_CH], a_ix0[A_CH], a_boff[A_CH];
  bool a_ok[A_CH];
#pragma unroll
  for (int i = 0; i < A_CH; ++i) {
    const int id = tid + 256 * i;
    const int m = m0 + (id >> 3);
    a_ok[i] = (id < BM * 8) && m < p.M;
    const int mx = m & (Mw - 1), my = (m >> p.lgMw) & (Mh - 1), b = m >> (p.lgMw + p.lgMh);
    a_iy0[i] = my * p.sy + oy0;
    a_ix0[i] = mx * p.sx + ox0;
    a_boff[i] = b * p.Hi * p.Wi * p.C;
  }
  const int kq = tid & 7;          // 16-byte column of the 32-wide K step
  const int kh = kq >> 2, kl = kq & 3;  // chunk (0/1) and 16-byte column inside the chunk

  f32x4 ra[DB ? 2 : 1][A_CH], rb[DB ? 2 : 1][B_CH];      // operand pieces in flight (DB: two steps' worth)
  const __amdgpu_buffer_rsrc_t rsA = vf_rsrc(p.A, p.a_bytes), rsW = vf_rsrc(p.Wt, p.w_bytes);

  auto tap_index = [&](int tap) {
    const int th = tap / p.TW, tw = tap - th * p.TW;
    return (kh0 + th * p.khs) * 4 + (kw0 + tw * p.kws);
  };

  // ---- vector path (V >= 1): everything that depends on the K step is wave-uniform (SALU); a lane only adds a
  // per-step delta to byte offsets it computed once, and tests one bit of its per-row tap-validity mask.
  unsigned a_mask[A_CH];   // bit t: tap t of this row is inside the image
  unsigned a_byte[A_CH];   // byte offset of (tap (0,0), c = 4*kl) for this row (wraps for padding rows; masked)
  unsigned w_byte[B_CH];   // per-thread weight byte offset at (tap 0, c0 = 0)
  bool w_ok[B_CH];
  if constexpr (V >= 1) {
#pragma unroll
    for (int i = 0; i < A_CH; ++i) {
      a_byte[i] = 4u * (unsigned)(a_boff[i] + (a_iy0[i] * p.Wi + a_ix0[i]) * p.C + 4 * kl);
      // bit (th*TW + tw) = row th and column tw of the tap window are inside the image: an outer product of a row
      // mask and a column mask (TH, TW <= 4) — a loop over all taps with a division each cost 8 us of prologue
      unsigned ym = 0, xm = 0;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        if (t < p.TH && (unsigned)(a_iy0[i] + t * p.ty) < (unsigned)p.Hi) ym |= 1u << t;
        if (t < p.TW && (unsigned)(a_ix0[i] + t * p.tx) < (unsigned)p.Wi) xm |= 1u << t;
      }
      unsigned mk = 0;
#pragma unroll
      for (int t = 0; t < 4; ++t)
        if ((ym >> t) & 1u) mk |= xm << (t * p.TW);
      a_mask[i] = a_ok[i] ? mk : 0u;
    }
#pragma unroll
    for (int i = 0; i < B_CH; ++i) {
      const int id = tid + 256 * i;
      if constexpr (!BKM) {
        const int n = n0 + (id >> 3);
        w_ok[i] = id < BN * 8 && n < p.N;
        w_byte[i] = 4u * (unsigned)(n * p.wsN + 4 * kl);
      } else {
        const int kk = id / (BN / 4), nq = id - kk * (BN / 4);
        const int n = n0 + 4 * nq;
        w_ok[i] = id < BN * 8 && n < p.N;
        w_byte[i] = 4u * (unsigned)((kk & 15) * p.wsC + n);
      }
    }
  }
  // uniform description of one 16-wide chunk: (tap, byte delta for A, byte delta for W, in range).  The K loop walks
  // chunks in order, so (tap, c0) advance incrementally — no division in the loop (a wave issues ~1 instruction per
  // 4-5 cycles; every instruction that is not an MFMA shortens the time the matrix pipe is fed).
  struct Chunk { int tap; unsigned dA, dW; bool ok; };
  const int lgTW = p.TW == 4 ? 2 : (p.TW == 2 ? 1 : 0);
  const int sA_th = 4 * p.ty * p.Wi * p.C, sA_tw = 4 * p.tx * p.C;
  const int sW_th = 4 * p.khs * 4 * p.wsTap, sW_tw = 4 * p.kws * p.wsTap, sW_0 = 4 * (kh0 * 4 + kw0) * p.wsTap;
  const int sW_c = 4 * (BKM ? p.wsC : 1);
  int it_tap = 0, it_c0 = 0, it_q = 0;     // state of the NEXT chunk to be described
  // the memory-order walk (KLIN) keeps its OWN two counters and is selected at compile time: written as a runtime
  // branch with the mirror-image update of it_tap / it_c0, LLVM merges the two branches into a select of ADDRESSES,
  // the counters move to scratch memory, and every K step of EVERY kernel starts with a scratch_load +
  // s_waitcnt vmcnt(0) in front of the operand prefetches (measured: -5 % per step)
  int kl_tap = 0, kl_c0 = 0;
  const int ntaps = p.TH * p.TW;
  if constexpr (V >= 1) {
    // chunk order: channel chunk OUTER, tap INNER — the 16 taps of one channel chunk touch the same input window, so
    // the re-reads are temporally close (L1/L2 hits) instead of one full window sweep per tap
    it_q = 2 * kt0;
    it_c0 = (it_q / ntaps) << 4;
    it_tap = it_q % ntaps;
    if constexpr (KLIN) {
      // a 1x1 output map (the bottleneck layers) has no window overlap between taps to exploit; walking K in memory
      // order instead makes each weight row a single forward stream of 128-byte lines per block rather than 64-byte
      // pieces 2 KB apart (the 131 MB weight matrix is what this GEMM is bounded by)
      const int cpt = p.C >> 4;
      kl_tap = it_q / cpt;
      kl_c0 = (it_q - kl_tap * cpt) << 4;
    }
  }
  auto next_chunk = [&]() {
    Chunk c;
    const int tap = KLIN ? kl_tap : it_tap, c0 = KLIN ? kl_c0 : it_c0;
    const int th = tap >> lgTW, tw = tap & (p.TW - 1);
    c.tap = tap;
    c.dA = (unsigned)(th * sA_th + tw * sA_tw + 4 * c0);
    c.dW = (unsigned)(sW_0 + th * sW_th + tw * sW_tw + c0 * sW_c);
    c.ok = it_q < p.nq;
    ++it_q;
    if constexpr (KLIN) {
      kl_c0 += 16;
      if (kl_c0 >= p.C) {
        kl_c0 = 0;
        ++kl_tap;
      }
    } else {
      if (++it_tap >= ntaps) {
        it_tap = 0;
        it_c0 += 16;
      }
    }
    return c;
  };

  // One K step's operands arrive as A_CH + B_CH independent pieces (one 16-byte register each).  `begin_tile` does
  // the wave-uniform part once per step; the pieces are issued one at a time BETWEEN the MFMAs of the running step
  // (a contiguous block of ~60 address/load instructions would leave the matrix pipe empty: it queues one MFMA).
  Chunk tc0, tc1;
  int t_tapv = 0;
  unsigned t_dA = 0;
  bool t_okq = false;
  int t_kt = 0;
  int s_dy[4], s_dx[4], s_c[4], s_w[4];     // scalar path (V = 0): per-step decomposition of this thread's k
  bool s_kok[4];
  auto begin_tile = [&](int kt, bool live) {
    t_kt = kt;
    if constexpr (V >= 1) {
      tc0 = next_chunk();      // SALU; begin_tile is called with consecutive kt
      tc1 = next_chunk();
      tc0.ok = tc0.ok && live;
      tc1.ok = tc1.ok && live;
      t_tapv = kh ? tc1.tap : tc0.tap;
      t_dA = kh ? tc1.dA : tc0.dA;
      t_okq = kh ? tc1.ok : tc0.ok;
    } else {
      t_okq = live;
      // scalar path: the (tap, c) decomposition of this thread's four k depends on the step only, not on the row —
      // done once here instead of once per row and element (a runtime division each; the 3-channel first layer
      // spent most of its time on them)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int k = kt * BK + 4 * kq + j;
        const int tap = k / p.C, c = k - tap * p.C;
        const int th = tap / p.TW, tw = tap - th * p.TW;
        s_dy[j] = th * p.ty;
        s_dx[j] = tw * p.tx;
        s_c[j] = c;
        s_kok[j] = live && k < Ktot;
        s_w[j] = tap_index(tap) * p.wsTap + c * p.wsC;
      }
    }
  };
  auto load_piece = [&](int pc, auto SET) {
    constexpr int rs = decltype(SET)::value;
    const int kt = t_kt;
    if constexpr (V >= 1) {
      if (pc < A_CH) {
        // ---------------- A: lanes with kh = 0 / 1 fetch chunk 0 / 1
        const int i = pc;
        const bool ok = t_okq && ((a_mask[i] >> t_tapv) & 1u);
        ra[rs][i] = vf_bload4(rsA, ok ? a_byte[i] + t_dA : VF_OOB);
      } else {
        // ---------------- B
        const int i = pc - A_CH;
        bool hi;
        if constexpr (!BKM) {
          hi = kh;
        } else {
          hi = (((tid + 256 * i) / (BN / 4)) >> 4) & 1;
        }
        const unsigned dW = hi ? tc1.dW : tc0.dW;
        const bool ok = w_ok[i] && (hi ? tc1.ok : tc0.ok);
        if constexpr (V == 2) {
          rb[rs][i] = vf_bload4(rsW, ok ? w_byte[i] + dW : VF_OOB);
        } else {   // V == 1: k-major B whose N is not a multiple of 4 -> four scalar loads along n
          const int n = n0 + 4 * ((tid + 256 * i) % (BN / 4));
          f32x4 v;
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = vf_bload1(rsW, (ok && n + j < p.N) ? w_byte[i] + dW + 4u * j : VF_OOB);
          rb[rs][i] = v;
        }
      }
    } else {
      // ---------------- scalar path: flattened K index, per-element (tap, c)
      if (pc < A_CH) {
        const int i = pc;
        f32x4 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int iy = a_iy0[i] + s_dy[j], ix = a_ix0[i] + s_dx[j];
          const bool ok = s_kok[j] && a_ok[i] && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
          v[j] = vf_bload1(rsA, ok ? 4u * (unsigned)(a_boff[i] + (iy * p.Wi + ix) * p.C + s_c[j]) : VF_OOB);
        }
        ra[rs][i] = v;
      } else {
        const int i = pc - A_CH;
        const int id = tid + 256 * i;
        f32x4 v;
        if constexpr (!BKM) {
          const int n = n0 + (id >> 3);
          const bool okn = t_okq && id < BN * 8 && n < p.N;
#pragma unroll
          for (int j = 0; j < 4; ++j)
            v[j] = vf_bload1(rsW, (okn && s_kok[j]) ? 4u * (unsigned)(n * p.wsN + s_w[j]) : VF_OOB);
        } else {
          const int kk = id / (BN / 4), nq = id - kk * (BN / 4);
          const int n = n0 + 4 * nq;
          const int k = kt * BK + kk;
          const int tap = k / p.C, c = k - tap * p.C;
          const bool okk = t_okq && id < BN * 8 && k < Ktot;
          const int base = tap_index(tap) * p.wsTap + c * p.wsC;
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = vf_bload1(rsW, (okk && n + j < p.N) ? 4u * (unsigned)(base + n + j) : VF_OOB);
        }
        rb[rs][i] = v;
      }
    }
  };
  auto load_tile = [&](int kt) {
    begin_tile(kt, true);
#pragma unroll
    for (int pc = 0; pc < A_CH + B_CH; ++pc) load_piece(pc, VfIC<0>{});
  };

  auto store_piece = [&](int buf, int pc, auto SET) {
    constexpr int rs = decltype(SET)::value;
    if constexpr (BF) {
      // mode 3: plane q = top 16 bits of the residual after the planes before it (vf_split3); mode 1: bf16(v), RNE
      __bf16* base = (__bf16*)smem + buf * (NP * PL_SZ);
      if (pc < A_CH) {
        const int i = pc, id = tid + 256 * i;
        if (256 * i + 255 < BM * 8 || id < BM * 8) {
          __bf16* dst = base + piece_off(id >> 3, kq);
          if constexpr (NP == 3) {
            const VfPlanes3 s3 = vf_split3(ra[rs][i]);
#pragma unroll
            for (int q = 0; q < 3; ++q) *(uint2*)(dst + q * PL_SZ) = s3.p[q];
          } else {
            *(bf16x4*)dst = __builtin_convertvector(ra[rs][i], bf16x4);
          }
        }
      } else {
        const int i = pc - A_CH, id = tid + 256 * i;
        if (256 * i + 255 < BN * 8 || id < BN * 8) {
          f32x4 v = rb[rs][i];
          int off;
          if constexpr (!BKM) {
            off = piece_off(id >> 3, kq);
          } else {      // the piece holds 4 consecutive n at one k: stored as it comes, transposed by the reads
            const int kk = id / (BN / 4), nq = id - kk * (BN / 4);
            off = kk * LDN + 4 * nq;
          }
          __bf16* dst = base + AH_SZ + off;
          if constexpr (NP == 3) {
            const VfPlanes3 s3 = vf_split3(v);
#pragma unroll
            for (int q = 0; q < 3; ++q) *(uint2*)(dst + q * PL_SZ) = s3.p[q];
          } else {
            *(bf16x4*)dst = __builtin_convertvector(v, bf16x4);
          }
        }
      }
      return;
    }
    float* As = smem + buf * (A_SZ + B_SZ);
    float* Bs = As + A_SZ;
    if (pc < A_CH) {
      const int i = pc, id = tid + 256 * i;
      if (256 * i + 255 < BM * 8 || id < BM * 8) *(f32x4*)(As + (id >> 3) * LDA + 4 * kq) = ra[rs][i];
    } else {
      const int i = pc - A_CH, id = tid + 256 * i;
      if (256 * i + 255 < BN * 8 || id < BN * 8) {     // constant-true for full pieces: no exec-mask branch in the loop
        if constexpr (!BKM) {
          *(f32x4*)(Bs + (id >> 3) * LDB + 4 * kq) = rb[rs][i];
        } else {
          const int kk = id / (BN / 4), nq = id - kk * (BN / 4);
          *(f32x4*)(Bs + kk * LDB + 4 * nq) = rb[rs][i];
        }
      }
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int pc = 0; pc < A_CH + B_CH; ++pc) store_piece(buf, pc, VfIC<0>{});
  };

  f32x16 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

  const int lr = lane & 31, lh = lane >> 5;

  if (kt0 < kt1) {
    load_tile(kt0);
    store_tile(0);
  }
  __syncthreads();
  if (p.stamps && tid == 0) p.stamps[8 * blockIdx.x + 1] = wall_clock64();
  if constexpr (DB) {
    // Step kt (register set S = parity of kt - kt0, LDS buffer S):
    //   MFMA slot i < NPC  is followed by operand piece i of step kt+2 (global -> register set S, whose previous
    //                      content — step kt — went to LDS during step kt-1);
    //   the last NPC slots are each followed by one piece of step kt+1 (register set S^1, loaded during step kt-1):
    //                      split into planes + three LDS writes into buffer S^1, in the shadow of the running MFMA;
    //   the fragments of the second 16-deep group are read right after the first MFMA.  One barrier per step.
    constexpr int NPC = A_CH + B_CH;
    constexpr int NMF = 6 * MT * NT;               // MFMAs per 16-deep group
    constexpr int SLOTS = (BK / 16) * NMF;
    static_assert(2 * NPC <= SLOTS, "loads and LDS writes of one K step must fit between its MFMAs");
    begin_tile(kt0 + 1, kt0 + 1 < kt1);
#pragma unroll
    for (int pc = 0; pc < NPC; ++pc) load_piece(pc, VfIC<1>{});
    auto step = [&](int kt, auto SET) {
      constexpr int S = decltype(SET)::value;
      const __bf16* base = (const __bf16*)smem + S * (NP * PL_SZ);
      // nothing of this step may float above this point (left alone, the split of step kt+1's pieces is hoisted to the
      // top of the step, in front of the loads issued here, and waits for ALL outstanding loads there)
      __builtin_amdgcn_sched_barrier(0);
      begin_tile(kt + 2, kt + 2 < kt1);
      bf16x8 a[2][NP][MT], b[2][NP][NT];
      auto read_frag = [&](int g) {
#pragma unroll
        for (int q = 0; q < NP; ++q) {
          const __bf16* Ah = base + q * PL_SZ;
          const __bf16* Bh = Ah + AH_SZ;
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) a[g][q][mt] = *(const bf16x8*)(Ah + frag_off(wm + mt * 32 + lr, 2 * g + lh));
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            if constexpr (!BKM)
              b[g][q][nt] = *(const bf16x8*)(Bh + frag_off(wn + nt * 32 + lr, 2 * g + lh));
            else
              b[g][q][nt] = vf_tr_frag<LDN>(Bh, wn + nt * 32, 16 * g, lane);
          }
        }
      };
      read_frag(0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int g = 0; g < BK / 16; ++g) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int t = 0; t < 6; ++t) {       // smallest terms first
              constexpr int qa[6] = {1, 0, 2, 0, 1, 0}, qb[6] = {1, 2, 0, 1, 0, 0};
              acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[g][qa[t]][mt], b[g][qb[t]][nt], acc[mt][nt], 0, 0, 0);
              const int slot = g * NMF + (mt * NT + nt) * 6 + t;
              if (slot == 0 && BK / 16 > 1) read_frag(1);
              if (slot < NPC) load_piece(slot, SET);
              // (unconditional: on the last step this writes zeros into the buffer nobody reads again; a branch here
              //  makes the compiler drain every outstanding load at the top of each step)
              if (slot >= SLOTS - NPC) store_piece(S ^ 1, slot - (SLOTS - NPC), VfIC<(S ^ 1)>{});
              __builtin_amdgcn_sched_barrier(0);
            }
      }
      __syncthreads();
    };
    for (int kt = kt0; kt < kt1; kt += 2) {
      step(kt, VfIC<0>{});
      if (kt + 1 < kt1) step(kt + 1, VfIC<1>{});
    }
  } else
  if constexpr (BF) {
    for (int kt = kt0; kt < kt1; ++kt) {
      const int buf = NBUF == 2 ? ((kt - kt0) & 1) : 0;
      const __bf16* base = (const __bf16*)smem + buf * (NP * PL_SZ);
      VF_SPY(0);
      begin_tile(kt + 1, kt + 1 < kt1);
#pragma unroll
      for (int pc = 0; pc < A_CH + B_CH; ++pc) load_piece(pc, VfIC<0>{});
      VF_SPY(1);
#pragma unroll
      for (int g = 0; g < BK / 16; ++g) {
        bf16x8 a[NP][MT], b[NP][NT];
#pragma unroll
        for (int q = 0; q < NP; ++q) {
          const __bf16* Ah = base + q * PL_SZ;
          const __bf16* Bh = Ah + AH_SZ;
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) a[q][mt] = *(const bf16x8*)(Ah + frag_off(wm + mt * 32 + lr, 2 * g + lh));
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            if constexpr (!BKM)
              b[q][nt] = *(const bf16x8*)(Bh + frag_off(wn + nt * 32 + lr, 2 * g + lh));
            else
              b[q][nt] = vf_tr_frag<LDN>(Bh, wn + nt * 32, 16 * g, lane);
          }
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            if constexpr (NP == 3) {      // smallest terms first
              acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][mt], b[1][nt], acc[mt][nt], 0, 0, 0);
              acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][mt], b[2][nt], acc[mt][nt], 0, 0, 0);
              acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2][mt], b[0][nt], acc[mt][nt], 0, 0, 0);
              acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][mt], b[1][nt], acc[mt][nt], 0, 0, 0);
              acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][mt], b[0][nt], acc[mt][nt], 0, 0, 0);
            }
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][mt], b[0][nt], acc[mt][nt], 0, 0, 0);
          }
      }
      // keep the split of the next step's pieces below this step's MFMAs: hoisted above them (it depends on the loads
      // only) it makes the wave wait for a load it issued two MFMAs earlier
      if constexpr (NBUF == 1) __builtin_amdgcn_sched_barrier(0);
      VF_SPY(2);
      if constexpr (NBUF == 1) __syncthreads();       // everyone has read the tile before it is overwritten
      VF_SPY(3);
      store_tile(NBUF == 2 ? (buf ^ 1) : 0);
      VF_SPY(4);
      __syncthreads();
      VF_SPY(5);
    }
  } else {
  constexpr int NPC = A_CH + B_CH;             // operand pieces per K step
  constexpr int NMF = 4 * MT * NT;             // MFMAs per 8-wide sub-step
  constexpr int SLOTS = (BK / 8) * NMF;        // MFMA slots per K step
  static_assert(2 * NPC <= SLOTS, "loads and LDS writes of one K step must fit between its MFMAs");
  for (int kt = kt0; kt < kt1; ++kt) {
    const int buf = (kt - kt0) & 1;
    const float* As = smem + buf * (A_SZ + B_SZ);
    const float* Bs = As + A_SZ;
    // The step's schedule, pinned with scheduling barriers (left alone, the compiler gathers all loads at the top and
    // all LDS writes at the bottom, and the matrix pipe idles through both):
    //   fragment reads run one sub-step ahead of the MFMAs that consume them (two waves sharing a SIMD interleave
    //   their MFMAs one for one and finish a group together, so a read issued only then is latency nobody covers);
    //   MFMA slot i < NPC is followed by operand piece i of step kt+1 (global -> register);
    //   the last NPC slots are each followed by one register -> LDS write into the other buffer.
    auto read_frag = [&](int ss, f32x4 (&a)[MT], f32x4 (&b)[NT]) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) a[mt] = *(const f32x4*)(As + (wm + mt * 32 + lr) * LDA + 8 * ss + 4 * lh);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        if constexpr (!BKM) {
          b[nt] = *(const f32x4*)(Bs + (wn + nt * 32 + lr) * LDB + 8 * ss + 4 * lh);
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) b[nt][j] = Bs[(8 * ss + 4 * lh + j) * LDB + wn + nt * 32 + lr];
        }
      }
    };
    const bool more = kt + 1 < kt1 && !(p.dbg & 1);
    begin_tile(kt + 1, more);
    f32x4 fa[2][MT], fb[2][NT];
    read_frag(0, fa[0], fb[0]);
#pragma unroll
    for (int ss = 0; ss < BK / 8; ++ss) {
      if (ss + 1 < BK / 8) read_frag(ss + 1, fa[(ss + 1) & 1], fb[(ss + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[ss & 1][mt][j], fb[ss & 1][nt][j], acc[mt][nt], 0, 0, 0);
            const int slot = ss * NMF + (j * MT + mt) * NT + nt;
            if (slot < NPC) load_piece(slot, VfIC<0>{});
            if (slot >= SLOTS - NPC) store_piece(buf ^ 1, slot - (SLOTS - NPC), VfIC<0>{});
            __builtin_amdgcn_sched_barrier(0);
          }
    }
    __syncthreads();
  }

  }
  if (p.stamps && tid == 0) p.stamps[8 * blockIdx.x + 2] = wall_clock64();
  // ---- epilogue: C/D layout col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
  float* out = p.ksplit > 1 ? p.slab + (int64_t)ks * p.out_elems : p.Y;
  const bool fin = p.ksplit == 1;
  const int stm = fin ? p.st.mode : 0;      // with split-K the statistics come out of the slab reduce instead
  float bv[NT], sv[NT], st1[NT], st2[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int n = n0 + wn + nt * 32 + lr;
    bv[nt] = (fin && p.bias && n < p.N) ? p.bias[n] : 0.f;
    // statistics: the shift (forward) or the saved mean of this row tile's batch group (backward)
    sv[nt] = (stm && n < p.N) ? p.st.vec[(stm == 2 ? (bx / p.st.tiles_per_group) * p.N : 0) + n] : 0.f;
    st1[nt] = 0.f;
    st2[nt] = 0.f;
  }
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = wm + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      const int m = m0 + row;
      if (m >= p.M) continue;
      const int mx = m & (Mw - 1), my = (m >> p.lgMw) & (Mh - 1), b = m >> (p.lgMw + p.lgMh);
      const int64_t pix = ((int64_t)b * p.outH + (my * p.osy + ooy0)) * p.outW + (mx * p.osx + oox0);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int n = n0 + wn + nt * 32 + lr;
        if (n < p.N) {
          float v = acc[mt][nt][r];
          if (fin) {
            v = vf_act_apply(v + bv[nt], p.act, p.slope);
            if (p.dmask) v = vf_act_grad(p.dmask[pix * p.N + n], v, p.dact, p.dslope);
            if (stm == 1) {
              const float d = v - sv[nt];
              st1[nt] += d;
              st2[nt] += d * d;
            } else if (stm == 2) {
              st1[nt] += v;
              st2[nt] += v * (p.st.x[pix * p.N + n] - sv[nt]);
            }
          }
          out[pix * p.N + n] = v;
        }
      }
    }
  }
  if (stm)      // (uniform over the block; the K loop ended on a barrier, so the tile memory is free)
    vf_bn_tile_partials<NT, BM / WM, BN>(p.st, st1, st2, smem, wave / WAVES_N, wn, lane, tid, n0, p.N, bx,
                                         p.parity ? ((ph << 1) | pw) : 0);
  if (p.stamps && tid == 0) {
    __builtin_amdgcn_s_waitcnt(0);     // stores retired (vmcnt 0)
    p.stamps[8 * blockIdx.x + 3] = wall_clock64();
    p.stamps[8 * blockIdx.x + 5] = (long long)__builtin_readcyclecounter();
  }
}

// y = act(sum_s slab[s] + bias)  or  dst = beta*dst + sum_s slab[s]  (wgrad).  Scalar form (any total).
__global__ void k_slab_reduce(const float* __restrict__ slab, float* __restrict__ dst, const float* __restrict__ bias,
                              int64_t total, int N, int ksplit, int act, float slope, float beta,
                              const float* __restrict__ dmask = nullptr, int dact = 0, float dslope = 0.f) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    float s = 0.f;
    for (int k = 0; k < ksplit; ++k) s += slab[(int64_t)k * total + i];
    if (bias) s += bias[i % N];
    if (beta != 0.f) s += beta * dst[i];
    s = vf_act_apply(s, act, slope);
    if (dmask) s = vf_act_grad(dmask[i], s, dact, dslope);
    dst[i] = s;
  }
}
// 16-byte form: COLS float4 columns x (256 / COLS) split lanes per block; lanes are combined in a fixed order.
// COLS = 64 is the workhorse; COLS = 16 (16 split lanes) serves small outputs with very deep splits (the image-side
// weight gradients: 48 x 64 outputs from 500+ slabs — twelve 64-column blocks walking 128 slabs each sat at 15 us).
template <int COLS>
__global__ __launch_bounds__(256) void k_slab_reduce4(const float* __restrict__ slab, float* __restrict__ dst,
                                                      const float* __restrict__ bias, int64_t total4, int N, int ksplit,
                                                      int act, float slope, float beta,
                                                      const float* __restrict__ dmask = nullptr, int dact = 0,
                                                      float dslope = 0.f) {
  constexpr int L = 256 / COLS;
  const int tx = threadIdx.x % COLS, sl = threadIdx.x / COLS;
  const int64_t i4 = (int64_t)blockIdx.x * COLS + tx;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  if (i4 < total4) {
#pragma unroll 4
    for (int k = sl; k < ksplit; k += L) s += ((const f32x4*)slab)[(int64_t)k * total4 + i4];
  }
  __shared__ f32x4 red[L][COLS];
  red[sl][tx] = s;
  __syncthreads();
  if (sl == 0 && i4 < total4) {
    f32x4 t = red[0][tx];
#pragma unroll
    for (int l = 1; l < L; ++l) t += red[l][tx];
    if (beta != 0.f) t += beta * ((const f32x4*)dst)[i4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float v = t[e];
      if (bias) v += bias[(i4 * 4 + e) % N];
      t[e] = vf_act_apply(v, act, slope);
    }
    if (dmask) {
      const f32x4 m = ((const f32x4*)dmask)[i4];
#pragma unroll
      for (int e = 0; e < 4; ++e) t[e] = vf_act_grad(m[e], t[e], dact, dslope);
    }
    ((f32x4*)dst)[i4] = t;
  }
}
// Split-K combine that also leaves BatchNorm statistics partials (VfBnSt): the output [M][N] is walked as units of 256
// consecutive floats (64 float4 lanes); a block owns one 256-wide column chunk (blockIdx.y; N <= 256: the unit holds 256 / N
// whole rows) and a run of `units_per_block` row units, 4 row lanes deep, every thread summing its element's slabs in
// slab order.  A thread always sees the same four channels, so its statistics stay in registers until the end; the lanes
// that alias a channel (4 row lanes; 256 / N column lanes when N < 256) meet in LDS in a fixed order.  Requires
// total % 256 == 0 and (256 % N == 0 or N % 256 == 0) — every layer the nets pair with a BatchNorm.
__global__ __launch_bounds__(256) void k_slab_reduce_st(const float* __restrict__ slab, float* __restrict__ dst,
                                                        const float* __restrict__ bias, int64_t total4, int N, int ksplit, int act,
                                                        float slope, const float* __restrict__ dmask, int dact, float dslope,
                                                        int units_per_block, int64_t row_units, int ncc, const VfBnSt st) {
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int cc = blockIdx.y;
  const int ch0 = (N <= 256 ? (tx * 4) % N : cc * 256 + tx * 4);      // first of this thread's four channels
  const int g = blockIdx.x / st.tiles_per_group;                        // batch group of this block's rows
  f32x4 bv = {0.f, 0.f, 0.f, 0.f}, sv = {0.f, 0.f, 0.f, 0.f};
  if (bias) bv = *(const f32x4*)(bias + ch0);
  if (st.mode) sv = *(const f32x4*)(st.vec + (st.mode == 2 ? g * N : 0) + ch0);
  f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
  const int64_t u0 = (int64_t)blockIdx.x * units_per_block, u1 = min(row_units, u0 + units_per_block);
  for (int64_t ru = u0 + ty; ru < u1; ru += 4) {
    const int64_t i4 = (ru * ncc + cc) * 64 + tx;
    f32x4 t = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int k = 0; k < ksplit; ++k) t += ((const f32x4*)slab)[(int64_t)k * total4 + i4];
    t += bv;
#pragma unroll
    for (int e = 0; e < 4; ++e) t[e] = vf_act_apply(t[e], act, slope);
    if (dmask) {
      const f32x4 m = ((const f32x4*)dmask)[i4];
#pragma unroll
      for (int e = 0; e < 4; ++e) t[e] = vf_act_grad(m[e], t[e], dact, dslope);
    }
    if (st.mode == 1) {
      const f32x4 d = t - sv;
      s1 += d;
      s2 += d * d;
    } else if (st.mode == 2) {
      const f32x4 xv = ((const f32x4*)st.x)[i4];
      s1 += t;
      s2 += t * (xv - sv);
    }
    ((f32x4*)dst)[i4] = t;
  }
  if (!st.mode) return;
  __shared__ f32x4 red[2][256];
  red[0][threadIdx.x] = s1;
  red[1][threadIdx.x] = s2;
  __syncthreads();
  const int lanes = N <= 256 ? N / 4 : 64;        // distinct float4 channel columns in this block
  if ((int)threadIdx.x < lanes) {
    double a[4] = {0, 0, 0, 0}, b[4] = {0, 0, 0, 0};
    for (int t = threadIdx.x; t < 256; t += lanes) {     // (t % 64) % lanes == threadIdx.x: same channels (lanes divides 64)
      const f32x4 u = red[0][t], w = red[1][t];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        a[e] += (double)u[e];
        b[e] += (double)w[e];
      }
    }
    const int local = blockIdx.x - g * st.tiles_per_group;
    double* o = st.part + ((int64_t)(g * st.rows_per_group + local) * 2) * N;
    const int c0 = (N <= 256 ? 0 : cc * 256) + threadIdx.x * 4;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      o[c0 + e] = a[e];
      o[N + c0 + e] = b[e];
    }
  }
}
// plan of that launch: (units_per_block, blocks per group); false: the shape does not fit (caller keeps the plain reduce)
static bool slab_st_plan(int64_t total, int N, int groups, int rows_cap, int* units_per_block, int* blocks_per_group, int* ncc_out,
                         int64_t* row_units_out) {
  if (total % 256 != 0 || N % 4 != 0 || !((N <= 256 && 256 % N == 0) || (N > 256 && N % 256 == 0))) return false;
  const int ncc = N <= 256 ? 1 : N / 256;
  const int64_t row_units = total / 256 / ncc;
  if (row_units % groups != 0) return false;
  const int64_t per_group = row_units / groups;
  // ~512 blocks over the whole launch, at least 4 row units (one per row lane) each, a divisor of the group's units
  int64_t upb = std::max<int64_t>(4, vf_cdiv(row_units * ncc, 512));
  while (upb < per_group && per_group % upb != 0) ++upb;
  if (upb > per_group) upb = per_group;
  if (per_group % upb != 0) return false;
  const int64_t bpg = per_group / upb;
  if (bpg > rows_cap) return false;
  *units_per_block = (int)upb;
  *blocks_per_group = (int)bpg;
  *ncc_out = ncc;
  *row_units_out = row_units;
  return true;
}
static int launch_slab_reduce(vf_ctx* ctx, const float* slab, float* dst, const float* bias, int64_t total, int N, int ksplit,
                              int act, float slope, float beta, const float* dmask = nullptr, int dact = 0,
                              float dslope = 0.f) {
  if (total % 4 == 0 && ((((uintptr_t)slab) | ((uintptr_t)dst) | ((uintptr_t)dmask)) & 15) == 0 && ksplit > 0) {
    const int64_t total4 = total / 4;
    if (total4 <= 8192 && ksplit >= 64)
      hipLaunchKernelGGL(k_slab_reduce4<16>, dim3((int)vf_cdiv(total4, 16)), dim3(256), 0, ctx->stream, slab, dst, bias, total4, N,
                         ksplit, act, slope, beta, dmask, dact, dslope);
    else
      hipLaunchKernelGGL(k_slab_reduce4<64>, dim3((int)vf_cdiv(total4, 64)), dim3(256), 0, ctx->stream, slab, dst, bias, total4, N,
                         ksplit, act, slope, beta, dmask, dact, dslope);
  } else {
    const int nb = (int)std::min<int64_t>(vf_cdiv(total, 256), 4096);
    hipLaunchKernelGGL(k_slab_reduce, dim3(nb), dim3(256), 0, ctx->stream, slab, dst, bias, total, N, ksplit, act, slope, beta,
                       dmask, dact, dslope);
  }
  VF_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------
struct WGrad {
  const float* U;   // plain operand at the LOW-res pixel grid: [P][Nu]
  const float* V;   // gathered operand on the HIGH-res grid: [B][Hv][Wv][Cv]
  float* dW;        // [Nu][16][Cv]
  float* slab;
  unsigned u_bytes, v_bytes;
  int P, lgMh, lgMw;
  int Nu, Cv, Hv, Wv;
  int stride, pad;  // iy = my*stride - pad + kh
  int ntaps;        // 16 (4x4 gather) or 1 (V already holds one K-row per pixel)
  int gx, gy, gz;   // logical grid (1-D launch, remapped per XCD)
  int ksplit, nk;   // nk = ceil(P/16)
  float beta;
};

// BM over n (64 or 128); BN = 128 columns (tap,c).  VU / VV: 16-byte loads legal for U / V.
// BF: bf16-operand mode — U and V go into LDS as k-major bf16 tiles (one 8-byte store per piece), the fragments come
// out through the transposing read (vf_tr_frag) and the products run on v_mfma_f32_32x32x16_bf16, fp32 accumulation.
template <int BM, bool VU, bool VV, int BF = 0>
__device__ __forceinline__ void wgrad_body(const WGrad& p, const int block_id) {
  constexpr int BN = 128, BK = BF ? 32 : 16;      // bf16 mode: two 16-deep MFMA groups per barrier
  constexpr int LDU = BM + 4, LDV = BN + 4;
  constexpr int WN = BM == 128 ? 64 : 32;
  constexpr int NT = WN / 32;
  constexpr int WAVES_N = BN / WN;
  constexpr int U_CH = (BK * BM / 4) / 256;  // 2 (BM=128) or 1 (BM=64)
  constexpr int V_CH = (BK * BN / 4) / 256;  // 2
  constexpr int U_SZ = BK * LDU, V_SZ = BK * LDV;
  constexpr int LDMU = BM + 32, LDMV = BN + 32;        // bf16 elements per k row (BF): k-major tiles, transposing reads
  constexpr int UH_SZ = BK * LDMU, VH_SZ = BK * LDMV;
  constexpr int NP = BF > 0 ? BF : 1;                  // bf16 planes per operand (3: exact split of the fp32 operands)
  constexpr int NBUF = BF == 3 ? 1 : 2;
  constexpr int PL_SZ = UH_SZ + VH_SZ;
  constexpr int SMEM_F = BF ? (NBUF * NP * PL_SZ + 1) / 2 : 2 * (U_SZ + V_SZ);
  __shared__ __attribute__((aligned(16))) float smem[SMEM_F];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = (wave / WAVES_N) * 64, wn = (wave % WAVES_N) * WN;
  const int lid = vf_xcd_remap(block_id, p.gx * p.gy * p.gz);
  const int n0 = ((lid / p.gx) % p.gy) * BM, j0 = (lid % p.gx) * BN;
  const int ks = lid / (p.gx * p.gy);
  const int steps = (p.nk + p.ksplit - 1) / p.ksplit;
  const int kt0 = ks * steps, kt1 = min(p.nk, kt0 + steps);
  const int Ncols = p.ntaps * p.Cv;
  const int Mw = 1 << p.lgMw, Mh = 1 << p.lgMh;

  // V columns owned by this thread: 4 consecutive columns starting at j0 + 4*cq
  const int cq = tid & 31;
  int v_dy[4], v_dx[4], v_c[4];
  bool v_okc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int col = j0 + 4 * cq + j;
    v_okc[j] = col < Ncols;
    const int tap = v_okc[j] ? col / p.Cv : 0;
    v_c[j] = v_okc[j] ? col - tap * p.Cv : 0;
    v_dy[j] = (tap >> 2) - p.pad;
    v_dx[j] = (tap & 3) - p.pad;
  }
  const int uq = tid % (BM / 4), ukk = tid / (BM / 4);  // U: 4 consecutive n at pixel row ukk (+ 1024/BM per chunk)
  const int un = n0 + 4 * uq;

  f32x4 ru[U_CH], rv[V_CH];
  const __amdgpu_buffer_rsrc_t rsU = vf_rsrc(p.U, p.u_bytes), rsV = vf_rsrc(p.V, p.v_bytes);

  // one K step = U_CH + V_CH operand pieces (one 16-byte register each); `live` false turns a piece into a no-op load
  auto load_piece = [&](int kt, int pc, bool live) {
    if (pc < U_CH) {
      const int i = pc;
      const int pp = kt * BK + ukk + i * (1024 / BM);
      const bool okp = live && pp < p.P;
      if constexpr (VU) {
        ru[i] = vf_bload4(rsU, (okp && un < p.Nu) ? 4u * (unsigned)(pp * p.Nu + un) : VF_OOB);
      } else {
        f32x4 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          v[j] = vf_bload1(rsU, (okp && un + j < p.Nu) ? 4u * (unsigned)(pp * p.Nu + un + j) : VF_OOB);
        }
        ru[i] = v;
      }
    } else {
      const int i = pc - U_CH;
      const int pp = kt * BK + (tid >> 5) + 8 * i;
      const bool okp = live && pp < p.P;
      const int mx = pp & (Mw - 1), my = (pp >> p.lgMw) & (Mh - 1), b = pp >> (p.lgMw + p.lgMh);
      const int boff = b * p.Hv * p.Wv * p.Cv;
      if constexpr (VV) {
        const int iy = my * p.stride + v_dy[0], ix = mx * p.stride + v_dx[0];
        const bool ok = okp && v_okc[0] && (unsigned)iy < (unsigned)p.Hv && (unsigned)ix < (unsigned)p.Wv;
        rv[i] = vf_bload4(rsV, ok ? 4u * (unsigned)(boff + (iy * p.Wv + ix) * p.Cv + v_c[0]) : VF_OOB);
      } else {
        f32x4 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int iy = my * p.stride + v_dy[j], ix = mx * p.stride + v_dx[j];
          const bool ok = okp && v_okc[j] && (unsigned)iy < (unsigned)p.Hv && (unsigned)ix < (unsigned)p.Wv;
          v[j] = vf_bload1(rsV, ok ? 4u * (unsigned)(boff + (iy * p.Wv + ix) * p.Cv + v_c[j]) : VF_OOB);
        }
        rv[i] = v;
      }
    }
  };
  auto store_piece = [&](int buf, int pc) {
    if constexpr (BF) {
      __bf16* base = (__bf16*)smem + buf * (NP * PL_SZ);
      f32x4 v;
      int off;
      if (pc < U_CH) {
        v = ru[pc];
        off = (ukk + pc * (1024 / BM)) * LDMU + 4 * uq;
      } else {
        const int i = pc - U_CH;
        v = rv[i];
        off = UH_SZ + ((tid >> 5) + 8 * i) * LDMV + 4 * cq;
      }
      if constexpr (NP == 3) {
        const VfPlanes3 s3 = vf_split3(v);
#pragma unroll
        for (int q = 0; q < 3; ++q) *(uint2*)(base + q * PL_SZ + off) = s3.p[q];
      } else {
        *(bf16x4*)(base + off) = __builtin_convertvector(v, bf16x4);
      }
      return;
    }
    float* Us = smem + buf * (U_SZ + V_SZ);
    float* Vs = Us + U_SZ;
    if (pc < U_CH) {
      *(f32x4*)(Us + (ukk + pc * (1024 / BM)) * LDU + 4 * uq) = ru[pc];
    } else {
      const int i = pc - U_CH;
      *(f32x4*)(Vs + ((tid >> 5) + 8 * i) * LDV + 4 * cq) = rv[i];
    }
  };
  constexpr int NPC = U_CH + V_CH;
  auto load_tile = [&](int kt) {
#pragma unroll
    for (int pc = 0; pc < NPC; ++pc) load_piece(kt, pc, true);
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int pc = 0; pc < NPC; ++pc) store_piece(buf, pc);
  };

  f32x16 acc[2][NT];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;
  const int lr = lane & 31, lh = lane >> 5;

  if (kt0 < kt1) {
    load_tile(kt0);
    store_tile(0);
  }
  __syncthreads();
  if constexpr (BF) {
    for (int kt = kt0; kt < kt1; ++kt) {
      const int buf = NBUF == 2 ? ((kt - kt0) & 1) : 0;
      const bool more = kt + 1 < kt1;
      const __bf16* base = (const __bf16*)smem + buf * (NP * PL_SZ);
#pragma unroll
      for (int pc = 0; pc < NPC; ++pc) load_piece(kt + 1, pc, more);
#pragma unroll
      for (int g = 0; g < BK / 16; ++g) {
        bf16x8 a[NP][2], b[NP][NT];
#pragma unroll
        for (int q = 0; q < NP; ++q) {
          const __bf16* Uh = base + q * PL_SZ;
          const __bf16* Vh = Uh + UH_SZ;
#pragma unroll
          for (int mt = 0; mt < 2; ++mt) a[q][mt] = vf_tr_frag<LDMU>(Uh, wm + mt * 32, 16 * g, lane);
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) b[q][nt] = vf_tr_frag<LDMV>(Vh, wn + nt * 32, 16 * g, lane);
        }
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            if constexpr (NP == 3) {
              acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][mt], b[1][nt], acc[mt][nt], 0, 0, 0);
              acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][mt], b[2][nt], acc[mt][nt], 0, 0, 0);
              acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2][mt], b[0][nt], acc[mt][nt], 0, 0, 0);
              acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][mt], b[1][nt], acc[mt][nt], 0, 0, 0);
              acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][mt], b[0][nt], acc[mt][nt], 0, 0, 0);
            }
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][mt], b[0][nt], acc[mt][nt], 0, 0, 0);
          }
      }
      if constexpr (NBUF == 1) __syncthreads();
      store_tile(NBUF == 2 ? (buf ^ 1) : 0);
      __syncthreads();
    }
  } else {
  constexpr int NMF = 4 * 2 * NT;              // MFMAs per 8-wide sub-step
  constexpr int SLOTS = 2 * NMF;               // MFMA slots per K step (BK = 16)
  static_assert(2 * NPC <= SLOTS, "loads and LDS writes of one K step must fit between its MFMAs");
  for (int kt = kt0; kt < kt1; ++kt) {
    const int buf = (kt - kt0) & 1;
    const bool more = kt + 1 < kt1;
    const float* Us = smem + buf * (U_SZ + V_SZ);
    const float* Vs = Us + U_SZ;
    // same pinned schedule as k_igemm: fragment reads one sub-step ahead; MFMA slot i < NPC is followed by operand
    // piece i of step kt+1, the last NPC slots by the register -> LDS writes into the other buffer
    auto read_frag = [&](int ss, float (&a)[2][4], float (&b)[NT][4]) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int k = 8 * ss + 4 * lh + j;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) a[mt][j] = Us[k * LDU + wm + mt * 32 + lr];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) b[nt][j] = Vs[k * LDV + wn + nt * 32 + lr];
      }
    };
    float fa[2][2][4], fb[2][NT][4];
    read_frag(0, fa[0], fb[0]);
#pragma unroll
    for (int ss = 0; ss < 2; ++ss) {
      if (ss == 0) read_frag(1, fa[1], fb[1]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[ss][mt][j], fb[ss][nt][j], acc[mt][nt], 0, 0, 0);
            const int slot = ss * NMF + (j * 2 + mt) * NT + nt;
            if (slot < NPC) load_piece(kt + 1, slot, more);
            if (slot >= SLOTS - NPC) store_piece(buf ^ 1, slot - (SLOTS - NPC));
            __builtin_amdgcn_sched_barrier(0);
          }
    }
    __syncthreads();
  }

  }
  const int64_t total = (int64_t)p.Nu * Ncols;
  float* out = p.ksplit > 1 ? p.slab + (int64_t)ks * total : p.dW;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int n = n0 + wm + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (n >= p.Nu) continue;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int col = j0 + wn + nt * 32 + lr;
        if (col < Ncols) {
          const int64_t idx = (int64_t)n * Ncols + col;
          float v = acc[mt][nt][r];
          if (p.ksplit == 1 && p.beta != 0.f) v += p.beta * out[idx];
          out[idx] = v;
        }
      }
    }
}

template <int BM, bool VU, bool VV, int BF = 0>
__global__ __launch_bounds__(256) void k_wgrad(const WGrad p) {
  wgrad_body<BM, VU, VV, BF>(p, blockIdx.x);
}

// Every weight gradient of a backward walk in ONE launch: dW_l needs only (input_l, gradOutput_l), both of which stay in
// their modules' buffers until the walk is over, so the GEMMs are recorded during the walk and launched together —
// one ramp and one tail instead of one per layer (each 512-block launch is a single round of blocks), and the
// descriptor table travels by value in the kernel arguments (no device table, nothing to keep in sync under a graph).
#define VF_WG_GROUP_MAX 16
struct WGradGroup {
  int n;
  int blk_off[VF_WG_GROUP_MAX + 1];   // multiples of 8: the XCD-aware tile order of a layer assumes blockIdx % 8 == local id % 8
  WGrad d[VF_WG_GROUP_MAX];
};
static_assert(sizeof(WGradGroup) <= 3584, "kernel arguments are limited to 4 KB");
template <int BM, bool VU, bool VV, int BF>
__global__ __launch_bounds__(256) void k_wgrad_group(const WGradGroup G) {
  int l = 0;
  while (l + 1 < G.n && (int)blockIdx.x >= G.blk_off[l + 1]) ++l;
  const int local = (int)blockIdx.x - G.blk_off[l];
  if (local >= G.d[l].gx * G.d[l].gy * G.d[l].gz) return;      // padding blocks (uniform exit)
  wgrad_body<BM, VU, VV, BF>(G.d[l], local);
}
struct SlabGroup {
  int n;
  int blk_off[VF_WG_GROUP_MAX + 1];
  struct { const float* slab; float* dst; int64_t total4; int ksplit; float beta; } d[VF_WG_GROUP_MAX];
};
__global__ __launch_bounds__(256) void k_slab_reduce4_group(const SlabGroup G) {
  int l = 0;
  while (l + 1 < G.n && (int)blockIdx.x >= G.blk_off[l + 1]) ++l;
  const int tx = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const int64_t i4 = (int64_t)((int)blockIdx.x - G.blk_off[l]) * 64 + tx;
  const int64_t total4 = G.d[l].total4;
  const int ksplit = G.d[l].ksplit;
  const float* slab = G.d[l].slab;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  if (i4 < total4) {
#pragma unroll 4
    for (int k = sl; k < ksplit; k += 4) s += ((const f32x4*)slab)[(int64_t)k * total4 + i4];
  }
  __shared__ f32x4 red[4][64];
  red[sl][tx] = s;
  __syncthreads();
  if (sl == 0 && i4 < total4) {
    f32x4 t = ((red[0][tx] + red[1][tx]) + red[2][tx]) + red[3][tx];
    f32x4* dst = (f32x4*)G.d[l].dst;
    if (G.d[l].beta != 0.f) t += G.d[l].beta * dst[i4];
    dst[i4] = t;
  }
}

// ------------------------------------------------------------------------------------------------
// Cout == 1, 4x4 stride-1 conv on a 4x4 map (netD's last layer, train.lua:195): dot products.
__global__ __launch_bounds__(256) void k_dot_fwd(const float* __restrict__ x, const float* __restrict__ w,
                                                 const float* __restrict__ bias, float* __restrict__ y, int K, int act,
                                                 float slope) {
  // one block per sample: 16-byte loads, four independent partial sums per thread, wave shuffle tree, 4 waves via LDS
  const int b = blockIdx.x;
  const float* xb = x + (int64_t)b * K;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  if ((K & 3) == 0) {
    for (int k = 4 * threadIdx.x; k < K; k += 1024) {
      const f32x4 xv = *(const f32x4*)(xb + k), wv = *(const f32x4*)(w + k);
      acc[0] += xv[0] * wv[0];
      acc[1] += xv[1] * wv[1];
      acc[2] += xv[2] * wv[2];
      acc[3] += xv[3] * wv[3];
    }
  } else {
    for (int k = threadIdx.x; k < K; k += 256) acc[0] += xb[k] * w[k];
  }
  float s = (acc[0] + acc[1]) + (acc[2] + acc[3]);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  __shared__ float red[4];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) y[b] = vf_act_apply(((red[0] + red[1]) + (red[2] + red[3])) + (bias ? bias[0] : 0.f), act, slope);
}
// ya != NULL: gy is the gradient w.r.t. the ACTIVATED output ya (the Sigmoid fused into this conv): its derivative rides here
__global__ void k_dot_bwd_data(const float* __restrict__ gy, const float* __restrict__ w, float* __restrict__ gx, int B,
                               int K, const float* __restrict__ ya, int act, float slope) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < (int64_t)B * K) {
    const int64_t b = i / K;
    const float g = ya ? vf_act_grad(ya[b], gy[b], act, slope) : gy[b];
    gx[i] = g * w[i % K];
  }
}
// gw[k] = beta*gw[k] + sum_b gy[b]*x[b][k]: a block owns 64 columns; its 8 waves take every 8th sample (coalesced 256-byte rows),
// eight samples in flight per wave, partials meet in LDS in a fixed order.  The per-sample factor (gy, times the derivative of a
// fused activation: see k_dot_bwd_data) is formed once per block, in LDS.  (One thread per column walking all B samples was a 64-deep
// dependent load chain on 32 blocks: 19 us for 2 MB; four waves with two sums in flight each still took 22 us at batch 128 — every
// iteration waited for its row AND for two scalar loads in front of it.)
__global__ __launch_bounds__(512) void k_dot_bwd_weight(const float* __restrict__ x, const float* __restrict__ gy,
                                                        float* __restrict__ gw, float* __restrict__ gb, int B, int K,
                                                        float beta, const float* __restrict__ ya, int act, float slope) {
  const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int k = blockIdx.x * 64 + lane;
  __shared__ float g[1024];
  __shared__ float red[8][64];
  float acc = 0.f;
  for (int b0 = 0; b0 < B; b0 += 1024) {                 // (B <= 1024 in every net: one pass)
    const int nb = min(1024, B - b0);
    __syncthreads();
    for (int b = threadIdx.x; b < nb; b += 512) g[b] = ya ? vf_act_grad(ya[b0 + b], gy[b0 + b], act, slope) : gy[b0 + b];
    __syncthreads();
    if (k < K) {
      const float* xp = x + (int64_t)b0 * K + k;
      int b = grp;
      for (; b + 56 < nb; b += 64) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = xp[(int64_t)(b + 8 * j) * K];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc += g[b + 8 * j] * v[j];
      }
      for (; b < nb; b += 8) acc += g[b] * xp[(int64_t)b * K];
    }
  }
  red[grp][lane] = acc;
  __syncthreads();
  if (grp == 0 && k < K) {
    const float t = ((red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane])) + ((red[4][lane] + red[5][lane]) + (red[6][lane] + red[7][lane]));
    gw[k] = (beta != 0.f ? beta * gw[k] : 0.f) + t;
  }
  if (gb && blockIdx.x == 0 && threadIdx.x == 0) {
    float t = 0.f;
    for (int b = 0; b < B; ++b) t += ya ? vf_act_grad(ya[b], gy[b], act, slope) : gy[b];
    gb[0] = (beta != 0.f ? beta * gb[0] : 0.f) + t;
  }
}

// ------------------------------------------------------------------------------------------------
// Thin-output transposed passes (image side of the nets: N = 3, 12 ...: netG's last full-conv, netD's first-layer
// data-grad).  A 32-wide MFMA tile would be >60% padding there, so the pass runs as a plain GEMM to a column matrix
//   cols[B*Hi*Wi][16*N] = A[pixel][:] . W[:][(kh,kw,n)]        (the [c][kh][kw][n] weights ARE that [C][16N] matrix)
// followed by k_col2im4x4, in which every output pixel gathers its 4 taps (+bias, activation).  Measured 98 -> 30 us.
// (The mirror-image trick for thin INPUTS — im2col then GEMM — was measured slower than the scalar-gather path of
// k_igemm<V=0> for these K = 48 problems and is not used.)
__global__ __launch_bounds__(256) void k_col2im4x4(const float* __restrict__ cols, const float* __restrict__ bias,
                                                   float* __restrict__ y, int B, int lgHi, int lgWi, int N, int act, float slope) {
  // stride 2, pad 1: output (oh, ow) = (2i-1+kh, 2j-1+kw)  =>  kh = (oh+1) & 1 + 2*th, i = (oh+1-kh)/2
  const int Hi = 1 << lgHi, Wi = 1 << lgWi;
  const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const int64_t total = ((int64_t)B << (lgHi + lgWi + 2)) * N;
  if (t >= total) return;
  const int n = (int)(t % N);
  const int64_t pix = t / N;
  const int ow = (int)(pix & (2 * Wi - 1)), oh = (int)((pix >> (lgWi + 1)) & (2 * Hi - 1)), b = (int)(pix >> (lgWi + lgHi + 2));
  float s = bias ? bias[n] : 0.f;
#pragma unroll
  for (int th = 0; th < 2; ++th) {
    const int kh = ((oh + 1) & 1) + 2 * th, i = (oh + 1 - kh) >> 1;
    if ((unsigned)i >= (unsigned)Hi) continue;
#pragma unroll
    for (int tw = 0; tw < 2; ++tw) {
      const int kw = ((ow + 1) & 1) + 2 * tw, j = (ow + 1 - kw) >> 1;
      if ((unsigned)j >= (unsigned)Wi) continue;
      s += cols[((((int64_t)b << lgHi) + i) * Wi + j) * (16 * N) + (kh * 4 + kw) * N + n];
    }
  }
  y[t] = vf_act_apply(s, act, slope);
}

// ================================================================================================ host
static inline bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

template <int BM, int BN, int WM, int WN, int BF>
static void launch_igemm_tile_m(vf_ctx* ctx, const IGemm& g, dim3 grid, bool bkm, int v, const char* name, double flops,
                                bool db = false) {
  dim3 block(256);
  // algorithmic bytes: both operands and the output once (small-batch passes over the bottleneck weights are bounded by these,
  // not by the matrix pipe: bench.py prices a kernel against whichever of its two rooflines is the longer)
  const double bytes = (double)g.a_bytes + (double)g.w_bytes + 4.0 * (double)g.out_elems;
  if constexpr (BF == 3 && BM == 64 && BN == 64) {
    if (db && v == 2) {
      if (!bkm)
        VF_LAUNCH_TIMED(ctx, name, flops, bytes, (k_igemm<BM, BN, WM, WN, false, 2, 3, false, true>), grid, block, g);
      else
        VF_LAUNCH_TIMED(ctx, name, flops, bytes, (k_igemm<BM, BN, WM, WN, true, 2, 3, false, true>), grid, block, g);
      return;
    }
  }
  if constexpr (BM == 64 && BN == 128) {
    if (g.klin && !bkm && v == 2) {
      VF_LAUNCH_TIMED(ctx, name, flops, bytes, (k_igemm<BM, BN, WM, WN, false, 2, BF, true>), grid, block, g);
      return;
    }
  }
  if (!bkm) {
    if (v == 2)
      VF_LAUNCH_TIMED(ctx, name, flops, bytes, (k_igemm<BM, BN, WM, WN, false, 2, BF>), grid, block, g);
    else
      VF_LAUNCH_TIMED(ctx, name, flops, bytes, (k_igemm<BM, BN, WM, WN, false, 0, BF>), grid, block, g);
  } else {
    if (v == 2)
      VF_LAUNCH_TIMED(ctx, name, flops, bytes, (k_igemm<BM, BN, WM, WN, true, 2, BF>), grid, block, g);
    else if (v == 1)
      VF_LAUNCH_TIMED(ctx, name, flops, bytes, (k_igemm<BM, BN, WM, WN, true, 1, BF>), grid, block, g);
    else
      VF_LAUNCH_TIMED(ctx, name, flops, bytes, (k_igemm<BM, BN, WM, WN, true, 0, BF>), grid, block, g);
  }
}
template <int BM, int BN, int WM, int WN>
static void launch_igemm_tile(vf_ctx* ctx, const IGemm& g, dim3 grid, bool bkm, int v, const char* name, double flops,
                              bool db = false) {
  if (ctx->mfma_bf16 == 3)
    launch_igemm_tile_m<BM, BN, WM, WN, 3>(ctx, g, grid, bkm, v, name, flops, db);
  else if (ctx->mfma_bf16 == 1)
    launch_igemm_tile_m<BM, BN, WM, WN, 1>(ctx, g, grid, bkm, v, name, flops);
  else
    launch_igemm_tile_m<BM, BN, WM, WN, 0>(ctx, g, grid, bkm, v, name, flops);
}

// vecA / vecB: 16-byte loads legal for the A / B operand.  top: this launch writes the pass's output tensor itself (an inner
// GEMM into a column buffer does not), so a pending BatchNorm-statistics attachment (vf_bn_fuse_next_*) applies to it.
int vf_internal_smallm_plan(int form, int M, int N, int K, size_t ws_bytes);                                                   // vf_smallm.hip
int vf_internal_smallm_launch(vf_ctx* ctx, int form, const float* A, const float* W, float* slab, int M, int N, int K, int ksplit);
static int launch_igemm(vf_ctx* ctx, IGemm& g, bool vecA, bool vecB, bool top = true) {
  const int zpar = g.parity ? 4 : 1;
  const bool bkm = g.wsN == 1 && g.wsC != 1;
  const int64_t a_elems = (int64_t)(g.M >> (g.lgMh + g.lgMw)) * g.Hi * g.Wi * g.C;
  const int64_t w_elems = bkm ? (int64_t)g.C * g.wsC : (int64_t)g.N * g.wsN;
  VF_REQUIRE(a_elems < ((int64_t)1 << 29) && w_elems < ((int64_t)1 << 29), "operand exceeds the 2 GiB buffer-descriptor range");
  g.a_bytes = (unsigned)(a_elems * 4);
  g.w_bytes = (unsigned)(w_elems * 4);
  int v = (vecA && vecB) ? 2 : ((vecA && bkm) ? 1 : 0);
  const int Ktot = g.TH * g.TW * g.C;
  g.nq = g.TH * g.TW * (g.C / 16);
  g.nk = v >= 1 ? (g.nq + 1) / 2 : (int)vf_cdiv(Ktot, 32);
  // ---- tile selection: the largest tile that still gives >= 2 blocks per CU, else the smallest + split-K
  struct Tile { int bm, bn; };
  Tile t;
  if (g.N <= 32) {
    t = {256, 32};
  } else if (g.M <= 64) {
    t = {64, 128};
  } else {
    const Tile cand[3] = {{128, 128}, {128, 64}, {64, 64}};
    // mode 3's single-buffered three-plane tiles hide their two barriers per K step with co-resident blocks: they want
    // twice the blocks (measured: +2 % per step at 1024 and above)
    static const int env_min_blocks = getenv("VF_TILE_MIN_BLOCKS") ? atoi(getenv("VF_TILE_MIN_BLOCKS")) : 0;
    const int tune_min_blocks = env_min_blocks ? env_min_blocks : (ctx->mfma_bf16 == 3 ? 1024 : 512);
    int pick = 2;
    for (int i = 0; i < 3; ++i) {
      if (cand[i].bn > 64 && g.N <= 64) continue;
      // a short K loop (<= 8 steps: the 64-channel transposed passes) leaves a 128-row block mostly prologue and
      // epilogue; four 64x64 blocks per CU cover each other's ends better (E2 data-gradient: 122 -> 112 us)
      if (g.parity && g.nk <= 8 && cand[i].bm > 64) continue;
      const int64_t blocks = vf_cdiv(g.M, cand[i].bm) * vf_cdiv(g.N, cand[i].bn) * zpar;
      if (blocks >= tune_min_blocks) {
        pick = i;
        break;
      }
    }
    t = cand[pick];
    // (tried: 256 x 64 tiles — four 64x64 wave tiles stacked in M, one block per CU — for the N = 64 layers: 153 us
    //  against 125 us for E2's data-gradient; one wave per SIMD does not cover its own LDS/global latency)
  }
  static const int env_split_blocks = getenv("VF_SPLIT_BLOCKS") ? atoi(getenv("VF_SPLIT_BLOCKS")) : 0;
  const int tune_split_blocks = env_split_blocks ? env_split_blocks : 512;     // (768 for the M <= 64 GEMMs: a wash)
  const int gm = (int)vf_cdiv(g.M, t.bm), gn = (int)vf_cdiv(g.N, t.bn);
  const int64_t blocks = (int64_t)gm * gn * zpar;
  int ksplit = 1;
  if (blocks < tune_split_blocks * 3 / 4 && g.nk >= 8) {
    ksplit = (int)std::min<int64_t>(g.nk / 4, vf_cdiv(tune_split_blocks, blocks));
    const size_t slab_bytes = (size_t)g.out_elems * sizeof(float);
    while (ksplit > 1 && (size_t)ksplit * slab_bytes > vf_ws_avail(ctx)) --ksplit;
    if (ksplit < 1) ksplit = 1;
    const int steps = (int)vf_cdiv(g.nk, ksplit);  // avoid empty trailing splits
    ksplit = (int)vf_cdiv(g.nk, steps);
  }
  // ---- the bottleneck GEMMs at a small batch (M <= 8 rows against hundreds of MB of weights): a weight-streaming kernel instead
  // of a tile (vf_smallm.hip); it leaves split-K slabs, which the combine below takes like the tiled kernel's
  int sm_form = -1;
  if (!g.parity && g.lgMh == 0 && g.lgMw == 0 && g.M <= 8 && v == 2) {
    if (!bkm && g.Hi == g.TH && g.Wi == g.TW && g.oy0 == 0 && g.ox0 == 0 && g.sy == 1 && g.sx == 1 && g.ty == 1 && g.tx == 1 && g.wsC == 1 &&
        g.wsTap == g.C && g.wsN == Ktot && g.kh0 == 0 && g.kw0 == 0 && g.khs == 1 && g.kws == 1)
      sm_form = 0;        // A rows and weight rows are K = taps * C contiguous floats
    else if (bkm && g.TH * g.TW == 1 && g.Hi == 1 && g.Wi == 1 && g.wsC == g.N)
      sm_form = 1;        // weights [C][N], N contiguous
    if (sm_form >= 0) {
      const int s = vf_internal_smallm_plan(sm_form, g.M, g.N, Ktot, vf_ws_avail(ctx));
      if (s >= 2) ksplit = s;
      else sm_form = -1;
    }
  }
  g.ksplit = ksplit;
  g.slab = ksplit > 1 ? (float*)vf_ws_ptr(ctx) : nullptr;
  // ---- BatchNorm statistics as a by-product (one-shot attachment): from the epilogue, or from the slab reduce under split-K
  g.st.mode = 0;
  bool slab_st = false;
  int st_upb = 0, st_bpg = 0, st_ncc = 0;
  int64_t st_units = 0;
  if (top && ctx->bnf.mode) {
    const int groups = ctx->bnf_groups;
    VfBnSt st = ctx->bnf;
    bool fused = false;
    if (ksplit == 1) {
      if (g.M % groups == 0 && (g.M / groups) % t.bm == 0 && (int64_t)(gm / groups) * zpar <= ctx->bnf_rows_cap) {
        st.tiles_per_group = gm / groups;
        st.zpar = zpar;
        st.rows_per_group = (gm / groups) * zpar;
        g.st = st;
        fused = true;
      }
    } else if (g.out_elems % 4 == 0 && slab_st_plan(g.out_elems, g.N, groups, ctx->bnf_rows_cap, &st_upb, &st_bpg, &st_ncc, &st_units)) {
      st.tiles_per_group = st_bpg;
      st.zpar = 1;
      st.rows_per_group = st_bpg;
      g.st = st;            // (the GEMM kernel ignores it under split-K; launch_slab_reduce_st gets it)
      slab_st = true;
      fused = true;
    }
    if (fused) {
      ctx->bnf_result_rows = st.rows_per_group;
      if (st.mode == 2) {   // the output is stored masked by the activation derivative: what BatchNorm's backward sums
        g.dmask = ctx->bnf_yact;
        g.dact = ctx->bnf_act;
        g.dslope = ctx->bnf_slope;
      }
    }
    ctx->bnf.mode = 0;
  }
  static const int tune_dbg = getenv("VF_IGEMM_DBG") ? atoi(getenv("VF_IGEMM_DBG")) : 0;
  g.dbg = tune_dbg;
  static const int tune_klin = getenv("VF_NO_KLIN") ? 0 : 1;
  g.klin = tune_klin && !g.parity && g.lgMh == 0 && g.lgMw == 0 && g.TH * g.TW > 1 && t.bm == 64 && t.bn == 128 && !bkm && v == 2;
  g.gm = gm; g.gn = gn; g.gz = zpar * ksplit;
  dim3 grid((unsigned)gm * gn * zpar * ksplit);
  // timing experiments only: per-block stamps; every 32nd launch is synchronised and appended to the file
  // VF_IGEMM_STAMPS names (the launches in between run unsynchronised, so the dumped one sees sustained conditions)
  static const char* stamp_file = getenv("VF_IGEMM_STAMPS");
  static long long* stamp_buf = nullptr;
  static unsigned stamp_count = 0;
  g.stamps = nullptr;
  bool stamp_dump = false;
  if (stamp_file && grid.x <= 32768) {
    if (!stamp_buf) VF_CHECK_HIP(hipHostMalloc((void**)&stamp_buf, 8 * 32768 * sizeof(long long), hipHostMallocDefault));
    g.stamps = stamp_buf;
    stamp_dump = (++stamp_count % 32) == 0;
    if (stamp_dump) VF_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  }
  char pname[64];
  // one name per kernel symbol (what rocprofv3 lists): k_igemm<bm, bn, ., ., kmajorB, v, mode>
  // double-buffered schedule: where the grid leaves at most two blocks per CU anyway (its 66 KB tiles cost nothing then)
  static const int tune_db = getenv("VF_IGEMM_DB") ? atoi(getenv("VF_IGEMM_DB")) : 1;
  const bool db = ctx->mfma_bf16 == 3 && v == 2 && t.bm == 64 && t.bn == 64 &&
                  (tune_db == 2 || (tune_db == 1 && (int64_t)grid.x <= 2 * 256));
  snprintf(pname, sizeof(pname), "igemm_%dx%d_%s_v%d%s%s", t.bm, t.bn, bkm ? "kmajorB" : "rowB", v,
           ctx->mfma_bf16 == 3 ? "_bf16x3" : (ctx->mfma_bf16 ? "_bf16" : ""), db ? "_db" : "");
  if (sm_form >= 0) {
    if (int rc = vf_internal_smallm_launch(ctx, sm_form, g.A, g.Wt, g.slab, g.M, g.N, Ktot, ksplit)) return rc;
  } else {
    const double fl = 2.0 * (double)g.M * g.N * Ktot * zpar;
    if (t.bm == 256)
      launch_igemm_tile<256, 32, 64, 32>(ctx, g, grid, bkm, v, pname, fl);
    else if (t.bm == 128 && t.bn == 128)
      launch_igemm_tile<128, 128, 64, 64>(ctx, g, grid, bkm, v, pname, fl);
    else if (t.bm == 128)
      launch_igemm_tile<128, 64, 64, 32>(ctx, g, grid, bkm, v, pname, fl);
    else if (t.bn == 128)
      launch_igemm_tile<64, 128, 64, 32>(ctx, g, grid, bkm, v, pname, fl);
    else
      launch_igemm_tile<64, 64, 32, 32>(ctx, g, grid, bkm, v, pname, fl, db);
  }
  VF_LAUNCH_CHECK();
  if (stamp_dump) {
    VF_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    if (FILE* f = fopen(stamp_file, "a")) {
      fprintf(f, "# %s blocks=%u M=%d N=%d K=%d ksplit=%d\n", pname, grid.x, g.M, g.N, Ktot, ksplit);
      for (unsigned b = 0; b < grid.x; ++b) {
        fprintf(f, "%u", b);
        for (int q = 0; q < 6; ++q) fprintf(f, " %lld", g.stamps[8 * b + q]);
        fprintf(f, "\n");
      }
#ifdef VF_IGEMM_SPY
      for (unsigned b = 4096; b < 4128; ++b) {
        fprintf(f, "S%u", b);
        for (int q = 0; q < 6; ++q) fprintf(f, " %lld", g.stamps[8 * b + q]);
        fprintf(f, "\n");
        for (int q = 0; q < 8; ++q) g.stamps[8 * b + q] = 0;
      }
#endif
      fclose(f);
    }
  }
  if (ksplit > 1) {
    VfProf prof(ctx, slab_st ? "slab_reduce_igemm_bnstats" : "slab_reduce_igemm", 0.0, 4.0 * (double)g.out_elems * (ksplit + 1));
    if (slab_st) {
      hipLaunchKernelGGL(k_slab_reduce_st, dim3((unsigned)(st_bpg * ctx->bnf_groups), (unsigned)st_ncc), dim3(256), 0, ctx->stream,
                         (const float*)g.slab, g.Y, g.bias, g.out_elems / 4, g.N, ksplit, g.act, g.slope, g.dmask, g.dact, g.dslope,
                         st_upb, st_units, st_ncc, g.st);
      VF_LAUNCH_CHECK();
      return 0;
    }
    return launch_slab_reduce(ctx, g.slab, g.Y, g.bias, g.out_elems, g.N, ksplit, g.act, g.slope, 0.f, g.dmask, g.dact, g.dslope);
  }
  return 0;
}

// the split-K combine for vf_pgemm.hip's launches (plain, or leaving BatchNorm statistics partials)
bool vf_internal_slab_st_ok(int64_t total, int N, int groups, int rows_cap, int* blocks_per_group) {
  int upb = 0, ncc = 0;
  int64_t units = 0;
  return total % 4 == 0 && slab_st_plan(total, N, groups, rows_cap, &upb, blocks_per_group, &ncc, &units);
}
int vf_internal_slab_reduce(vf_ctx* ctx, const float* slab, float* dst, const float* bias, int64_t total, int N, int ksplit, int act,
                            float slope, const float* dmask, int dact, float dslope, const VfBnSt* st, int st_groups) {
  if (st) {
    int upb = 0, bpg = 0, ncc = 0;
    int64_t units = 0;
    VF_REQUIRE(total % 4 == 0 && slab_st_plan(total, N, st_groups, st->rows_per_group, &upb, &bpg, &ncc, &units) && bpg == st->rows_per_group,
               "vf_internal_slab_reduce: the statistics plan changed between the GEMM and its combine");
    hipLaunchKernelGGL(k_slab_reduce_st, dim3((unsigned)(bpg * st_groups), (unsigned)ncc), dim3(256), 0, ctx->stream, slab, dst, bias,
                       total / 4, N, ksplit, act, slope, dmask, dact, dslope, upb, units, ncc, *st);
    VF_LAUNCH_CHECK();
    return 0;
  }
  return launch_slab_reduce(ctx, slab, dst, bias, total, N, ksplit, act, slope, 0.f, dmask, dact, dslope);
}

// ---- BatchNorm statistics attachment (see VfBnSt; consumed by the next conv-like launch of this context)
VF_API int vf_bn_fuse_next_fwd(vf_ctx* ctx, const float* shift, double* part, int part_rows_cap, int groups) {
  VF_REQUIRE(shift && part && part_rows_cap > 0 && groups >= 1 && groups <= 64, "vf_bn_fuse_next_fwd: bad arguments");
  memset(&ctx->bnf, 0, sizeof(ctx->bnf));
  ctx->bnf.mode = 1;
  ctx->bnf.vec = shift;
  ctx->bnf.part = part;
  ctx->bnf_groups = groups;
  ctx->bnf_rows_cap = part_rows_cap / groups;
  ctx->bnf_result_rows = 0;
  return 0;
}
VF_API int vf_bn_fuse_next_bwd(vf_ctx* ctx, const float* x, const float* y_act, int act, float slope, const float* save_mean,
                               double* part, int part_rows_cap, int groups) {
  VF_REQUIRE(x && save_mean && part && part_rows_cap > 0 && groups >= 1 && groups <= 64, "vf_bn_fuse_next_bwd: bad arguments");
  VF_REQUIRE(act == VF_ACT_NONE || y_act != nullptr, "vf_bn_fuse_next_bwd: the activation derivative needs the activated output");
  memset(&ctx->bnf, 0, sizeof(ctx->bnf));
  ctx->bnf.mode = 2;
  ctx->bnf.vec = save_mean;
  ctx->bnf.x = x;
  ctx->bnf.part = part;
  ctx->bnf_yact = act == VF_ACT_NONE ? nullptr : y_act;
  ctx->bnf_act = act;
  ctx->bnf_slope = slope;
  ctx->bnf_groups = groups;
  ctx->bnf_rows_cap = part_rows_cap / groups;
  ctx->bnf_result_rows = 0;
  return 0;
}
VF_API int vf_bn_fuse_result(vf_ctx* ctx, int* rows_per_group) {
  VF_REQUIRE(rows_per_group != nullptr, "vf_bn_fuse_result: NULL");
  *rows_per_group = ctx->bnf_result_rows;
  ctx->bnf_result_rows = 0;
  ctx->bnf.mode = 0;      // an attachment no launch took (thin or generic shapes) does not linger
  return 0;
}

static int check_conv_args(int B, int H, int W, int Cin, int Cout, int k, int stride, int pad) {
  VF_REQUIRE(k == 4, "only 4x4 kernels are built by the reference (got k=%d)", k);
  VF_REQUIRE((stride == 2 && pad == 1) || (stride == 1 && pad == 0), "unsupported stride/pad %d/%d", stride, pad);
  VF_REQUIRE(B > 0 && Cin > 0 && Cout > 0, "bad sizes B=%d Cin=%d Cout=%d", B, Cin, Cout);
  VF_REQUIRE(vf_is_pow2(H) && vf_is_pow2(W), "spatial sizes must be powers of two (got %dx%d)", H, W);
  if (stride == 1) VF_REQUIRE(H >= 4 && W >= 4, "4x4 stride-1 conv needs H,W >= 4");
  if (stride == 2) VF_REQUIRE(H >= 2 && W >= 2, "stride-2 conv needs H,W >= 2");
  VF_REQUIRE((int64_t)Cin * Cout * 16 < ((int64_t)1 << 31), "weight tensor too large for 32-bit element offsets");
  return 0;
}

// Generic "conv-like" pass: Y[b,oy,ox,n] = sum_{kh,kw,c} A[b, oy*s-pad+kh, ox*s-pad+kw, c] * Wt(n,kh,kw,c)
// (conv forward: A = x, n = Cout, weights [n][kh][kw][c];  full-conv data-grad: A = gy, n = Cin_full,
//  weights [n][kh][kw][c] as well — same physical layout by construction.)
static int conv_like_fwd(vf_ctx* ctx, const float* A, const float* w, const float* bias, float* Y, int B, int Hi, int Wi,
                         int C, int N, int stride, int pad, int act, float slope) {
  const int Ho = (Hi + 2 * pad - 4) / stride + 1, Wo = (Wi + 2 * pad - 4) / stride + 1;
  VF_REQUIRE(vf_is_pow2(Ho) && vf_is_pow2(Wo), "output spatial sizes must be powers of two");
  if (N == 1 && stride == 1 && Hi == 4 && Wi == 4) {
    hipLaunchKernelGGL(k_dot_fwd, dim3(B), dim3(256), 0, ctx->stream, A, w, bias, Y, 16 * C, act, slope);
    VF_LAUNCH_CHECK();
    return 0;
  }
  IGemm g;
  memset(&g, 0, sizeof(g));
  g.A = A; g.Wt = w; g.bias = bias; g.Y = Y;
  g.lgMh = vf_ilog2(Ho); g.lgMw = vf_ilog2(Wo);
  g.M = B * Ho * Wo;
  g.Hi = Hi; g.Wi = Wi; g.C = C; g.N = N;
  g.TH = 4; g.TW = 4;
  g.sy = stride; g.ty = 1; g.oy0 = -pad; g.sx = stride; g.tx = 1; g.ox0 = -pad;
  g.kh0 = 0; g.khs = 1; g.kw0 = 0; g.kws = 1;
  g.wsN = 16 * C; g.wsC = 1; g.wsTap = C;
  g.outH = Ho; g.outW = Wo; g.osy = 1; g.osx = 1;
  g.out_elems = (int64_t)g.M * N;
  g.act = act; g.slope = slope;
  return launch_igemm(ctx, g, (C % 16 == 0) && aligned16(A), (C % 16 == 0) && aligned16(w));
}

// Generic "transposed" pass: Y[b,oh,ow,n] = sum_{kh,kw,c : oh = 2i-1+kh ...} A[b,i,j,c] * Wt[c][kh][kw][n]
// (conv data-grad: A = gy, c = Cout, n = Cin;  full-conv forward: A = x, c = Cin_full, n = Cout_full.)
extern "C" int vf_act_bwd(vf_ctx* ctx, const float* y, const float* gy, float* gx, int64_t n, int act, float slope);
int vf_internal_deconv_thin_out(vf_ctx* ctx, const float* x, const float* w, const float* bias, float* y, int B, int Hi, int Wi, int C,
                                int N, int act, float slope);      // vf_conv_thin.hip
static int conv_like_bwd(vf_ctx* ctx, const float* A, const float* w, const float* bias, float* Y, int B, int Hi, int Wi,
                         int C, int N, int stride, int pad, int act, float slope, const float* dmask = nullptr, int dact = 0,
                         float dslope = 0.f) {
  if (stride == 2 && pad == 1 && N <= 4 && !dmask && !ctx->bnf.mode) {      // the image side: direct kernel, no column matrix
    const int rc = vf_internal_deconv_thin_out(ctx, A, w, bias, Y, B, Hi, Wi, C, N, act, slope);
    if (rc >= 0) return rc;
  }
  IGemm g;
  memset(&g, 0, sizeof(g));
  g.A = A; g.Wt = w; g.bias = bias; g.Y = Y;
  g.dmask = dmask; g.dact = dact; g.dslope = dslope;
  g.Hi = Hi; g.Wi = Wi; g.C = C; g.N = N;
  g.wsN = 1; g.wsC = 16 * N; g.wsTap = N;
  g.act = act; g.slope = slope;
  const bool vecA = (C % 16 == 0) && aligned16(A);
  bool vecB = (N % 4 == 0) && aligned16(w);
  if (stride == 2 && N < 32 && vecA && aligned16(w) && ctx->ws_front == 0) {
    const int64_t Mi = (int64_t)B * Hi * Wi;
    const size_t col_bytes = (size_t)Mi * 16 * N * sizeof(float);
    if (col_bytes + ((size_t)32 << 20) <= ctx->ws_bytes) {
      // thin output: cols[pixel][(kh,kw,n)] = A[pixel][:] . W[:][(kh,kw,n)]  (columns contiguous in [c][kh][kw][n]), then
      // every output pixel gathers its 4 taps
      float* cols = (float*)ctx->ws;
      ctx->ws_front = (col_bytes + 255) & ~(size_t)255;
      IGemm q;
      memset(&q, 0, sizeof(q));
      q.A = A; q.Wt = w; q.bias = nullptr; q.Y = cols;
      q.M = (int)Mi; q.lgMh = 0; q.lgMw = 0;
      q.Hi = 1; q.Wi = 1; q.C = C; q.N = 16 * N;
      q.TH = 1; q.TW = 1; q.sy = 1; q.ty = 1; q.sx = 1; q.tx = 1; q.khs = 1; q.kws = 1;
      q.wsN = 1; q.wsC = 16 * N; q.wsTap = 0;
      q.outH = 1; q.outW = 1; q.osy = 1; q.osx = 1;
      q.out_elems = Mi * 16 * N;
      q.act = VF_ACT_NONE;
      const int rc = launch_igemm(ctx, q, true, true, false);
      ctx->ws_front = 0;
      if (rc) return rc;
      const int64_t total = Mi * 4 * N;
      VfProf prof(ctx, "col2im4x4", 0.0, 4.0 * ((double)Mi * 16 * N + (double)total));
      hipLaunchKernelGGL(k_col2im4x4, dim3((unsigned)vf_cdiv(total, 256)), dim3(256), 0, ctx->stream, (const float*)cols, bias, Y,
                         B, vf_ilog2(Hi), vf_ilog2(Wi), N, act, slope);
      VF_LAUNCH_CHECK();
      if (dmask) return vf_act_bwd(ctx, dmask, Y, Y, total, dact, dslope);      // thin path: one more pass
      return 0;
    }
  }
  if (stride == 2) {
    // output (2*Hi) x (2*Wi); per parity class a 2x2-tap GEMM over the low-res grid
    g.lgMh = vf_ilog2(Hi); g.lgMw = vf_ilog2(Wi);
    g.M = B * Hi * Wi;
    g.TH = 2; g.TW = 2;
    g.sy = 1; g.ty = -1; g.oy0 = 0; g.sx = 1; g.tx = -1; g.ox0 = 0;  // iy = my + ph - th
    g.khs = 2; g.kws = 2;                                            // kh = (1-ph) + 2*th
    g.outH = 2 * Hi; g.outW = 2 * Wi; g.osy = 2; g.osx = 2;
    g.parity = 1;
    g.out_elems = (int64_t)B * g.outH * g.outW * N;
  } else {
    // stride 1, pad 0, low-res 1x1 -> 4x4: plain GEMM whose N' = (kh,kw,n) columns are contiguous in [c][kh][kw][n]
    VF_REQUIRE(Hi == 1 && Wi == 1, "stride-1 transposed pass is built for the 1x1 bottleneck only (got %dx%d)", Hi, Wi);
    g.lgMh = 0; g.lgMw = 0;
    g.M = B;
    g.TH = 1; g.TW = 1;
    g.sy = 1; g.ty = 1; g.sx = 1; g.tx = 1;
    g.N = 16 * N;
    g.wsC = 16 * N; g.wsTap = 0;
    g.outH = 1; g.outW = 1; g.osy = 1; g.osx = 1;
    g.out_elems = (int64_t)B * 16 * N;
    vecB = ((16 * N) % 4 == 0) && aligned16(w);
  }
  VF_REQUIRE(!(dmask && bias), "activation-backward epilogue is for data-gradient passes (no bias)");
  const float* real_bias = bias;
  const int real_act = act;
  if (stride == 1 && bias) {  // bias index = column % N: apply bias + activation in a pointwise pass
    g.bias = nullptr;
    g.act = VF_ACT_NONE;
  }
  // (the 1x1 -> 4x4 form's GEMM columns are (kh, kw, n): not one channel per column, so no statistics by-product there)
  if (int rc = launch_igemm(ctx, g, vecA, vecB, stride == 2)) return rc;
  if (stride == 1 && real_bias) {
    const int64_t total = (int64_t)B * 16 * N;
    const int nb = (int)std::min<int64_t>(vf_cdiv(total, 256), 4096);
    hipLaunchKernelGGL(k_slab_reduce, dim3(nb), dim3(256), 0, ctx->stream, (const float*)Y, Y, real_bias, total, N, 0,
                       real_act, slope, 1.f);
    VF_LAUNCH_CHECK();
  }
  return 0;
}

// dW[n][kh][kw][c] = beta*dW + sum_p U[p][n] * V[b, my*s-pad+kh, mx*s-pad+kw, c]
// ---- recorder of a weight-gradient group (see k_wgrad_group)
struct WgRec {
  WGrad g;
  int bm, bf;          // tile rows (128 / 64), matrix-core mode; bf = 4: operands are pre-split planes (pw, k_pwgrad_group)
  VfPWGrad pw;
  bool plain_ok;       // a bottleneck layer (fp32 operands, K = batch) that can ride in a k_pwgrad_group launch: pw is its
                       // fp32-fed descriptor, used when the group has planes layers (else it stays with k_wgrad_group)
  int blocks;
  int64_t total;       // dW elements
  double flops;
};
struct WgRecorder {
  std::vector<WgRec> recs;
  size_t ws_used = 0;
};
void vf_internal_wg_free(vf_ctx* ctx) {
  delete (WgRecorder*)ctx->wg_rec;
  ctx->wg_rec = nullptr;
}
template <int BM, int BF>
static void launch_wg_group(vf_ctx* ctx, const WGradGroup& G, int blocks, const char* name, double flops) {
  VF_LAUNCH_TIMED(ctx, name, flops, 0.0, (k_wgrad_group<BM, true, true, BF>), dim3(blocks), dim3(256), G);
}
static int wg_flush(vf_ctx* ctx) {
  WgRecorder* R = (WgRecorder*)ctx->wg_rec;
  if (!R || R->recs.empty()) return 0;
  // the layers whose operands came as planes: one k_pwgrad_group launch (vf_pgemm.hip).  The bottleneck layers of the same
  // walk go with them in their fp32-fed form: write-bound tiles (131 MB of dW from K = batch) that hide under the MFMA-bound
  // ones of the same launch, as they did in k_wgrad_group
  {
    bool any_planes = false;
    for (const WgRec& r : R->recs) any_planes |= (r.bf == 4);
    if (any_planes)
      for (WgRec& r : R->recs)
        if ((r.bf == 3 || r.bf == 1) && r.plain_ok) {
          r.bf = 4;
          r.blocks = r.pw.gx * r.pw.gy;
        }
    VfPWGradGroup G;
    int blocks = 0;
    double fl = 0;
    G.n = 0;
    auto fire = [&]() -> int {
      if (G.n == 0) return 0;
      G.blk_off[G.n] = blocks;
      const int rc = vf_internal_pwgrad_group(ctx, G, blocks, "pwgrad_group_128x128x32", fl);
      G.n = 0;
      blocks = 0;
      fl = 0;
      return rc;
    };
    for (const WgRec& r : R->recs) {
      if (r.bf != 4) continue;
      G.blk_off[G.n] = blocks;
      G.d[G.n] = r.pw;
      ++G.n;
      blocks += (r.blocks + 7) & ~7;
      fl += r.flops;
      if (G.n == VF_PWG_MAX)
        if (int rc = fire()) return rc;
    }
    if (int rc = fire()) return rc;
  }
  // one launch per (tile rows, mode) family, VF_WG_GROUP_MAX layers at a time
  for (int bm : {128, 64})
    for (int bf : {3, 1, 0}) {
      WGradGroup G;
      int blocks = 0;
      double fl = 0;
      G.n = 0;
      auto fire = [&]() {
        if (G.n == 0) return;
        G.blk_off[G.n] = blocks;
        char name[64];
        snprintf(name, sizeof(name), "wgrad_group_%dx128%s", bm, bf == 3 ? "_bf16x3" : (bf ? "_bf16" : ""));
        if (bm == 128) {
          if (bf == 3) launch_wg_group<128, 3>(ctx, G, blocks, name, fl);
          else if (bf == 1) launch_wg_group<128, 1>(ctx, G, blocks, name, fl);
          else launch_wg_group<128, 0>(ctx, G, blocks, name, fl);
        } else {
          if (bf == 3) launch_wg_group<64, 3>(ctx, G, blocks, name, fl);
          else if (bf == 1) launch_wg_group<64, 1>(ctx, G, blocks, name, fl);
          else launch_wg_group<64, 0>(ctx, G, blocks, name, fl);
        }
        G.n = 0;
        blocks = 0;
        fl = 0;
      };
      for (const WgRec& r : R->recs) {
        if (r.bm != bm || r.bf != bf) continue;
        G.blk_off[G.n] = blocks;
        G.d[G.n] = r.g;
        ++G.n;
        blocks += (r.blocks + 7) & ~7;
        fl += r.flops;
        if (G.n == VF_WG_GROUP_MAX) fire();
      }
      fire();
    }
  VF_LAUNCH_CHECK();
  // the split-K slabs of all layers: one launch per VF_WG_GROUP_MAX layers
  {
    SlabGroup S;
    int blocks = 0;
    double bytes = 0;
    S.n = 0;
    auto fire = [&]() {
      if (S.n == 0) return;
      S.blk_off[S.n] = blocks;
      VfProf prof(ctx, "slab_reduce_wgrad_group", 0.0, bytes);
      hipLaunchKernelGGL(k_slab_reduce4_group, dim3(blocks), dim3(256), 0, ctx->stream, S);
      S.n = 0;
      blocks = 0;
      bytes = 0;
    };
    for (const WgRec& r : R->recs) {
      if (r.g.ksplit <= 1) continue;
      S.blk_off[S.n] = blocks;
      S.d[S.n].slab = r.g.slab;
      S.d[S.n].dst = r.g.dW;
      S.d[S.n].total4 = r.total / 4;
      S.d[S.n].ksplit = r.g.ksplit;
      S.d[S.n].beta = r.g.beta;
      ++S.n;
      blocks += (int)vf_cdiv(r.total / 4, 64);
      bytes += 4.0 * (double)r.total * (r.g.ksplit + 1);
      if (S.n == VF_WG_GROUP_MAX) fire();
    }
    fire();
  }
  VF_LAUNCH_CHECK();
  R->recs.clear();
  R->ws_used = 0;
  return 0;
}
VF_API int vf_wgrad_group_begin(vf_ctx* ctx) {
  if (!ctx->wg_rec) ctx->wg_rec = new WgRecorder();
  if (ctx->wg_active) {      // the previous walk was abandoned before its _end (a host-side error): drop what it recorded
    ((WgRecorder*)ctx->wg_rec)->recs.clear();
    ((WgRecorder*)ctx->wg_rec)->ws_used = 0;
  }
  ctx->wg_active = 1;
  return 0;
}
VF_API int vf_wgrad_group_abort(vf_ctx* ctx) {
  if (ctx->wg_rec) {
    ((WgRecorder*)ctx->wg_rec)->recs.clear();
    ((WgRecorder*)ctx->wg_rec)->ws_used = 0;
  }
  ctx->wg_active = 0;
  return 0;
}
VF_API int vf_wgrad_group_end(vf_ctx* ctx) {
  VF_REQUIRE(ctx->wg_active, "vf_wgrad_group_end: no group is open");
  ctx->wg_active = 0;
  return wg_flush(ctx);
}

// weight gradients recorded so far in the open group (not yet launched)
VF_API int vf_wgrad_group_count(vf_ctx* ctx, int* count) {
  VF_REQUIRE(count != nullptr, "vf_wgrad_group_count: NULL");
  *count = (ctx->wg_active && ctx->wg_rec) ? (int)((WgRecorder*)ctx->wg_rec)->recs.size() : 0;
  return 0;
}
// launch the FIRST `count` recorded weight gradients (record order = walk order: the layers nearest the net's output) as one
// group and keep the group open with the rest still recorded: a data-parallel host starts the exchange of the finished bucket
// and then ends the group — both launches sit at the END of the backward walk, none in the middle of its data-gradient chain
VF_API int vf_wgrad_group_end_partial(vf_ctx* ctx, int count) {
  VF_REQUIRE(ctx->wg_active, "vf_wgrad_group_end_partial: no group is open");
  WgRecorder* R = (WgRecorder*)ctx->wg_rec;
  if (!R || R->recs.empty() || count <= 0) return 0;
  if (count >= (int)R->recs.size()) {
    const size_t used = R->ws_used;
    const int rc = wg_flush(ctx);
    R->ws_used = used;      // (later records must not reuse slab regions a launch in flight still reads)
    return rc;
  }
  std::vector<WgRec> rest(R->recs.begin() + count, R->recs.end());
  R->recs.resize((size_t)count);
  const size_t used = R->ws_used;
  const int rc = wg_flush(ctx);
  R->recs = rest;
  R->ws_used = used;
  return rc;
}

int vf_internal_wgrad_smallk(vf_ctx* ctx, const float* U, const float* V, float* dW, int K, int Nu, int Ncols, float beta);  // vf_wgrad_small.hip
static int wgrad(vf_ctx* ctx, const float* U, const float* V, float* dW, int B, int Hl, int Wl, int Nu, int Hv, int Wv,
                 int Cv, int stride, int pad, float beta, int ntaps = 16, const void* Up = nullptr, const void* Vp = nullptr) {
  // the bottleneck pair (1x1 map on one side, 4x4 on the other): dW = U^T V with K = batch — write-bound, its own kernel
  if (Hl == 1 && Wl == 1 && Hv == 4 && Wv == 4 && stride == 1 && pad == 0 && ntaps == 16) {
    const int rc = vf_internal_wgrad_smallk(ctx, U, V, dW, B, Nu, 16 * Cv, beta);
    if (rc >= 0) return rc;
  }
  WGrad g;
  memset(&g, 0, sizeof(g));
  g.U = U; g.V = V; g.dW = dW;
  g.P = B * Hl * Wl;
  g.lgMh = vf_ilog2(Hl); g.lgMw = vf_ilog2(Wl);
  g.Nu = Nu; g.Cv = Cv; g.Hv = Hv; g.Wv = Wv;
  g.stride = stride; g.pad = pad;
  g.ntaps = ntaps;
  g.nk = (int)vf_cdiv(g.P, ctx->mfma_bf16 ? 32 : 16);      // K steps of the kernel's BK
  VF_REQUIRE((int64_t)g.P * Nu < ((int64_t)1 << 29) && (int64_t)B * Hv * Wv * Cv < ((int64_t)1 << 29),
             "operand exceeds the 2 GiB buffer-descriptor range");
  g.u_bytes = (unsigned)((int64_t)g.P * Nu * 4);
  g.v_bytes = (unsigned)((int64_t)B * Hv * Wv * Cv * 4);
  const bool vecU = (Nu % 4 == 0) && aligned16(U);
  const bool vecV = (Cv % 4 == 0) && aligned16(V);
  g.beta = beta;
  const int BM = Nu > 64 ? 128 : 64;
  const int gy = (int)vf_cdiv(Nu, BM), gx = (int)vf_cdiv(ntaps * (int64_t)Cv, 128);
  const int64_t blocks = (int64_t)gx * gy;
  const int64_t total = (int64_t)Nu * ntaps * Cv;
  int ksplit = 1;
  // operands that arrive as planes (vf_*_bwd_weight_planes): the LDS-DMA kernel of vf_pgemm.hip, whole 128 x 128 x 32 tiles only
  const bool use_pw = Up && Vp && (ctx->mfma_bf16 == 3 || ctx->mfma_bf16 == 1) && ntaps == 16 && stride == 2 && pad == 1 && Nu % 128 == 0 &&
                      Cv % 64 == 0 && g.P % 32 == 0 && (int64_t)g.P * Nu * 6 < ((int64_t)1 << 31) &&
                      (int64_t)B * Hv * Wv * Cv * 6 < ((int64_t)1 << 31) && total % 4 == 0;
  // split-K target, blocks per layer.  The planes layers ride in ONE group launch of seven or more layers, which fills the chip as
  // a whole: half the splits per layer leave k_pwgrad_group's time where it was and halve the slabs it writes and the combine
  // reads (same-box: combine 2 x 37.4 -> 2 x 21.8 us); the fp32-fed thin layers' kernel wants the full 512 (62 -> 86 us at 256)
  static const int env_wg_blocks = getenv("VF_WGRAD_BLOCKS") ? atoi(getenv("VF_WGRAD_BLOCKS")) : 0;
  static const int env_pwg_blocks = getenv("VF_PWGRAD_BLOCKS") ? atoi(getenv("VF_PWGRAD_BLOCKS")) : 0;
  const int tune_wg_blocks = use_pw ? (env_pwg_blocks ? env_pwg_blocks : (env_wg_blocks ? env_wg_blocks : 256)) : (env_wg_blocks ? env_wg_blocks : 512);
  if (blocks < tune_wg_blocks * 3 / 4 && g.nk >= 16) {
    ksplit = (int)std::min<int64_t>(g.nk / 8, vf_cdiv(tune_wg_blocks, blocks));
    while (ksplit > 1 && (size_t)ksplit * total * sizeof(float) > vf_ws_avail(ctx)) --ksplit;
    if (ksplit < 1) ksplit = 1;
    const int steps = (int)vf_cdiv(g.nk, ksplit);
    ksplit = (int)vf_cdiv(g.nk, steps);
  }
  g.ksplit = ksplit;
  g.slab = ksplit > 1 ? (float*)vf_ws_ptr(ctx) : nullptr;
  g.gx = gx; g.gy = gy; g.gz = ksplit;
  dim3 grid((unsigned)gx * gy * ksplit), block(256);
  const bool own_group = use_pw && !ctx->wg_active;      // a planes layer outside a group: a group of one, launched at once
  if (own_group) {
    if (!ctx->wg_rec) ctx->wg_rec = new WgRecorder();
    ctx->wg_active = 1;
  }
  if (ctx->wg_active && (use_pw || (vecU && vecV && (ksplit == 1 || total % 4 == 0)))) {
    // recorded, not launched: the group runs at vf_wgrad_group_end.  Every recorded layer keeps its own slab region.
    WgRecorder* R = (WgRecorder*)ctx->wg_rec;
    const size_t need = ksplit > 1 ? (((size_t)ksplit * total * sizeof(float) + 255) & ~(size_t)255) : 0;
    if (R->ws_used + need > vf_ws_avail(ctx)) {
      if (int rc = wg_flush(ctx)) return rc;
    }
    VF_REQUIRE(need <= vf_ws_avail(ctx), "workspace too small for this layer's split-K slabs");
    if (ksplit > 1) g.slab = (float*)(vf_ws_ptr(ctx) + R->ws_used);
    R->ws_used += need;
    WgRec r;
    r.g = g;
    r.bm = BM;
    r.bf = ctx->mfma_bf16;
    r.blocks = gx * gy * ksplit;
    r.total = total;
    r.flops = 2.0 * (double)g.P * Nu * (double)ntaps * Cv;
    r.plain_ok = false;
    if (!use_pw && (ctx->mfma_bf16 == 3 || ctx->mfma_bf16 == 1) && BM == 128 && ksplit == 1 && Hl == 1 && Wl == 1 && Hv == 4 && Wv == 4 && stride == 1 &&
        pad == 0 && ntaps == 16 && Nu % 4 == 0 && (16 * Cv) % 128 == 0 && vecU && vecV) {
      r.plain_ok = true;
      VfPWGrad& w = r.pw;
      memset(&w, 0, sizeof(w));
      w.Uf = U; w.Vf = V;                  // U [B][Nu], V [B][4][4][Cv] = [B][16 * Cv]
      w.out = dW;
      w.P = g.P; w.Nu = Nu; w.Cv = Cv; w.Hv = Hv; w.Wv = Wv;
      w.gx = (16 * Cv) / 128; w.gy = (int)vf_cdiv(Nu, 128); w.gz = 1;
      w.ksplit = 1; w.nk = (int)vf_cdiv(g.P, 32);
      w.beta = beta;
    }
    if (use_pw) {
      r.bf = 4;
      VfPWGrad& w = r.pw;
      memset(&w, 0, sizeof(w));
      w.Up = Up; w.Vp = Vp;
      w.out = ksplit > 1 ? g.slab : dW;
      w.u_ps = (unsigned)((int64_t)g.P * Nu * 2);
      w.v_ps = (unsigned)((int64_t)B * Hv * Wv * Cv * 2);
      w.P = g.P; w.lgMh = g.lgMh; w.lgMw = g.lgMw;
      w.Nu = Nu; w.Cv = Cv; w.Hv = Hv; w.Wv = Wv;
      w.gx = gx; w.gy = gy; w.gz = ksplit;
      w.ksplit = ksplit; w.nk = g.nk;
      w.beta = beta;
    }
    R->recs.push_back(r);
    if (own_group) {
      ctx->wg_active = 0;
      return wg_flush(ctx);
    }
    return 0;
  }
  if (ctx->wg_active) {
    // a launch outside the group must not overwrite recorded slabs: its own slabs go BEHIND them (the group stays recorded: a
    // thin-channel layer at the end of a walk no longer launches the group ahead of vf_wgrad_group_end / _end_partial); only if
    // the workspace cannot hold both is the group finished first
    WgRecorder* R = (WgRecorder*)ctx->wg_rec;
    const size_t need = ksplit > 1 ? (((size_t)ksplit * total * sizeof(float) + 255) & ~(size_t)255) : 0;
    if (R && R->ws_used + need <= vf_ws_avail(ctx)) {
      if (ksplit > 1) g.slab = (float*)(vf_ws_ptr(ctx) + R->ws_used);
      R->ws_used += need;
    } else if (int rc = wg_flush(ctx)) {
      return rc;
    }
  }
  {
    const int bf = ctx->mfma_bf16;
    const char* wname = BM == 128 ? (bf == 3 ? "wgrad_128x128_bf16x3" : bf ? "wgrad_128x128_bf16" : "wgrad_128x128")
                                  : (bf == 3 ? "wgrad_64x128_bf16x3" : bf ? "wgrad_64x128_bf16" : "wgrad_64x128");
    const double wfl = 2.0 * (double)g.P * Nu * (double)ntaps * Cv;
#define VF_WG(BM_, BF_)                                                                              \
  do {                                                                                              \
    if (vecU && vecV)                                                                               \
      VF_LAUNCH_TIMED(ctx, wname, wfl, 0.0, (k_wgrad<BM_, true, true, BF_>), grid, block, g);       \
    else if (vecU)                                                                                  \
      VF_LAUNCH_TIMED(ctx, wname, wfl, 0.0, (k_wgrad<BM_, true, false, BF_>), grid, block, g);      \
    else if (vecV)                                                                                  \
      VF_LAUNCH_TIMED(ctx, wname, wfl, 0.0, (k_wgrad<BM_, false, true, BF_>), grid, block, g);      \
    else                                                                                            \
      VF_LAUNCH_TIMED(ctx, wname, wfl, 0.0, (k_wgrad<BM_, false, false, BF_>), grid, block, g);     \
  } while (0)
    if (BM == 128) {
      if (bf == 3) VF_WG(128, 3); else if (bf) VF_WG(128, 1); else VF_WG(128, 0);
    } else {
      if (bf == 3) VF_WG(64, 3); else if (bf) VF_WG(64, 1); else VF_WG(64, 0);
    }
#undef VF_WG
  }
  VF_LAUNCH_CHECK();
  if (ksplit > 1) {
    VfProf prof(ctx, "slab_reduce_wgrad", 0.0, 4.0 * (double)total * (ksplit + 1));
    return launch_slab_reduce(ctx, g.slab, dW, nullptr, total, 1, ksplit, VF_ACT_NONE, 0.f, beta);
  }
  return 0;
}

int vf_internal_colsum(vf_ctx* ctx, const float* g, float* gb, int64_t P, int C, float beta);  // vf_bn.hip
static int bias_grad(vf_ctx* ctx, const float* g, float* gb, int64_t P, int C, float beta) {
  return vf_internal_colsum(ctx, g, gb, P, C, beta);
}

// ------------------------------------------------------------------------------------------------ C ABI
// The option branches' convolutions (5x5 stride 2 pad 2 / 34, 1x1: train.lua:109-113,158-170) go to vf_conv_generic.hip.
int vf_internal_gconv_fwd(vf_ctx* ctx, const float* x, const float* w, const float* bias, float* y, int B, int H, int W, int Cin,
                          int Cout, int k, int stride, int pad, int act, float slope);
int vf_internal_gconv_bwd_data(vf_ctx* ctx, const float* gy, const float* w, float* gx, int B, int H, int W, int Cin, int Cout,
                               int k, int stride, int pad);
int vf_internal_gconv_bwd_weight(vf_ctx* ctx, const float* x, const float* gy, float* gw, float* gb, int B, int H, int W, int Cin,
                                 int Cout, int k, int stride, int pad, float beta);
static bool main_net_shape(int H, int W, int k, int stride, int pad) {
  return k == 4 && ((stride == 2 && pad == 1) || (stride == 1 && pad == 0 && H == 4 && W == 4)) && vf_is_pow2(H) && vf_is_pow2(W);
}

// 1 if this geometry runs on the matrix-core kernels, 0 if vf_conv_generic.hip serves it (include/vf_hip.h, "Shapes")
VF_API int vf_conv_is_fast(int H, int W, int k, int stride, int pad) { return main_net_shape(H, W, k, stride, pad) ? 1 : 0; }

int vf_internal_conv_thin_fwd(vf_ctx* ctx, const float* x, const float* w, const float* bias, float* y, void* y_planes, int B, int H,
                              int W, int Cin, int Cout, int act, float slope);      // vf_conv_thin.hip
VF_API int vf_conv2d_fwd(vf_ctx* ctx, const float* x, const float* w, const float* bias, float* y, int B, int H, int W,
                         int Cin, int Cout, int k, int stride, int pad, int act, float slope) {
  if (!main_net_shape(H, W, k, stride, pad)) return vf_internal_gconv_fwd(ctx, x, w, bias, y, B, H, W, Cin, Cout, k, stride, pad, act, slope);
  if (int rc = check_conv_args(B, H, W, Cin, Cout, k, stride, pad)) return rc;
  if (stride == 2 && Cin == 3 && !ctx->bnf.mode) {      // the image-side layers: direct convolution (vf_conv_thin.hip)
    static const int no_thin = getenv("VF_NO_THIN") ? 1 : 0;
    if (!no_thin) {
      const int rc = vf_internal_conv_thin_fwd(ctx, x, w, bias, y, nullptr, B, H, W, Cin, Cout, act, slope);
      if (rc >= 0) return rc;
    }
  }
  return conv_like_fwd(ctx, x, w, bias, y, B, H, W, Cin, Cout, stride, pad, act, slope);
}

VF_API int vf_conv2d_bwd_data(vf_ctx* ctx, const float* gy, const float* w, float* gx, int B, int H, int W, int Cin,
                              int Cout, int k, int stride, int pad) {
  if (!main_net_shape(H, W, k, stride, pad)) return vf_internal_gconv_bwd_data(ctx, gy, w, gx, B, H, W, Cin, Cout, k, stride, pad);
  if (int rc = check_conv_args(B, H, W, Cin, Cout, k, stride, pad)) return rc;
  const int Ho = (H + 2 * pad - 4) / stride + 1, Wo = (W + 2 * pad - 4) / stride + 1;
  if (stride == 1) {
    VF_REQUIRE(H == 4 && W == 4, "stride-1 conv data-grad is built for the 4x4 bottleneck input only");
    if (Cout == 1) {
      const int64_t n = (int64_t)B * 16 * Cin;
      const float* ya = ctx->dot_act_y;      // one-shot: the derivative of the activation fused into this conv (vf_net.hip)
      ctx->dot_act_y = nullptr;
      hipLaunchKernelGGL(k_dot_bwd_data, dim3((int)vf_cdiv(n, 256)), dim3(256), 0, ctx->stream, gy, w, gx, B, 16 * Cin, ya, ctx->dot_act,
                         ctx->dot_act_slope);
      VF_LAUNCH_CHECK();
      return 0;
    }
  }
  return conv_like_bwd(ctx, gy, w, nullptr, gx, B, Ho, Wo, Cout, Cin, stride, pad, VF_ACT_NONE, 0.f);
}

VF_API int vf_conv2d_bwd_data_act(vf_ctx* ctx, const float* gy, const float* w, float* gx, const float* x_act, int act,
                                  float slope, int B, int H, int W, int Cin, int Cout, int k, int stride, int pad) {
  if (int rc = check_conv_args(B, H, W, Cin, Cout, k, stride, pad)) return rc;
  VF_REQUIRE(x_act != nullptr && (act == VF_ACT_LRELU || act == VF_ACT_RELU),
             "vf_conv2d_bwd_data_act: needs the activated input and a (leaky) ReLU");
  VF_REQUIRE(stride == 2, "vf_conv2d_bwd_data_act: only the stride-2 layers follow a bare conv + activation pair");
  const int Ho = (H + 2 * pad - 4) / stride + 1, Wo = (W + 2 * pad - 4) / stride + 1;
  return conv_like_bwd(ctx, gy, w, nullptr, gx, B, Ho, Wo, Cout, Cin, stride, pad, VF_ACT_NONE, 0.f, x_act, act, slope);
}

static int conv2d_bwd_weight_impl(vf_ctx* ctx, const float* x, const float* gy, const void* x_planes, const void* gy_planes,
                                  float* gw, float* gb, int B, int H, int W, int Cin, int Cout, int k, int stride, int pad, float beta);
VF_API int vf_conv2d_bwd_weight(vf_ctx* ctx, const float* x, const float* gy, float* gw, float* gb, int B, int H, int W,
                                int Cin, int Cout, int k, int stride, int pad, float beta) {
  return conv2d_bwd_weight_impl(ctx, x, gy, nullptr, nullptr, gw, gb, B, H, W, Cin, Cout, k, stride, pad, beta);
}
// the same with the bf16 planes of both operands at hand (vf_planes_split layout): the weight gradient then runs on the
// planes-fed kernel where its shape allows (whole 128 x 128 x 32 tiles), on the fp32 operands otherwise
VF_API int vf_conv2d_bwd_weight_planes(vf_ctx* ctx, const float* x, const float* gy, const void* x_planes, const void* gy_planes,
                                       float* gw, float* gb, int B, int H, int W, int Cin, int Cout, int k, int stride, int pad,
                                       float beta) {
  return conv2d_bwd_weight_impl(ctx, x, gy, x_planes, gy_planes, gw, gb, B, H, W, Cin, Cout, k, stride, pad, beta);
}
static int conv2d_bwd_weight_impl(vf_ctx* ctx, const float* x, const float* gy, const void* x_planes, const void* gy_planes,
                                  float* gw, float* gb, int B, int H, int W, int Cin, int Cout, int k, int stride, int pad, float beta) {
  if (!main_net_shape(H, W, k, stride, pad))
    return vf_internal_gconv_bwd_weight(ctx, x, gy, gw, gb, B, H, W, Cin, Cout, k, stride, pad, beta);
  if (int rc = check_conv_args(B, H, W, Cin, Cout, k, stride, pad)) return rc;
  const int Ho = (H + 2 * pad - 4) / stride + 1, Wo = (W + 2 * pad - 4) / stride + 1;
  if (Cout == 1 && stride == 1 && H == 4 && W == 4) {
    const float* ya = ctx->dot_act_y;      // (see vf_conv2d_bwd_data)
    ctx->dot_act_y = nullptr;
    hipLaunchKernelGGL(k_dot_bwd_weight, dim3((int)vf_cdiv(16 * Cin, 64)), dim3(512), 0, ctx->stream, x, gy, gw, gb, B,
                       16 * Cin, beta, ya, ctx->dot_act, ctx->dot_act_slope);
    VF_LAUNCH_CHECK();
    return 0;
  }
  // (tried: the bias gradient as a by-product of this kernel's own gy fragments — the waves that carried it made
  //  their blocks the slowest of every launch: +0.25 ms per step against the 0.19 ms of the separate column sums)
  if (int rc = wgrad(ctx, gy, x, gw, B, Ho, Wo, Cout, H, W, Cin, stride, pad, beta, 16, gy_planes, x_planes)) return rc;
  if (gb) return bias_grad(ctx, gy, gb, (int64_t)B * Ho * Wo, Cout, beta);
  return 0;
}

VF_API int vf_deconv2d_fwd(vf_ctx* ctx, const float* x, const float* w, const float* bias, float* y, int B, int H, int W,
                           int Cin, int Cout, int k, int stride, int pad, int act, float slope) {
  VF_REQUIRE(k == 4 && ((stride == 2 && pad == 1) || (stride == 1 && pad == 0)), "unsupported full-conv shape");
  VF_REQUIRE(vf_is_pow2(H) && vf_is_pow2(W), "spatial sizes must be powers of two");
  return conv_like_bwd(ctx, x, w, bias, y, B, H, W, Cin, Cout, stride, pad, act, slope);
}

VF_API int vf_deconv2d_bwd_data(vf_ctx* ctx, const float* gy, const float* w, float* gx, int B, int H, int W, int Cin,
                                int Cout, int k, int stride, int pad) {
  VF_REQUIRE(k == 4 && ((stride == 2 && pad == 1) || (stride == 1 && pad == 0)), "unsupported full-conv shape");
  const int Ho = (H - 1) * stride - 2 * pad + 4, Wo = (W - 1) * stride - 2 * pad + 4;
  VF_REQUIRE(vf_is_pow2(H) && vf_is_pow2(W), "spatial sizes must be powers of two");
  // conv of gy (Ho x Wo, Cout channels) with weights [Cin][kh][kw][Cout] -> gx (H x W, Cin channels)
  return conv_like_fwd(ctx, gy, w, nullptr, gx, B, Ho, Wo, Cout, Cin, stride, pad, VF_ACT_NONE, 0.f);
}

static int deconv2d_bwd_weight_impl(vf_ctx* ctx, const float* x, const float* gy, const void* x_planes, const void* gy_planes,
                                    float* gw, float* gb, int B, int H, int W, int Cin, int Cout, int k, int stride, int pad,
                                    float beta) {
  VF_REQUIRE(k == 4 && ((stride == 2 && pad == 1) || (stride == 1 && pad == 0)), "unsupported full-conv shape");
  VF_REQUIRE(vf_is_pow2(H) && vf_is_pow2(W), "spatial sizes must be powers of two");
  const int Ho = (H - 1) * stride - 2 * pad + 4, Wo = (W - 1) * stride - 2 * pad + 4;
  // gw[ci][kh][kw][co] = sum_{b,i,j} x[b,i,j,ci] * gy[b, i*s-pad+kh, j*s-pad+kw, co]
  if (int rc = wgrad(ctx, x, gy, gw, B, H, W, Cin, Ho, Wo, Cout, stride, pad, beta, 16, x_planes, gy_planes)) return rc;
  if (gb) return bias_grad(ctx, gy, gb, (int64_t)B * Ho * Wo, Cout, beta);
  return 0;
}
VF_API int vf_deconv2d_bwd_weight(vf_ctx* ctx, const float* x, const float* gy, float* gw, float* gb, int B, int H, int W,
                                  int Cin, int Cout, int k, int stride, int pad, float beta) {
  return deconv2d_bwd_weight_impl(ctx, x, gy, nullptr, nullptr, gw, gb, B, H, W, Cin, Cout, k, stride, pad, beta);
}
VF_API int vf_deconv2d_bwd_weight_planes(vf_ctx* ctx, const float* x, const float* gy, const void* x_planes, const void* gy_planes,
                                         float* gw, float* gb, int B, int H, int W, int Cin, int Cout, int k, int stride, int pad,
                                         float beta) {
  return deconv2d_bwd_weight_impl(ctx, x, gy, x_planes, gy_planes, gw, gb, B, H, W, Cin, Cout, k, stride, pad, beta);
}
