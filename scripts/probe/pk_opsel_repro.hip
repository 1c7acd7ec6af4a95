// PROBE — a stand-alone attempt to reproduce, outside the library, the packed-FMA fault of DESIGN.md 4.9:
//   v_pk_fma_f32 D, A, B, C op_sel:[0,1,0]   (the LOW result takes the HIGH dword of src1)
// occasionally leaving the low dword of an in-place accumulator unwritten in lanes 48-63 when other processes share the CU.
// Each wave runs the instruction pattern of the failing row-dot kernel — two interleaved accumulator pairs, four steps per
// round: e0 / e2 / e3 with op_sel_hi:[1,0,1] (x.lo or x.hi' broadcast the safe way), e1 with op_sel:[0,1,0] — from inline asm, so
// the forms are exactly these whatever the compiler would choose, and beside it the same arithmetic with scalar v_fma_f32 on
// separate registers.  The two are compared bit for bit every round; mismatches are logged with lane and half.
//   VARIANT 0: e1, e3 = op_sel:[0,1,0] (the suspect form)      VARIANT 1: e1, e3 with src0 / src1 exchanged, op_sel:[1,0,0] (the control)
//   hipcc --offload-arch=gfx950 -O2 -DVARIANT=0 scripts/probe/pk_opsel_repro.hip -o scripts/probe/pk_opsel_repro
//   ./pk_opsel_repro [seconds]          (run it beside busy neighbours: scripts/probe/pk_opsel_repro.sh)
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#ifndef VARIANT
#define VARIANT 0
#endif
#ifndef UNROLL
#define UNROLL 1      // rounds unrolled into straight-line code: 512 makes the kernel ~10x the instruction cache (fetch stalls between the
#endif                // packed instructions even with the GPU to itself)

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct Mis {
  int block, wave, lane, round, chain, half;
  float want, got;
};

__global__ __launch_bounds__(256) void k_repro(const float* __restrict__ wsrc, const float* __restrict__ xsrc, int rounds, Mis* log,
                                               int* nlog, unsigned long long* checked, int* per_cu = nullptr) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63;
  // per-lane operands: 4 steps x 2 chains of (w pair), 2 x pairs per round, reloaded every round from a table that differs per lane
  f32x2 accA = {0.f, 0.f}, accB = {0.f, 0.f};
  float rA0 = 0.f, rA1 = 0.f, rB0 = 0.f, rB1 = 0.f;
  unsigned long long n = 0;
  for (int r0 = 0; r0 < rounds; r0 += UNROLL)
#pragma unroll
  for (int u = 0; u < UNROLL; ++u) {
    const int r = r0 + u;
    const int base = ((tid * 131 + r * 17) & 4095) * 16;
    f32x2 w[4], v[4], x0, x1;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      w[k] = *(const f32x2*)(wsrc + base + 2 * k);
      v[k] = *(const f32x2*)(wsrc + base + 8 + 2 * k);
    }
    x0 = *(const f32x2*)(xsrc + base);
    x1 = *(const f32x2*)(xsrc + base + 2);
    // ---- packed chains (in place), the two chains interleaved as in the failing kernel
    asm volatile(
        "v_pk_fma_f32 %0, %2, %10, %0 op_sel_hi:[1,0,1]\n\t"      // A e0: x0.lo for both halves
        "v_pk_fma_f32 %1, %6, %10, %1 op_sel_hi:[1,0,1]\n\t"      // B e0
#if VARIANT == 0
        "v_pk_fma_f32 %0, %3, %10, %0 op_sel:[0,1,0]\n\t"         // A e1: x0.hi for both halves — the suspect form
        "v_pk_fma_f32 %1, %7, %10, %1 op_sel:[0,1,0]\n\t"         // B e1
#else
        "v_pk_fma_f32 %0, %10, %3, %0 op_sel:[1,0,0]\n\t"         // A e1, operands exchanged: the high-dword select on src0
        "v_pk_fma_f32 %1, %10, %7, %1 op_sel:[1,0,0]\n\t"
#endif
        "v_pk_fma_f32 %0, %4, %11, %0 op_sel_hi:[1,0,1]\n\t"      // A e2: x1.lo
        "v_pk_fma_f32 %1, %8, %11, %1 op_sel_hi:[1,0,1]\n\t"
        "s_nop 0\n\t"
#if VARIANT == 0
        "v_pk_fma_f32 %0, %5, %11, %0 op_sel:[0,1,0]\n\t"         // A e3: x1.hi (second instance of the suspect form per round)
        "v_pk_fma_f32 %1, %9, %11, %1 op_sel:[0,1,0]\n\t"
#else
        "v_pk_fma_f32 %0, %11, %5, %0 op_sel:[1,0,0]\n\t"         // A e3, operands exchanged
        "v_pk_fma_f32 %1, %11, %9, %1 op_sel:[1,0,0]\n\t"
#endif
        "s_nop 1"
        : "+v"(accA), "+v"(accB)
        : "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]), "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(x0), "v"(x1));
    // ---- the same arithmetic, scalar
    rA0 = fmaf(w[0][0], x0[0], rA0); rA1 = fmaf(w[0][1], x0[0], rA1);
    rB0 = fmaf(v[0][0], x0[0], rB0); rB1 = fmaf(v[0][1], x0[0], rB1);
    rA0 = fmaf(w[1][0], x0[1], rA0); rA1 = fmaf(w[1][1], x0[1], rA1);
    rB0 = fmaf(v[1][0], x0[1], rB0); rB1 = fmaf(v[1][1], x0[1], rB1);
    rA0 = fmaf(w[2][0], x1[0], rA0); rA1 = fmaf(w[2][1], x1[0], rA1);
    rB0 = fmaf(v[2][0], x1[0], rB0); rB1 = fmaf(v[2][1], x1[0], rB1);
    rA0 = fmaf(w[3][0], x1[1], rA0); rA1 = fmaf(w[3][1], x1[1], rA1);
    rB0 = fmaf(v[3][0], x1[1], rB0); rB1 = fmaf(v[3][1], x1[1], rB1);
    const float got[4] = {accA[0], accA[1], accB[0], accB[1]}, want[4] = {rA0, rA1, rB0, rB1};
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (__float_as_uint(got[q]) != __float_as_uint(want[q])) {
        const int i = atomicAdd(nlog, 1);
        if (per_cu) {       // where the wave runs: XCC_ID (hwreg 20) bits 3:0; HW_ID (hwreg 4): cu_id 11:8, sh_id 12, se_id 15:13
          const unsigned xcc = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20), hw = __builtin_amdgcn_s_getreg((16 - 1) << 11 | 0 << 6 | 4);
          atomicAdd(per_cu + (xcc & 7) * 128 + ((hw >> 13) & 7) * 16 + ((hw >> 8) & 15), 1);
        }
        if (i < 256) log[i] = Mis{(int)blockIdx.x, (int)(threadIdx.x >> 6), lane, r, q >> 1, q & 1, want[q], got[q]};
      }
    // resynchronise (a fault must not be counted again every round) and keep the magnitudes bounded
    accA[0] = rA0 = rA0 * 0.5f; accA[1] = rA1 = rA1 * 0.5f;
    accB[0] = rB0 = rB0 * 0.5f; accB[1] = rB1 = rB1 * 0.5f;
    n += 4;
  }
  if (lane == 0) atomicAdd(checked, n * 64ull);
}

// ---- optional neighbours in the SAME process, on a second stream (argv[2]: 1 matrix cores, 2 LDS, 3 global memory, 4 plain VALU, 5 the
//      library's own implicit GEMM — E3's forward pass through the C-ABI of video-filler_amd/lib/libvf_hip.so, the kernel family that
//      triggers the fault from ANOTHER process): which unit's traffic on the CU does the fault need, and does it need another process?
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void k_nb_mfma(float* out, int iters) {
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (threadIdx.x + i)); b[i] = (__bf16)(0.002f * (threadIdx.x - i)); }
  f32x16 acc = {};
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 16; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
  }
  if (acc[0] == 12345.f) out[0] = acc[1];
}
__global__ __launch_bounds__(256) void k_nb_lds(float* out, int iters) {
  __shared__ float sh[4096];
  for (int i = threadIdx.x; i < 4096; i += 256) sh[i] = i;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      t += sh[(threadIdx.x * 17 + i + 61 * j) & 4095];
      sh[(threadIdx.x * 5 + i + 97 * j) & 4095] = t;
    }
  }
  if (t == 12345.f) out[0] = t;
}
__global__ __launch_bounds__(256) void k_nb_vmem(const float4* __restrict__ src, float4* __restrict__ dst, size_t n4, int iters) {
  for (int it = 0; it < iters; ++it)
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
// 6: bf16 MFMAs fed from LDS with a barrier per step (a GEMM's inner loop without the global side)
__global__ __launch_bounds__(256) void k_nb_mfma_lds(float* out, int iters) {
  __shared__ __attribute__((aligned(16))) __bf16 sh[2 * 64 * 64];
  for (int i = threadIdx.x; i < 2 * 64 * 64; i += 256) sh[i] = (__bf16)(0.001f * (i & 63));
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f32x16 acc = {};
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const bf16x8 a = *(const bf16x8*)(sh + ((wave & 1) * 32 + (lane & 31)) * 64 + 16 * g + 8 * (lane >> 5));
      const bf16x8 b = *(const bf16x8*)(sh + 64 * 64 + ((wave >> 1) * 32 + (lane & 31)) * 64 + 16 * g + 8 * (lane >> 5));
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, acc, 0, 0, 0);
    }
    __syncthreads();
  }
  if (acc[0] == 12345.f) out[0] = acc[1];
}
// 7: the transposing LDS read of gfx950 (ds_read_b64_tr_b16), 8: the plane-split VALU mix (v_perm_b32 + subtractions),
// 9: bf16 MFMAs with independent VALU work between them (matrix-core / VALU co-execution)
__global__ __launch_bounds__(256) void k_nb_trread(float* out, int iters) {
  __shared__ __attribute__((aligned(16))) __bf16 sh[64 * 64];
  for (int i = threadIdx.x; i < 64 * 64; i += 256) sh[i] = (__bf16)(0.001f * (i & 63));
  __syncthreads();
  unsigned addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)sh + (threadIdx.x & 63) * 8;
  unsigned long long t = 0, v;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr), "n"(0));
      t += v;
    }
  }
  if (t == 12345ull) out[0] = 1.f;
}
__global__ __launch_bounds__(256) void k_nb_split(float* out, int iters) {
  float r0 = threadIdx.x * 0.37f + 1.f, r1 = r0 * 1.7f;
  unsigned acc = 0;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const unsigned u0 = __float_as_uint(r0), u1 = __float_as_uint(r1);
      acc ^= __builtin_amdgcn_perm(u1, u0, 0x07060302u);
      r0 -= __uint_as_float(u0 & 0xffff0000u);
      r1 -= __uint_as_float(u1 & 0xffff0000u);
      r0 = r0 * 3.1f + 0.77f;
      r1 = r1 * 2.9f + 0.31f;
    }
  }
  if (acc == 12345u) out[0] = r0;
}
__global__ __launch_bounds__(256) void k_nb_coexec(float* out, int iters) {
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (threadIdx.x + i)); b[i] = (__bf16)(0.002f * (threadIdx.x - i)); }
  f32x16 acc = {};
  float x = threadIdx.x * 0.01f, y = 1.0001f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
      x = fmaf(x, y, 0.5f); x = fmaf(x, y, 0.25f); x = fmaf(x, y, 0.125f); x = fmaf(x, y, 0.0625f);
    }
  }
  if (acc[0] == 12345.f || x == 12345.f) out[0] = acc[1];
}
// 11+: a GEMM-like loop assembled from pieces (bit mask `what`: 1 v_perm split VALU, 2 ds_write_b64, 4 ds_read_b128 + bf16 MFMAs, 8 a barrier
// on either side of the LDS phase) — which combination does it take?
__global__ __launch_bounds__(256) void k_nb_combo(float* out, int iters, int what) {
  __shared__ __attribute__((aligned(16))) __bf16 sh[3 * 128 * 32];
  for (int i = threadIdx.x; i < 3 * 128 * 32; i += 256) sh[i] = (__bf16)(0.001f * (i & 63));
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f32x16 acc = {};
  float r0 = threadIdx.x * 0.37f + 1.f, r1 = r0 * 1.7f, r2 = r0 * 0.3f, r3 = r1 * 0.9f;
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  for (int i = 0; i < iters; ++i) {
    u32x2 pl[3] = {};
    if (what & 1) {
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const unsigned u0 = __float_as_uint(r0), u1 = __float_as_uint(r1), u2 = __float_as_uint(r2), u3 = __float_as_uint(r3);
        pl[q][0] = __builtin_amdgcn_perm(u1, u0, 0x07060302u);
        pl[q][1] = __builtin_amdgcn_perm(u3, u2, 0x07060302u);
        r0 -= __uint_as_float(u0 & 0xffff0000u); r1 -= __uint_as_float(u1 & 0xffff0000u);
        r2 -= __uint_as_float(u2 & 0xffff0000u); r3 -= __uint_as_float(u3 & 0xffff0000u);
      }
      r0 = r0 * 3.1f + 0.77f; r1 = r1 * 2.9f + 0.31f; r2 = r2 * 1.3f + 0.11f; r3 = r3 * 0.7f + 0.05f;
    }
    if (what & 8) __syncthreads();
    if (what & 2) {
#pragma unroll
      for (int q = 0; q < 3; ++q) *(u32x2*)(sh + q * 128 * 32 + (threadIdx.x >> 3) * 32 + 4 * (threadIdx.x & 7)) = pl[q];
    }
    if (what & 8) __syncthreads();
    if (what & 4) {
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        bf16x8 a[3], b[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          a[q] = *(const bf16x8*)(sh + q * 128 * 32 + ((wave & 1) * 32 + (lane & 31)) * 32 + 16 * g + 8 * (lane >> 5));
          b[q] = *(const bf16x8*)(sh + q * 128 * 32 + (64 + (wave >> 1) * 32 + (lane & 31)) * 32 + 16 * g + 8 * (lane >> 5));
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc, 0, 0, 0);
      }
    }
  }
  if (acc[0] == 12345.f || r0 == 12345.f) out[0] = acc[1] + r1 + r2 + r3;
}
// 11: nothing but a workgroup that HOLDS KB kilobytes of LDS (gfx950 allows 160 KB per workgroup; up to gfx942 the limit was 64) and
// reads/writes a few words of it, at the bottom (HIGH = 0) or at the top (HIGH = 1) of the allocation
template <int KB, int HIGH>
__global__ __launch_bounds__(256) void k_nb_biglds(float* out, int iters) {
  __shared__ float sh[KB * 256];
  const int at = (HIGH ? KB * 256 - 512 : 0) + threadIdx.x;
  sh[at] = threadIdx.x;
  float a = 0.f;
  for (int i = 0; i < iters; ++i) {
    a += sh[at ^ (i & 63)];
    sh[at] = a * 0.5f;
    __builtin_amdgcn_s_sleep(4);
  }
  if (a == 12345.f) out[0] = a;
}
// 200 + mask: the schedule the ablation of the library's double-buffered GEMM points at (profiles/r04_pk_opsel_reproducer.txt): work issued
// in the SHADOW of a running MFMA, pinned there with scheduling barriers.  Mask: 1 the v_perm/v_sub plane split, 2 a ds_write_b64 of the
// result, 4 fragments re-read from LDS (ds_read_b128) every round, 8 a barrier per round
template <int what>
__global__ __launch_bounds__(256) void k_nb_shadow(float* out, int iters) {
  __shared__ __attribute__((aligned(16))) __bf16 sh[2 * 3 * 128 * 32];
  for (int i = threadIdx.x; i < 2 * 3 * 128 * 32; i += 256) sh[i] = (__bf16)(0.001f * (i & 63));
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  f32x16 acc = {};
  float r0 = threadIdx.x * 0.37f + 1.f, r1 = r0 * 1.7f, r2 = r0 * 0.3f, r3 = r1 * 0.9f;
  bf16x8 a[3], b[3];
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    a[q] = *(const bf16x8*)(sh + q * 128 * 32 + ((wave & 1) * 32 + (lane & 31)) * 32 + 8 * (lane >> 5));
    b[q] = *(const bf16x8*)(sh + q * 128 * 32 + (64 + (wave >> 1) * 32 + (lane & 31)) * 32 + 8 * (lane >> 5));
  }
  for (int i = 0; i < iters; ++i) {
    __bf16* wr = sh + (1 - (i & 1)) * 3 * 128 * 32;      // writes go to the buffer nobody reads this round
    const __bf16* rd = sh + (i & 1) * 3 * 128 * 32;
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < 6; ++t) {
      constexpr int qa[6] = {1, 0, 2, 0, 1, 0}, qb[6] = {1, 2, 0, 1, 0, 0};
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[qa[t]], b[qb[t]], acc, 0, 0, 0);
      if (t >= 3) {        // one piece per MFMA slot, as the GEMM does it
        u32x2 pl[3] = {};
        if (what & 1) {
#pragma unroll
          for (int q = 0; q < 3; ++q) {
            const unsigned u0 = __float_as_uint(r0), u1 = __float_as_uint(r1), u2 = __float_as_uint(r2), u3 = __float_as_uint(r3);
            pl[q][0] = __builtin_amdgcn_perm(u1, u0, 0x07060302u);
            pl[q][1] = __builtin_amdgcn_perm(u3, u2, 0x07060302u);
            r0 -= __uint_as_float(u0 & 0xffff0000u); r1 -= __uint_as_float(u1 & 0xffff0000u);
            r2 -= __uint_as_float(u2 & 0xffff0000u); r3 -= __uint_as_float(u3 & 0xffff0000u);
          }
          r0 = r0 * 3.1f + 0.77f; r1 = r1 * 2.9f + 0.31f; r2 = r2 * 1.3f + 0.11f; r3 = r3 * 0.7f + 0.05f;
        }
        if (what & 2) {
#pragma unroll
          for (int q = 0; q < 3; ++q) *(u32x2*)(wr + q * 128 * 32 + ((t - 3) * 32 + (threadIdx.x >> 3)) * 32 + 4 * (threadIdx.x & 7)) = pl[q];
        }
      }
      if (t == 0 && (what & 4)) {
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          a[q] = *(const bf16x8*)(rd + q * 128 * 32 + ((wave & 1) * 32 + (lane & 31)) * 32 + 8 * (lane >> 5));
          b[q] = *(const bf16x8*)(rd + q * 128 * 32 + (64 + (wave >> 1) * 32 + (lane & 31)) * 32 + 8 * (lane >> 5));
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (what & 8) __syncthreads();
  }
  if (acc[0] == 12345.f || r0 == 12345.f) out[0] = acc[1] + r1 + r2 + r3;
}
// 300 + form: the smallest trigger found — one MFMA and one LDS store per round, nothing else.  form 0/1/2: ds_write_b32 / b64 / b128 right
// behind the MFMA in the same wave; 3: waves 0-1 issue only the MFMAs, waves 2-3 only the ds_write_b64 (other SIMDs of the CU);
// 4: the ds_write_b64 alone; 5: a 16x16x32 MFMA with the ds_write_b64; 6: an fp32 MFMA (32x32x2) with the ds_write_b64
template <int FORM>
__global__ __launch_bounds__(256) void k_nb_min(float* out, int iters) {
  __shared__ __attribute__((aligned(16))) float sh[256 * 4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f32x16 acc = {};
  f32x4 acc4 = {};
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(0.01f * (lane + j)); b[j] = (__bf16)(0.02f * (lane - j)); }
  const unsigned at = threadIdx.x * 16;
  sh[threadIdx.x] = 0.f;
  f32x4 val = {1.f * lane, 2.f, 3.f, 4.f};
  const bool do_mfma = FORM != 4 && (FORM != 3 || wave < 2), do_write = FORM != 3 || wave >= 2;
  for (int i = 0; i < iters; ++i) {
    __builtin_amdgcn_sched_barrier(0);
    if (do_mfma) {
      if constexpr (FORM == 5) acc4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc4, 0, 0, 0);
      else if constexpr (FORM == 6) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(val[0], val[1], acc, 0, 0, 0);
      else acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    }
    if (do_write) {       // (inline assembly: a volatile C++ store through a generic pointer becomes flat_store; sh is the only LDS object, at 0)
      if constexpr (FORM == 0) asm volatile("ds_write_b32 %0, %1" ::"v"(at), "v"(val[0]) : "memory");
      else if constexpr (FORM == 2) asm volatile("ds_write_b128 %0, %1" ::"v"(at), "v"(val) : "memory");
      else asm volatile("ds_write_b64 %0, %1" ::"v"(at), "v"(f32x2{val[0], val[1]}) : "memory");
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  if (acc[0] == 12345.f || acc4[0] == 12345.f) out[0] = acc[1] + acc4[1] + sh[lane];
}
// 10: bf16 MFMAs whose accumulator lives in AGPRs (as the library's GEMM kernels' do)
__global__ __launch_bounds__(256) void k_nb_mfma_agpr(float* out, int iters) {
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (threadIdx.x + i)); b[i] = (__bf16)(0.002f * (threadIdx.x - i)); }
  f32x16 acc = {};
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 16; ++j) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
  }
  if (acc[0] == 12345.f) out[0] = acc[1];
}
__global__ __launch_bounds__(256) void k_nb_valu(float* out, int iters) {
  float a = threadIdx.x * 0.001f, b = 1.0001f, c = 0.5f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 32; ++j) a = fmaf(a, b, c);
  }
  if (a == 12345.f) out[0] = a;
}

int main(int argc, char** argv) {
  const double seconds = argc > 1 ? atof(argv[1]) : 20.0;
  const int neighbour = argc > 2 ? atoi(argv[2]) : 0;
  const int N = 4096 * 16 + 64;
  std::vector<float> hw(N), hx(N);
  unsigned s = 12345u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 32768.0f - 1.0f; };
  for (int i = 0; i < N; ++i) { hw[i] = rnd(); hx[i] = rnd(); }
  float *dw, *dx;
  Mis* dlog;
  int* dn;
  unsigned long long* dc;
  hipMalloc(&dw, N * 4); hipMalloc(&dx, N * 4); hipMalloc(&dlog, 256 * sizeof(Mis)); hipMalloc(&dn, 4); hipMalloc(&dc, 8);
  hipMemcpy(dw, hw.data(), N * 4, hipMemcpyHostToDevice); hipMemcpy(dx, hx.data(), N * 4, hipMemcpyHostToDevice);
  hipMemset(dn, 0, 4); hipMemset(dc, 0, 8);
  int* dcu;
  hipMalloc(&dcu, 1024 * 4); hipMemset(dcu, 0, 1024 * 4);
  hipStream_t s1, s2;
  hipStreamCreate(&s1); hipStreamCreate(&s2);
  float* nbo; float4 *nbs = nullptr, *nbd = nullptr;
  hipMalloc(&nbo, 64);
  const size_t n4 = (size_t)64 << 20;       // 1 GiB each way
  if (neighbour == 3) { hipMalloc(&nbs, n4 * 16); hipMalloc(&nbd, n4 * 16); hipMemset(nbs, 0, n4 * 16); }
  // neighbour 5: the library's conv forward on stream s2
  typedef int (*fn_create)(void**, int, void*);
  typedef int (*fn_ws)(void*, void*, size_t);
  typedef int (*fn_conv)(void*, const float*, const float*, const float*, float*, int, int, int, int, int, int, int, int, int, float);
  void* vctx = nullptr;
  fn_conv conv = nullptr;
  float *cx = nullptr, *cw = nullptr, *cy = nullptr;
  if (neighbour == 5) {
    void* lib = dlopen(getenv("VF_HIP_LIB") ? getenv("VF_HIP_LIB") : "video-filler_amd/lib/libvf_hip.so", RTLD_NOW);
    if (!lib) { printf("dlopen failed: %s\n", dlerror()); return 2; }
    fn_create create = (fn_create)dlsym(lib, "vf_ctx_create");
    fn_ws setws = (fn_ws)dlsym(lib, "vf_ctx_set_workspace");
    conv = (fn_conv)dlsym(lib, "vf_conv2d_fwd");
    void* ws;
    hipMalloc(&ws, (size_t)256 << 20);
    if (create(&vctx, 0, (void*)s2) || setws(vctx, ws, (size_t)256 << 20)) { printf("vf_ctx setup failed\n"); return 2; }
    // NB_MFMA_MODE: the library's GEMM arithmetic (0 fp32 MFMA, 1 one bf16 plane, 3 three bf16 planes = its default)
    if (getenv("NB_MFMA_MODE")) ((int (*)(void*, int))dlsym(lib, "vf_ctx_set_mfma_mode"))(vctx, atoi(getenv("NB_MFMA_MODE")));
    hipMalloc(&cx, (size_t)64 * 32 * 32 * 64 * 4); hipMalloc(&cw, (size_t)128 * 16 * 64 * 4); hipMalloc(&cy, (size_t)64 * 16 * 16 * 128 * 4);
    hipMemset(cx, 0, (size_t)64 * 32 * 32 * 64 * 4); hipMemset(cw, 0, (size_t)128 * 16 * 64 * 4);
  }
  // NB_CONV_B: batch of the library launch (64 = E3's own 512 blocks; 1 = 8 blocks, i.e. 8 of the 256 CUs run the GEMM)
  const int conv_b = getenv("NB_CONV_B") ? atoi(getenv("NB_CONV_B")) : 64;
  const auto t0 = std::chrono::steady_clock::now();
  int launches = 0;
  while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
    if (neighbour == 5)
      for (int q = 0; q < 2; ++q) conv(vctx, cx, cw, nullptr, cy, conv_b, 32, 32, 64, 128, 4, 2, 1, 0, 0.f);      // E3 forward, ~45 us each
    if (neighbour >= 300) {
      const int w = neighbour - 300;
      auto f = w == 0 ? k_nb_min<0> : w == 1 ? k_nb_min<1> : w == 2 ? k_nb_min<2> : w == 3 ? k_nb_min<3> : w == 4 ? k_nb_min<4> : w == 5 ? k_nb_min<5> : k_nb_min<6>;
      if (launches % 16 == 0) hipLaunchKernelGGL(f, dim3(512), dim3(256), 0, s2, nbo, 100000);
    } else
    if (neighbour >= 200) {
      const int w = neighbour - 200;
      auto f = w == 2 ? k_nb_shadow<2> : w == 3 ? k_nb_shadow<3> : w == 6 ? k_nb_shadow<6> : w == 7 ? k_nb_shadow<7> : w == 10 ? k_nb_shadow<10>
             : w == 11 ? k_nb_shadow<11> : w == 14 ? k_nb_shadow<14> : k_nb_shadow<15>;
      if (launches % 16 == 0) hipLaunchKernelGGL(f, dim3(512), dim3(256), 0, s2, nbo, 20000);
    } else
    if (neighbour >= 100) {         // NB_ITERS / NB_PER_LAUNCH: short kernels launched often (as the library's 45 us GEMMs are) or one long one
      static const int nb_iters = getenv("NB_ITERS") ? atoi(getenv("NB_ITERS")) : 20000, nb_per = getenv("NB_PER_LAUNCH") ? atoi(getenv("NB_PER_LAUNCH")) : 0;
      static const int nb_blocks = getenv("NB_BLOCKS") ? atoi(getenv("NB_BLOCKS")) : 512;
      if (nb_per) for (int q = 0; q < nb_per; ++q) hipLaunchKernelGGL(k_nb_combo, dim3(nb_blocks), dim3(256), 0, s2, nbo, nb_iters, neighbour - 100);
      else if (launches % 16 == 0) hipLaunchKernelGGL(k_nb_combo, dim3(nb_blocks), dim3(256), 0, s2, nbo, nb_iters, neighbour - 100);
    }
    if (launches % 16 == 0) {       // ~16 reproducer launches' worth of neighbour work, two blocks per CU
      if (neighbour == 1) hipLaunchKernelGGL(k_nb_mfma, dim3(512), dim3(256), 0, s2, nbo, 40000);
      if (neighbour == 2) hipLaunchKernelGGL(k_nb_lds, dim3(512), dim3(256), 0, s2, nbo, 40000);
      if (neighbour == 3) hipLaunchKernelGGL(k_nb_vmem, dim3(1024), dim3(256), 0, s2, nbs, nbd, n4, 4);
      if (neighbour == 4) hipLaunchKernelGGL(k_nb_valu, dim3(512), dim3(256), 0, s2, nbo, 100000);
      if (neighbour == 6) hipLaunchKernelGGL(k_nb_mfma_lds, dim3(512), dim3(256), 0, s2, nbo, 20000);
      if (neighbour == 7) hipLaunchKernelGGL(k_nb_trread, dim3(512), dim3(256), 0, s2, nbo, 40000);
      if (neighbour == 8) hipLaunchKernelGGL(k_nb_split, dim3(512), dim3(256), 0, s2, nbo, 60000);
      if (neighbour == 9) hipLaunchKernelGGL(k_nb_coexec, dim3(512), dim3(256), 0, s2, nbo, 40000);
      if (neighbour == 11) {
        static const int kb = getenv("NB_LDS_KB") ? atoi(getenv("NB_LDS_KB")) : 66, high = getenv("NB_LDS_HIGH") ? atoi(getenv("NB_LDS_HIGH")) : 0;
        auto f = kb == 48 ? k_nb_biglds<48, 0> : kb == 64 ? (high ? k_nb_biglds<64, 1> : k_nb_biglds<64, 0>)
               : kb == 66 ? (high ? k_nb_biglds<66, 1> : k_nb_biglds<66, 0>) : (high ? k_nb_biglds<128, 1> : k_nb_biglds<128, 0>);
        hipLaunchKernelGGL(f, dim3(256), dim3(256), 0, s2, nbo, 20000);
      }
      if (neighbour == 10) hipLaunchKernelGGL(k_nb_mfma_agpr, dim3(512), dim3(256), 0, s2, nbo, 40000);
    }
    hipLaunchKernelGGL(k_repro, dim3(512), dim3(256), 0, s1, dw, dx, 2000, dlog, dn, dc, dcu);      // short launches: waves come and go beside the neighbours'
    if (++launches % 16 == 0) hipDeviceSynchronize();
  }
  hipDeviceSynchronize();
  int n = 0;
  unsigned long long c = 0;
  std::vector<Mis> h(256);
  hipMemcpy(&n, dn, 4, hipMemcpyDeviceToHost); hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
  hipMemcpy(h.data(), dlog, 256 * sizeof(Mis), hipMemcpyDeviceToHost);
  printf("neighbour %d (0 none, 1 MFMA, 2 LDS, 3 global memory, 4 VALU, 5 the library's implicit GEMM, 6 MFMA from LDS + barrier, 7 ds_read_b64_tr_b16, 8 v_perm split, 9 MFMA + VALU, 10 MFMA into AGPRs, 11 a workgroup holding NB_LDS_KB of LDS, 100 + mask: GEMM-like combination, 200 + mask: work in the shadow of MFMAs, 300 + form: one MFMA + one LDS store per round) | ", neighbour);
  printf("variant %d: %d launches, %.3e packed results checked, %d mismatches\n", VARIANT, launches, (double)c, n);
  int q48 = 0, lo = 0;
  for (int i = 0; i < n && i < 256; ++i) {
    if (h[i].lane >= 48) ++q48;
    if (h[i].half == 0) ++lo;
    if (i < 12)
      printf("  block %d wave %d lane %d round %d chain %d half %s: want %.9g got %.9g\n", h[i].block, h[i].wave, h[i].lane, h[i].round,
             h[i].chain, h[i].half ? "hi" : "lo", h[i].want, h[i].got);
  }
  if (n) {
    std::vector<int> cu(1024);
    hipMemcpy(cu.data(), dcu, 1024 * 4, hipMemcpyDeviceToHost);
    int hit = 0;
    for (int i = 0; i < 1024; ++i) hit += cu[i] != 0;
    printf("  mismatching waves ran on %d distinct (xcc, se, cu) places:", hit);
    for (int i = 0, shown = 0; i < 1024 && shown < 16; ++i)
      if (cu[i]) { printf(" x%d.s%d.c%d=%d", i >> 7, (i >> 4) & 7, i & 15, cu[i]); ++shown; }
    printf("\n");
  }
  if (n) printf("  of the first %d logged: %d in lanes 48-63, %d in the low dword\n", n < 256 ? n : 256, q48, lo);
  return 0;
}
