// PROBE ONLY — never linked into the shipped library.  Variants of vf_smallm.hip's row-dot kernel, each differing from the
// round-3 pre-fix form (git bcfc5f7) in ONE respect, to decide which of the changes of commit 82a7d2b removed the run-to-run
// differences.  scripts/probe/build_rowdot_variants.sh compiles this file once per variant and links it in
// place of vf_smallm.o into video-filler_amd/lib/alt/libvf_hip_rd<variant>.so; scripts/probe/smallm_det.py runs the pass in
// place through VF_HIP_LIB=<that library>.
//
//   -DRED=0  __shfl_xor butterfly, 32 values x 6 steps (pre-fix)        -DLOADS=0  conditional stage loads (pre-fix)
//   -DRED=1  lanes meet in LDS in lane order (shipped)                   -DLOADS=1  unconditional loads + sched_barrier (shipped)
//   -DRED=2  butterfly, at most ONE value's ds_bpermute in flight (s_waitcnt lgkmcnt(0) after each value)
//   -DRED=3  DPP row reduction + v_readlane: no LDS-pipe instruction at all
//   -DRED=4  butterfly, but the result stored by 32 lanes (one value each) instead of 32 values from lane 0
//   -DRED=5  butterfly and lane-0 stores as pre-fix, but the block ALLOCATES 33 KB of LDS (written once, never needed)
//   -DCHECK=1 (with DIAG=1) a checker kernel follows every launch on the same stream: it recomputes every lane's partial sum from
//             A and W (same fmaf chain) and the butterfly of the stored partials (same tree), compares both bit for bit with what
//             the row-dot kernel left (g_dbg / slab) and logs the first mismatches; vf_probe_rowdot_report() prints them
//   -DDIAG=1 also stores every lane's partial sums before the reduction (vf_probe_rowdot_dbg returns the device buffer)
#include <algorithm>
#include <cstdlib>

#include "vf_common.h"

#ifndef RED
#define RED 0
#endif
#ifndef LOADS
#define LOADS 0
#endif
#ifndef DIAG
#define DIAG 0
#endif
#ifndef CHECK
#define CHECK 0
#endif

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

__device__ float g_dbg[128 * 32 * 64];      // [wave][value][lane], small probe shape only (128 waves)

template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}

template <int MT>
__global__ __launch_bounds__(256) void k_rowdot_var(const float* __restrict__ A, const float* __restrict__ W, float* __restrict__ slab,
                                                    int M, int N, int K, int steps_per_split, int ngroups, int dbg_on) {
  constexpr int R = 8;
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
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
#if LOADS == 0
  auto load = [&](int s, Stage& st) {
    const int64_t o = (int64_t)s * 256;
#pragma unroll
    for (int b = 0; b < MT; ++b) st.x[b] = b < M ? *(const f32x4*)(ap + (int64_t)b * K + o) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < R; ++r) st.w[r] = __builtin_nontemporal_load((const f32x4*)(wp + (int64_t)r * K + o));
  };
  auto fma = [&](Stage& st) {
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int b = 0; b < MT; ++b)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[r][b] = fmaf(st.w[r][e], st.x[b][e], acc[r][b]);
  };
  Stage s0, s1;
  load(0, s0);
  for (int s = 0; s < steps_per_split; s += 2) {
    if (s + 1 < steps_per_split) load(s + 1, s1);
    fma(s0);
    if (s + 2 < steps_per_split) load(s + 2, s0);
    if (s + 1 < steps_per_split) fma(s1);
  }
#else
  float xm[MT];
#pragma unroll
  for (int b = 0; b < MT; ++b) xm[b] = b < M ? 1.f : 0.f;
  auto load = [&](int s, Stage& st) {
    const int64_t o = (int64_t)s * 256;
#pragma unroll
    for (int b = 0; b < MT; ++b) st.x[b] = *(const f32x4*)(ap + (int64_t)(b < M ? b : 0) * K + o);
#pragma unroll
    for (int r = 0; r < R; ++r) st.w[r] = __builtin_nontemporal_load((const f32x4*)(wp + (int64_t)r * K + o));
  };
  auto fma = [&](Stage& st) {
#pragma unroll
    for (int b = 0; b < MT; ++b) st.x[b] = st.x[b] * xm[b];
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
  for (int s = 0; s < steps_per_split; s += 2) {
    load(s + 1, s1);
    __builtin_amdgcn_sched_barrier(0);
    fma(s0);
    __builtin_amdgcn_sched_barrier(0);
    load(min(s + 2, last), s0);
    __builtin_amdgcn_sched_barrier(0);
    fma(s1);
    __builtin_amdgcn_sched_barrier(0);
  }
#endif
#if DIAG
  if (dbg_on && wv < 128) {
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int b = 0; b < MT; ++b) g_dbg[(wv * 32 + r * MT + b) * 64 + lane] = acc[r][b];
  }
#endif
#if RED == 5
  __shared__ float dummy[4][R * MT][65];
  dummy[threadIdx.x >> 6][lane & 31][lane] = acc[0][0];
  if (steps_per_split < 0) {          // (never: keeps the allocation alive)
    __syncthreads();
    acc[0][0] += dummy[0][0][(lane + 1) & 63];
  }
#endif
#if RED == 1
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
#else
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int b = 0; b < MT; ++b) {
      float v = acc[r][b];
#if RED == 3
      v += dpp_mov<0xB1>(v);        // quad_perm [1,0,3,2]
      v += dpp_mov<0x4E>(v);        // quad_perm [2,3,0,1]
      v += dpp_mov<0x141>(v);       // row_half_mirror
      v += dpp_mov<0x140>(v);       // row_mirror: every lane holds its 16-lane row's sum
      const float q0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
      const float q1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
      const float q2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
      const float q3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
      v = (q0 + q1) + (q2 + q3);
#else
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
#if RED == 2
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
#endif
#endif
      acc[r][b] = v;
    }
#if RED == 4
  {
    float t = 0.f;
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int b = 0; b < MT; ++b) t = lane == r * MT + b ? acc[r][b] : t;
    const int r = lane / MT, bb = lane - r * MT;
    if (lane < R * MT && bb < M) slab[(int64_t)ks * M * N + (int64_t)bb * N + n0 + r] = t;
  }
  if (false) {
#else
  if (lane == 0) {
#endif
    float* o = slab + (int64_t)ks * M * N + n0;
#pragma unroll
    for (int b = 0; b < MT; ++b)
      if (b < M) {
#pragma unroll
        for (int r = 0; r < R; ++r) o[(int64_t)b * N + r] = acc[r][b];
      }
  }
#endif
}

}  // namespace

#if CHECK
namespace {
struct Mis { int kind, launch, wave, value, lane; float want, got; float lanes[64]; float w[8], x[8]; };
__device__ Mis g_mis[32];
__device__ int g_nmis, g_nchecked;
// one block per wave of the row-dot launch, 64 threads = its lanes
__global__ __launch_bounds__(64) void k_rowdot_check(const float* __restrict__ A, const float* __restrict__ W, const float* __restrict__ slab,
                                                      int M, int N, int K, int steps_per_split, int ngroups, int launch) {
  const int wv = blockIdx.x, lane = threadIdx.x;
  const int ks = wv / ngroups, grp = wv - ks * ngroups, n0 = grp * 8;
  const int64_t k0 = (int64_t)ks * steps_per_split * 256 + 4 * lane;
  __shared__ float t[64];
  for (int r = 0; r < 8; ++r)
    for (int b = 0; b < 4; ++b) {
      const int v = r * 4 + b;
      float acc = 0.f;
      if (b < M)
        for (int s = 0; s < steps_per_split; ++s)
          for (int e = 0; e < 4; ++e)
            acc = fmaf(W[(int64_t)(n0 + r) * K + k0 + (int64_t)s * 256 + e], A[(int64_t)b * K + k0 + (int64_t)s * 256 + e], acc);
      const float stored = g_dbg[(wv * 32 + v) * 64 + lane];
      if (__float_as_uint(acc) != __float_as_uint(stored)) {
        const int i = atomicAdd(&g_nmis, 1);
        if (i < 32) {
          g_mis[i].kind = 1; g_mis[i].launch = launch; g_mis[i].wave = wv; g_mis[i].value = v; g_mis[i].lane = lane; g_mis[i].want = acc; g_mis[i].got = stored;
          for (int s = 0; s < 2 && s < steps_per_split; ++s)
            for (int e = 0; e < 4; ++e) {
              g_mis[i].w[s * 4 + e] = W[(int64_t)(n0 + r) * K + k0 + (int64_t)s * 256 + e];
              g_mis[i].x[s * 4 + e] = A[(int64_t)b * K + k0 + (int64_t)s * 256 + e];
            }
        }
      }
      __syncthreads();
      t[lane] = stored;
      __syncthreads();
      for (int off = 32; off > 0; off >>= 1) {
        const float s2 = t[lane] + t[lane ^ off];
        __syncthreads();
        t[lane] = s2;
        __syncthreads();
      }
      if (lane == 0 && b < M) {
        const float got = slab[(int64_t)ks * M * N + (int64_t)b * N + n0 + r];
        if (__float_as_uint(got) != __float_as_uint(t[0])) {
          const int i = atomicAdd(&g_nmis, 1);
          if (i < 32) {
            g_mis[i].kind = 2; g_mis[i].launch = launch; g_mis[i].wave = wv; g_mis[i].value = v; g_mis[i].lane = 0; g_mis[i].want = t[0]; g_mis[i].got = got;
            for (int l = 0; l < 64; ++l) g_mis[i].lanes[l] = g_dbg[(wv * 32 + v) * 64 + l];
          }
        }
      }
    }
  if (wv == 0 && lane == 0) atomicAdd(&g_nchecked, 1);
}
}  // namespace
extern "C" __attribute__((visibility("default"))) int vf_probe_rowdot_report() {
  static Mis h[32];
  int n = 0, c = 0;
  hipDeviceSynchronize();
  hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_nmis), sizeof(int));
  hipMemcpyFromSymbol(&c, HIP_SYMBOL(g_nchecked), sizeof(int));
  hipMemcpyFromSymbol(h, HIP_SYMBOL(g_mis), sizeof(h));
  fprintf(stderr, "rowdot check: %d launches checked, %d mismatches\n", c, n);
  for (int i = 0; i < n && i < 32; ++i) {
    fprintf(stderr, "  %s launch %d wave %d value %d (r %d b %d) lane %d: want %.9g got %.9g", h[i].kind == 1 ? "PARTIAL" : "SLAB", h[i].launch, h[i].wave,
            h[i].value, h[i].value / 4, h[i].value % 4, h[i].lane, h[i].want, h[i].got);
    if (h[i].kind == 1) {
      fprintf(stderr, " | w:");
      for (int l = 0; l < 8; ++l) fprintf(stderr, " %.9g", h[i].w[l]);
      fprintf(stderr, " | x:");
      for (int l = 0; l < 8; ++l) fprintf(stderr, " %.9g", h[i].x[l]);
    }
    if (h[i].kind == 2) {
      fprintf(stderr, " | lanes:");
      for (int l = 0; l < 64; ++l) fprintf(stderr, " %.6g", h[i].lanes[l]);
    }
    fprintf(stderr, "\n");
  }
  return n;
}
#endif
extern "C" __attribute__((visibility("default"))) void* vf_probe_rowdot_dbg() {
  void* p = nullptr;
  if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_dbg)) != hipSuccess) return nullptr;
  return p;
}
extern "C" __attribute__((visibility("default"))) int vf_probe_rowdot_variant() { return RED * 100 + LOADS * 10 + DIAG; }

int vf_internal_smallm_plan(int form, int M, int N, int K, size_t ws_bytes) {
  if (form != 0 || M < 1 || M > 4) return 0;                     // (row-dot, M <= 4 only; everything else keeps the tiled kernel)
  static const int want_waves = getenv("VF_SMALLM_WAVES") ? atoi(getenv("VF_SMALLM_WAVES")) : 4096;
  if (K % 256 != 0 || N % 32 != 0) return 0;
  const int steps = K / 256, groups = N / 8;
  int ksplit = 0;
  for (int s = 2; s <= steps; ++s)
    if (steps % s == 0 && (steps / s) % 2 == 0) {
      ksplit = s;
      if ((int64_t)groups * s >= want_waves) break;
    }
  if (ksplit < 2 || (size_t)ksplit * M * N * sizeof(float) > ws_bytes) return 0;
  return ksplit;
}

int vf_internal_smallm_launch(vf_ctx* ctx, int form, const float* A, const float* W, float* slab, int M, int N, int K, int ksplit) {
  const int groups = N / 8, steps = K / 256 / ksplit;
  const dim3 grid((unsigned)((int64_t)groups * ksplit / 4));
  const int dbg_on = (int64_t)groups * ksplit <= 128;
  hipLaunchKernelGGL(k_rowdot_var<4>, grid, dim3(256), 0, ctx->stream, A, W, slab, M, N, K, steps, groups, dbg_on);
  VF_LAUNCH_CHECK();
#if CHECK
  static int launch = 0;
  if (dbg_on) hipLaunchKernelGGL(k_rowdot_check, dim3(groups * ksplit), dim3(64), 0, ctx->stream, A, W, (const float*)slab, M, N, K, steps, groups, launch++);
  VF_LAUNCH_CHECK();
#endif
  return 0;
}
