// vf_core.hip — context, memory, layout conversion, pointwise modules, criteria and the fused Adam step.
// Everything here is HBM-bound byte work: 16-byte accesses, grid-stride loops capped at ~2048 blocks.
#include <algorithm>
#include <cmath>

#include "vf_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------------ errors / ctx
static thread_local char g_err[1024] = "";

void vf_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

VF_API const char* vf_last_error(void) { return g_err; }
VF_API int vf_version(void) { return 100; }

VF_API int vf_ctx_create(vf_ctx** out, int device, void* stream) {
  VF_REQUIRE(out != nullptr, "vf_ctx_create: out is NULL");
  int ndev = 0;
  VF_CHECK_HIP(hipGetDeviceCount(&ndev));
  VF_REQUIRE(ndev > 0, "no HIP device visible: the gfx950 backend has no CPU fallback");
  VF_REQUIRE(device >= 0 && device < ndev, "device %d out of range (have %d)", device, ndev);
  VF_CHECK_HIP(hipSetDevice(device));
  vf_ctx* c = new vf_ctx();
  c->device = device;
  c->stream = (hipStream_t)stream;
  c->ws = nullptr;
  c->ws_bytes = 0;
  c->ws_front = 0;
  c->wg_active = 0;
  c->wg_rec = nullptr;
  memset(&c->bnf, 0, sizeof(c->bnf));
  c->bnf_yact = nullptr;
  c->bnf_act = 0;
  c->bnf_slope = 0.f;
  c->bnf_groups = 1;
  c->bnf_rows_cap = 0;
  c->bnf_result_rows = 0;
  c->mfma_bf16 = 3;      // fp32 operands as three exact bf16 planes (see vf_ctx_set_mfma_mode)
  c->act_bits_out = nullptr;
  c->act_bits_written = 0;
  c->dmask_bits = nullptr;
  c->dot_act_y = nullptr;
  c->dot_act = 0;
  c->dot_act_slope = 0.f;
  *out = c;
  return 0;
}
VF_API int vf_ctx_destroy(vf_ctx* ctx) {
  if (ctx) vf_internal_wg_free(ctx);
  delete ctx;
  return 0;
}
VF_API int vf_ctx_set_mfma_mode(vf_ctx* ctx, int mode) {
  VF_REQUIRE(mode == 0 || mode == 1 || mode == 3,
             "vf_ctx_set_mfma_mode: 0 = fp32 operands, 1 = bf16 operands, 3 = fp32 operands as three bf16 planes (got %d)", mode);
  ctx->mfma_bf16 = mode;
  return 0;
}
VF_API int vf_ctx_set_stream(vf_ctx* ctx, void* stream) {
  ctx->stream = (hipStream_t)stream;
  return 0;
}
VF_API int vf_ctx_set_workspace(vf_ctx* ctx, void* ptr, size_t bytes) {
  ctx->ws = ptr;
  ctx->ws_bytes = bytes;
  return 0;
}
VF_API size_t vf_workspace_bytes_hint(void) { return (size_t)1 << 30; }   // room for every layer's split-K slabs of a grouped launch
VF_API int vf_stream_synchronize(vf_ctx* ctx) {
  VF_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  return 0;
}
VF_API int vf_malloc(void** out, size_t bytes) {
  VF_CHECK_HIP(hipMalloc(out, bytes));
  return 0;
}
VF_API int vf_free(void* ptr) {
  VF_CHECK_HIP(hipFree(ptr));
  return 0;
}
VF_API int vf_memcpy_h2d(vf_ctx* ctx, void* dst, const void* src, size_t bytes) {
  VF_CHECK_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
  return 0;
}
VF_API int vf_memcpy_d2h(vf_ctx* ctx, void* dst, const void* src, size_t bytes) {
  VF_CHECK_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
  VF_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  return 0;
}
VF_API int vf_zero(vf_ctx* ctx, void* ptr, size_t bytes) {
  VF_CHECK_HIP(hipMemsetAsync(ptr, 0, bytes, ctx->stream));
  return 0;
}

// ------------------------------------------------------------------------------------------------ profiling
#include <map>
#include <string>
#include <vector>
struct ProfRec {
  std::string name;
  double flops, bytes;
  hipEvent_t e0, e1;
};
struct ProfAgg {
  std::string name;
  int64_t launches;
  double ms, flops, bytes;
};
static bool g_prof_on = false;
static std::vector<ProfRec> g_prof_recs;
static std::vector<ProfAgg> g_prof_agg;
bool vf_prof_enabled() { return g_prof_on; }
void vf_prof_push(vf_ctx* ctx, const char* name, double flops, double bytes, bool begin) {
  if (begin) {
    ProfRec r;
    r.name = name;
    r.flops = flops;
    r.bytes = bytes;
    hipEventCreate(&r.e0);
    hipEventCreate(&r.e1);
    hipEventRecord(r.e0, ctx->stream);
    g_prof_recs.push_back(r);
  } else if (!g_prof_recs.empty()) {
    hipEventRecord(g_prof_recs.back().e1, ctx->stream);
  }
}
bool vf_prof_ext(const char* name, double flops, double bytes, hipEvent_t* e0, hipEvent_t* e1) {
  if (!g_prof_on) return false;
  ProfRec r;
  r.name = name;
  r.flops = flops;
  r.bytes = bytes;
  hipEventCreate(&r.e0);
  hipEventCreate(&r.e1);
  *e0 = r.e0;
  *e1 = r.e1;
  g_prof_recs.push_back(r);
  return true;
}
VF_API int vf_prof_begin(vf_ctx* ctx) {
  VF_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  g_prof_recs.clear();
  g_prof_agg.clear();
  g_prof_on = true;
  return 0;
}
VF_API int vf_prof_end(vf_ctx* ctx) {
  g_prof_on = false;
  VF_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  std::map<std::string, size_t> idx;
  for (auto& r : g_prof_recs) {
    float ms = 0.f;
    hipEventElapsedTime(&ms, r.e0, r.e1);
    auto it = idx.find(r.name);
    if (it == idx.end()) {
      idx[r.name] = g_prof_agg.size();
      g_prof_agg.push_back({r.name, 0, 0.0, 0.0, 0.0});
      it = idx.find(r.name);
    }
    ProfAgg& a = g_prof_agg[it->second];
    a.launches += 1;
    a.ms += ms;
    a.flops += r.flops;
    a.bytes += r.bytes;
    hipEventDestroy(r.e0);
    hipEventDestroy(r.e1);
  }
  g_prof_recs.clear();
  return 0;
}
VF_API int vf_prof_count(void) { return (int)g_prof_agg.size(); }
VF_API int vf_prof_get(int i, char* name, int name_cap, int64_t* launches, double* ms, double* flops, double* bytes) {
  VF_REQUIRE(i >= 0 && i < (int)g_prof_agg.size(), "vf_prof_get: index %d out of range", i);
  const ProfAgg& a = g_prof_agg[i];
  if (name && name_cap > 0) {
    strncpy(name, a.name.c_str(), name_cap - 1);
    name[name_cap - 1] = 0;
  }
  if (launches) *launches = a.launches;
  if (ms) *ms = a.ms;
  if (flops) *flops = a.flops;
  if (bytes) *bytes = a.bytes;
  return 0;
}

static inline int grid_for(int64_t n, int per_thread = 4) {
  return (int)std::max<int64_t>(1, std::min<int64_t>(vf_cdiv(n, 256 * (int64_t)per_thread), 2048));
}

__global__ void k_zero_segments(float* __restrict__ base, const int64_t* __restrict__ offs, const int64_t* __restrict__ lens) {
  const int64_t off = offs[blockIdx.x], len = lens[blockIdx.x];
  for (int64_t i = threadIdx.x; i < len; i += blockDim.x) base[off + i] = 0.f;
}
VF_API int vf_zero_segments(vf_ctx* ctx, float* base, const int64_t* offs, const int64_t* lens, int nseg) {
  if (nseg <= 0) return 0;
  hipLaunchKernelGGL(k_zero_segments, dim3(nseg), dim3(256), 0, ctx->stream, base, offs, lens);
  VF_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------ layout
// per image: src [R][S] -> dst [S][R]   (NCHW->NHWC: R = C, S = H*W ; NHWC->NCHW: R = H*W, S = C)
__global__ void k_transpose(const float* __restrict__ src, float* __restrict__ dst, int R, int S) {
  __shared__ float tile[32][33];
  const int64_t img = (int64_t)blockIdx.z * R * S;
  const int s0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  for (int j = threadIdx.y; j < 32; j += 8) {
    const int r = r0 + j, s = s0 + threadIdx.x;
    if (r < R && s < S) tile[j][threadIdx.x] = src[img + (int64_t)r * S + s];
  }
  __syncthreads();
  for (int j = threadIdx.y; j < 32; j += 8) {
    const int s = s0 + j, r = r0 + threadIdx.x;
    if (r < R && s < S) dst[img + (int64_t)s * R + r] = tile[threadIdx.x][j];
  }
}
VF_API int vf_nchw_to_nhwc(vf_ctx* ctx, const float* src, float* dst, int B, int C, int H, int W) {
  const int S = H * W;
  hipLaunchKernelGGL(k_transpose, dim3((int)vf_cdiv(S, 32), (int)vf_cdiv(C, 32), B), dim3(32, 8), 0, ctx->stream, src, dst, C, S);
  VF_LAUNCH_CHECK();
  return 0;
}
VF_API int vf_nhwc_to_nchw(vf_ctx* ctx, const float* src, float* dst, int B, int C, int H, int W) {
  const int S = H * W;
  hipLaunchKernelGGL(k_transpose, dim3((int)vf_cdiv(C, 32), (int)vf_cdiv(S, 32), B), dim3(32, 8), 0, ctx->stream, src, dst, S, C);
  VF_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------ pointwise
enum { OP_ACT_FWD, OP_ACT_BWD, OP_AXPBY, OP_CMUL, OP_SCALE_SHIFT, OP_COMPOSE, OP_MSE_BWD };

template <int OP>
__global__ void k_pointwise(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ c,
                            float* __restrict__ out, int64_t n, float f0, float f1, int act, int vec) {
  const int64_t n4 = vec ? n >> 2 : 0;      // vec = 0: some operand is not 16-byte aligned (a sub-batch view of a tiny tensor)
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  auto op = [&](float av, float bv, float cv, float ov) -> float {
    if constexpr (OP == OP_ACT_FWD) return vf_act_apply(av, act, f0);
    if constexpr (OP == OP_ACT_BWD) return vf_act_grad(av, bv, act, f0);        // a = y, b = gy
    if constexpr (OP == OP_AXPBY) return f0 * av + f1 * ov;                       // out = f0*a + f1*out
    if constexpr (OP == OP_CMUL) return ov * av;
    if constexpr (OP == OP_SCALE_SHIFT) return ov * f0 + f1;
    if constexpr (OP == OP_COMPOSE) return cv != 0.f ? bv : av;                   // a = real, b = fake, c = mask
    if constexpr (OP == OP_MSE_BWD) return f0 * (av - bv);
    return 0.f;
  };
  constexpr bool RA = OP != OP_SCALE_SHIFT;
  constexpr bool RB = OP == OP_ACT_BWD || OP == OP_COMPOSE || OP == OP_MSE_BWD;
  constexpr bool RC = OP == OP_COMPOSE;
  constexpr bool RO = OP == OP_AXPBY || OP == OP_CMUL || OP == OP_SCALE_SHIFT;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += stride) {
    f32x4 av = {0, 0, 0, 0}, bv = av, cv = av, ov = av;
    if (RA) av = ((const f32x4*)a)[i];
    if (RB) bv = ((const f32x4*)b)[i];
    if (RC) cv = ((const f32x4*)c)[i];
    if (RO) ov = ((const f32x4*)out)[i];
    f32x4 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) r[e] = op(av[e], bv[e], cv[e], ov[e]);
    ((f32x4*)out)[i] = r;
  }
  // tail
  for (int64_t i = (n4 << 2) + blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += stride)
    out[i] = op(RA ? a[i] : 0.f, RB ? b[i] : 0.f, RC ? c[i] : 0.f, RO ? out[i] : 0.f);
}

template <int OP>
static int launch_pw(vf_ctx* ctx, const float* a, const float* b, const float* c, float* out, int64_t n, float f0, float f1,
                     int act) {
  if (n <= 0) return 0;
  const uintptr_t bits = (uintptr_t)a | (uintptr_t)b | (uintptr_t)c | (uintptr_t)out;
  VF_REQUIRE((bits & 3) == 0, "pointwise operands must be float aligned");
  const int vec = (bits & 15) == 0;
  VF_REQUIRE(vec || n <= (1 << 16), "pointwise operands of this size must be 16-byte aligned");
  static const char* names[] = {"pw_act_fwd", "pw_act_bwd", "pw_axpby", "pw_cmul", "pw_scale_shift", "pw_compose", "pw_mse_bwd"};
  const int nops = (a != nullptr) + (b != nullptr) + (c != nullptr) + 1 + ((OP == OP_AXPBY || OP == OP_CMUL || OP == OP_SCALE_SHIFT) ? 1 : 0);
  VfProf prof(ctx, names[OP], 0.0, 4.0 * (double)n * nops);
  hipLaunchKernelGGL((k_pointwise<OP>), dim3(grid_for(n)), dim3(256), 0, ctx->stream, a, b, c, out, n, f0, f1, act, vec);
  VF_LAUNCH_CHECK();
  return 0;
}

VF_API int vf_act_fwd(vf_ctx* ctx, const float* x, float* y, int64_t n, int act, float slope) {
  return launch_pw<OP_ACT_FWD>(ctx, x, nullptr, nullptr, y, n, slope, 0.f, act);
}
VF_API int vf_act_bwd(vf_ctx* ctx, const float* y, const float* gy, float* gx, int64_t n, int act, float slope) {
  return launch_pw<OP_ACT_BWD>(ctx, y, gy, nullptr, gx, n, slope, 0.f, act);
}
VF_API int vf_axpby(vf_ctx* ctx, float a, const float* x, float b, float* y, int64_t n) {
  return launch_pw<OP_AXPBY>(ctx, x, nullptr, nullptr, y, n, a, b, 0);
}
VF_API int vf_cmul(vf_ctx* ctx, const float* x, float* y, int64_t n) {
  return launch_pw<OP_CMUL>(ctx, x, nullptr, nullptr, y, n, 0.f, 0.f, 0);
}
VF_API int vf_scale_shift(vf_ctx* ctx, float* y, float a, float b, int64_t n) {
  return launch_pw<OP_SCALE_SHIFT>(ctx, nullptr, nullptr, nullptr, y, n, a, b, 0);
}
VF_API int vf_masked_compose(vf_ctx* ctx, float* out, const float* real, const float* fake, const float* mask, int64_t n) {
  return launch_pw<OP_COMPOSE>(ctx, real, fake, mask, out, n, 0.f, 0.f, 0);
}
VF_API int vf_mse_bwd(vf_ctx* ctx, const float* x, const float* t, float* gx, int64_t n) {
  return launch_pw<OP_MSE_BWD>(ctx, x, t, nullptr, gx, n, 2.f / (float)n, 0.f, 0);
}

// ------------------------------------------------------------------------------------------------ criteria
__device__ __forceinline__ void block_add_double(double v, double* dst) {
  __shared__ double red[256];
  red[threadIdx.x] = v;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) atomicAdd(dst, red[0]);
}

// nn.BCECriterion: single block (n = batch size)
__global__ __launch_bounds__(256) void k_bce_fwd(const float* __restrict__ x, float label, int n, double* __restrict__ loss) {
  const double EPS = 1e-12;
  double s = 0;
  for (int i = threadIdx.x; i < n; i += 256) {
    const double xv = x[i], t = label;
    s -= log(xv + EPS) * t + log(1. - xv + EPS) * (1. - t);
  }
  __shared__ double red[256];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) *loss = red[0] / (double)n;
}
__global__ void k_bce_bwd(const float* __restrict__ x, float label, float* __restrict__ gx, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double EPS = 1e-12, xv = x[i], t = label;
  gx[i] = (float)(-(1.0 / (double)n) * (t - xv) / ((1. - xv + EPS) * (xv + EPS)));
}
// forward value AND gradient of one or two groups of n scores each in ONE launch (block g = group g: the closures always ask for
// both, back to back — train.lua:331-349 for netD's real and fake halves, :364-366 for the generator's cost); the arithmetic of
// k_bce_fwd / k_bce_bwd, element for element
__global__ __launch_bounds__(256) void k_bce_fwd_bwd(const float* __restrict__ x, float label0, float label1, int n,
                                                      double* __restrict__ loss0, double* __restrict__ loss1, float* __restrict__ gx) {
  const double EPS = 1e-12;
  const int g = blockIdx.x;
  const double t = g ? label1 : label0;
  x += (int64_t)g * n;
  gx += (int64_t)g * n;
  double s = 0;
  for (int i = threadIdx.x; i < n; i += 256) {
    const double xv = x[i];
    s -= log(xv + EPS) * t + log(1. - xv + EPS) * (1. - t);
    gx[i] = (float)(-(1.0 / (double)n) * (t - xv) / ((1. - xv + EPS) * (xv + EPS)));
  }
  __shared__ double red[256];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) *(g ? loss1 : loss0) = red[0] / (double)n;
}
VF_API int vf_bce_fwd_bwd(vf_ctx* ctx, const float* x, float label0, float label1, int n_per_group, int groups, double* loss0,
                          double* loss1, float* gx) {
  VF_REQUIRE(groups == 1 || groups == 2, "vf_bce_fwd_bwd: groups must be 1 or 2 (got %d)", groups);
  VF_REQUIRE(x && gx && loss0 && (groups == 1 || loss1) && n_per_group > 0, "vf_bce_fwd_bwd: bad arguments");
  hipLaunchKernelGGL(k_bce_fwd_bwd, dim3(groups), dim3(256), 0, ctx->stream, x, label0, label1, n_per_group, loss0, loss1, gx);
  VF_LAUNCH_CHECK();
  return 0;
}
VF_API int vf_bce_fwd(vf_ctx* ctx, const float* x, float label, int n, double* loss) {
  hipLaunchKernelGGL(k_bce_fwd, dim3(1), dim3(256), 0, ctx->stream, x, label, n, loss);
  VF_LAUNCH_CHECK();
  return 0;
}
VF_API int vf_bce_bwd(vf_ctx* ctx, const float* x, float label, float* gx, int n) {
  hipLaunchKernelGGL(k_bce_bwd, dim3((int)vf_cdiv(n, 256)), dim3(256), 0, ctx->stream, x, label, gx, n);
  VF_LAUNCH_CHECK();
  return 0;
}

// sum((x-t)^2)/n -> loss (double atomics across blocks; order only perturbs the 16th digit)
__global__ __launch_bounds__(256) void k_mse_fwd(const float* __restrict__ x, const float* __restrict__ t, int64_t n,
                                                 double inv_n, double* __restrict__ loss) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x, n4 = n >> 2;
  float s = 0.f;
  double sd = 0;
  int cnt = 0;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += stride) {
    const f32x4 d = ((const f32x4*)x)[i] - ((const f32x4*)t)[i];
    s += d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3];
    if (++cnt == 16) {
      sd += (double)s;
      s = 0.f;
      cnt = 0;
    }
  }
  for (int64_t i = (n4 << 2) + blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += stride) {
    const float d = x[i] - t[i];
    s += d * d;
  }
  sd += (double)s;
  block_add_double(sd * inv_n, loss);
}
VF_API int vf_mse_fwd(vf_ctx* ctx, const float* x, const float* t, int64_t n, double* loss) {
  VF_CHECK_HIP(hipMemsetAsync(loss, 0, sizeof(double), ctx->stream));
  VfProf prof(ctx, "mse_fwd", 0.0, 8.0 * (double)n);
  hipLaunchKernelGGL(k_mse_fwd, dim3(grid_for(n, 16)), dim3(256), 0, ctx->stream, x, t, n, 1.0 / (double)n, loss);
  VF_LAUNCH_CHECK();
  return 0;
}

// fused: loss = mean((x-t)^2) ; df_dg = alpha*df_dg + (2/n)(x-t)*wgt
__global__ __launch_bounds__(256) void k_recon_grad_mix(float* __restrict__ dfdg, const float* __restrict__ x,
                                                        const float* __restrict__ t, const float* __restrict__ mask,
                                                        float alpha, float c0, float c1, int band, int HW, int C, int64_t n,
                                                        float two_over_n, double inv_n, double* __restrict__ loss) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  double sd = 0;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += stride) {
    const float d = x[i] - t[i];
    sd += (double)(d * d);
    float w;
    if (mask) {
      w = c0 + c1 * mask[i];
    } else if (band > 0) {
      const int64_t pix = i / C;
      const int wcol = (int)(pix % HW), hrow = (int)((pix / HW) % HW);
      const bool inside = hrow >= band && hrow < HW - band && wcol >= band && wcol < HW - band;
      w = inside ? c0 : c0 + c1;
    } else {
      w = c0;
    }
    dfdg[i] = alpha * dfdg[i] + (two_over_n * d) * w;
  }
  block_add_double(sd * inv_n, loss);
}
// the same pass in 16-byte pieces, four of them per thread with every load issued before the first use (the scalar form is a chain of
// 4-byte loads, eight per thread one after the other: 12.5 us for 786 432 elements that move 13 MB)
typedef float vf_f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_recon_grad_mix4(float* __restrict__ dfdg, const float* __restrict__ x,
                                                         const float* __restrict__ t, const float* __restrict__ mask,
                                                         float alpha, float c0, float c1, int band, int HW, int lgHW, int C, int64_t n4,
                                                         float two_over_n, double inv_n, double* __restrict__ loss) {
  constexpr int NV = 4;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t i0 = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  double sd = 0;
  for (int64_t base = i0; base < n4; base += NV * stride) {
    vf_f32x4 xv[NV], tv[NV], gv[NV], mv[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int64_t i = base + j * stride;
      if (i < n4) {
        xv[j] = ((const vf_f32x4*)x)[i];
        tv[j] = ((const vf_f32x4*)t)[i];
        gv[j] = ((const vf_f32x4*)dfdg)[i];
        if (mask) mv[j] = ((const vf_f32x4*)mask)[i];
      }
    }
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int64_t i = base + j * stride;
      if (i >= n4) continue;
      vf_f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float d = xv[j][e] - tv[j][e];
        sd += (double)(d * d);
        float w;
        if (mask) {
          w = c0 + c1 * mv[j][e];
        } else if (band > 0) {
          // (32-bit arithmetic: the host takes this kernel for n < 2^31 only; 64-bit divisions by C and HW for every element
          //  made the pass ALU-bound)
          const unsigned pix = ((unsigned)(4 * i) + (unsigned)e) / (unsigned)C;
          const unsigned wcol = lgHW >= 0 ? (pix & (unsigned)(HW - 1)) : pix % (unsigned)HW;
          const unsigned hrow = lgHW >= 0 ? ((pix >> lgHW) & (unsigned)(HW - 1)) : (pix / (unsigned)HW) % (unsigned)HW;
          const bool inside = (int)hrow >= band && (int)hrow < HW - band && (int)wcol >= band && (int)wcol < HW - band;
          w = inside ? c0 : c0 + c1;
        } else {
          w = c0;
        }
        o[e] = alpha * gv[j][e] + (two_over_n * d) * w;
      }
      ((vf_f32x4*)dfdg)[i] = o;
    }
  }
  block_add_double(sd * inv_n, loss);
}
VF_API int vf_recon_grad_mix(vf_ctx* ctx, float* df_dg, const float* x, const float* t, const float* mask, float alpha,
                             float c0, float c1, int band, int HW, int C, int64_t n, double* loss) {
  VF_CHECK_HIP(hipMemsetAsync(loss, 0, sizeof(double), ctx->stream));
  VfProf prof(ctx, "recon_grad_mix", 0.0, 4.0 * (double)n * (mask ? 5 : 4));
  if (n % 4 == 0 && n < ((int64_t)1 << 31) && ((((uintptr_t)df_dg) | ((uintptr_t)x) | ((uintptr_t)t) | ((uintptr_t)mask)) & 15) == 0) {
    int lgHW = -1;
    if (HW > 0 && (HW & (HW - 1)) == 0)
      for (lgHW = 0; (1 << lgHW) < HW; ++lgHW) {}
    hipLaunchKernelGGL(k_recon_grad_mix4, dim3(grid_for(n / 4, 4)), dim3(256), 0, ctx->stream, df_dg, x, t, mask, alpha, c0, c1, band,
                       HW, lgHW, C, n / 4, 2.f / (float)n, 1.0 / (double)n, loss);
    VF_LAUNCH_CHECK();
    return 0;
  }
  hipLaunchKernelGGL(k_recon_grad_mix, dim3(grid_for(n, 8)), dim3(256), 0, ctx->stream, df_dg, x, t, mask, alpha, c0, c1, band,
                     HW, C, n, 2.f / (float)n, 1.0 / (double)n, loss);
  VF_LAUNCH_CHECK();
  return 0;
}

// nn.GDLCriterion(1) forward on NHWC, with the reference's flattened pairing:
// per (b,c) plane, element k of the H x (W-1) crops pairs with element k of the (H-1) x W crops.
__global__ __launch_bounds__(256) void k_gdl_fwd(const float* __restrict__ yh, const float* __restrict__ y, int B, int H, int W,
                                                 int C, double inv_cnt, double* __restrict__ loss) {
  const int64_t m = (int64_t)(H - 1) * W;
  const int64_t total = (int64_t)B * m * C;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  double sd = 0;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += stride) {
    const int c = (int)(e % C);
    const int64_t k = (e / C) % m;
    const int64_t b = e / (C * m);
    const int64_t base = b * H * W;
    const int64_t r = k / (W - 1), cc = k % (W - 1);
    const int64_t i2 = (base + r * W + cc) * C + c;       // X[.., :, 0:W-1] flattened
    const int64_t j2 = i2 + C;                             // X[.., :, 1:W]
    const int64_t i1 = (base + k) * C + c;                 // X[.., 0:H-1, :] flattened
    const int64_t j1 = (base + W + k) * C + c;             // X[.., 1:H, :]
    const float t1 = fabsf(y[i2] - y[i1]), t2 = fabsf(yh[i2] - yh[i1]);
    const float t3 = fabsf(y[j2] - y[j1]), t4 = fabsf(yh[j2] - yh[j1]);
    sd += (double)fabsf(t1 - t2) + (double)fabsf(t3 - t4);
  }
  block_add_double(sd * inv_cnt, loss);
}
VF_API int vf_gdl_fwd(vf_ctx* ctx, const float* yhat, const float* y, int B, int H, int W, int C, double* loss) {
  VF_REQUIRE(H == W, "GDLCriterion needs square maps (the reference's CSubTable pairs H x (W-1) with (H-1) x W)");
  VF_CHECK_HIP(hipMemsetAsync(loss, 0, sizeof(double), ctx->stream));
  const int64_t cnt = (int64_t)B * C * (H - 1) * W;
  hipLaunchKernelGGL(k_gdl_fwd, dim3(grid_for(cnt, 8)), dim3(256), 0, ctx->stream, yhat, y, B, H, W, C, 1.0 / (double)cnt, loss);
  VF_LAUNCH_CHECK();
  return 0;
}

// nn.GDLCriterion(1) updateGradInput (gdl_criterion.lua:47-53), NHWC, gather form: every element of gYhat collects the (up to)
// four pairings it takes part in — as an element of the i2 crop (cols 0..W-2), of i1 (rows 0..H-2), of j2 (cols 1..W-1) and of
// j1 (rows 1..H-1) — so nothing is accumulated across threads.  pair(k) = the gradient handed to d = X_2[k] - X_1[k]:
//   -(sign_ge0(|Y_2[k] - Y_1[k]| - |d|) / count) * sign_ge0(d)      (THNN AbsCriterion / Abs: derivative +1 at 0)
__global__ __launch_bounds__(256) void k_gdl_bwd(const float* __restrict__ yh, const float* __restrict__ y, float* __restrict__ g,
                                                 int B, int H, int W, int C, float norm) {
  const int64_t m = (int64_t)(H - 1) * W, hw = (int64_t)H * W;
  const int64_t total = (int64_t)B * hw * C;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += stride) {
    const int c = (int)(e % C);
    const int64_t q = (e / C) % hw, b = e / (C * hw);
    const int64_t base = b * hw;
    const int64_t r = q / W, cc = q % W;
    auto at = [&](const float* t, int64_t pos) { return t[(base + pos) * C + c]; };
    // i-pairing k: i2 = (k / (W-1)) * W + k % (W-1), i1 = k;  j-pairing k: j2 = i2 + 1, j1 = W + k
    auto pair_i = [&](int64_t k) {
      const int64_t i2 = (k / (W - 1)) * W + k % (W - 1);
      const float d = at(yh, i2) - at(yh, k), t12 = fabsf(at(y, i2) - at(y, k)) - fabsf(d);
      return -(t12 >= 0.f ? norm : -norm) * (d >= 0.f ? 1.f : -1.f);
    };
    auto pair_j = [&](int64_t k) {
      const int64_t j2 = (k / (W - 1)) * W + k % (W - 1) + 1;
      const float d = at(yh, j2) - at(yh, W + k), t34 = fabsf(at(y, j2) - at(y, W + k)) - fabsf(d);
      return -(t34 >= 0.f ? norm : -norm) * (d >= 0.f ? 1.f : -1.f);
    };
    float acc = 0.f;        // same order as the oracle's scatter visits an element is not needed: at most four terms, summed
    if (cc < W - 1) acc += pair_i(r * (W - 1) + cc);          // in ascending pairing index per kind
    if (q < m) acc -= pair_i(q);
    if (cc >= 1) acc += pair_j(r * (W - 1) + cc - 1);
    if (q >= W) acc -= pair_j(q - W);
    g[e] = acc;
  }
}
VF_API int vf_gdl_bwd(vf_ctx* ctx, const float* yhat, const float* y, float* gyhat, int B, int H, int W, int C) {
  VF_REQUIRE(H == W, "GDLCriterion needs square maps (the reference's CSubTable pairs H x (W-1) with (H-1) x W)");
  const int64_t cnt = (int64_t)B * C * (H - 1) * W;
  hipLaunchKernelGGL(k_gdl_bwd, dim3(grid_for((int64_t)B * C * H * W, 4)), dim3(256), 0, ctx->stream, yhat, y, gyhat, B, H, W, C,
                     (float)(1.0 / (double)cnt));
  VF_LAUNCH_CHECK();
  return 0;
}

__global__ __launch_bounds__(256) void k_masked_mse(const float* __restrict__ x, const float* __restrict__ xh,
                                                    const uint8_t* __restrict__ mask, float w, float* __restrict__ gx,
                                                    int64_t n, double inv_n, double* __restrict__ loss) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  double sd = 0;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += stride) {
    const double wm = (1.0 - (double)w) * mask[i] + (double)w;
    const double d = (double)x[i] - (double)xh[i];
    const double v = wm * d * d;
    if (gx) gx[i] = (float)((v >= 0 ? 1.0 : -1.0) * inv_n * wm * 2.0 * d);
    sd += fabs(v);
  }
  if (loss) block_add_double(sd * inv_n, loss);
}
VF_API int vf_masked_mse_fwd(vf_ctx* ctx, const float* x, const float* xhat, const uint8_t* mask, float w, int64_t n,
                             double* loss) {
  VF_CHECK_HIP(hipMemsetAsync(loss, 0, sizeof(double), ctx->stream));
  hipLaunchKernelGGL(k_masked_mse, dim3(grid_for(n, 8)), dim3(256), 0, ctx->stream, x, xhat, mask, w, (float*)nullptr, n,
                     1.0 / (double)n, loss);
  VF_LAUNCH_CHECK();
  return 0;
}
VF_API int vf_masked_mse_bwd(vf_ctx* ctx, const float* x, const float* xhat, const uint8_t* mask, float w, float* gx,
                             int64_t n) {
  hipLaunchKernelGGL(k_masked_mse, dim3(grid_for(n, 8)), dim3(256), 0, ctx->stream, x, xhat, mask, w, gx, n, 1.0 / (double)n,
                     (double*)nullptr);
  VF_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------ optim.adam
// state[0] = t (int32), state[1] = bit pattern of the fp32 step size lr*sqrt(1-b2^t)/(1-b1^t)
__global__ void k_adam_prep(int32_t* state, double lr, double b1, double b2) {
  const int t = state[0] + 1;
  state[0] = t;
  const double bc1 = 1.0 - pow(b1, (double)t), bc2 = 1.0 - pow(b2, (double)t);
  state[1] = __float_as_int((float)(lr * sqrt(bc2) / bc1));
}
// One pass: read x,g,m,v (16 B) ; write x,m,v (12 B) = 28 B/param.  fp32 op order follows optim/adam.lua.
template <bool NT>
__global__ __launch_bounds__(256) void k_adam(float* __restrict__ x, const float* __restrict__ g, float* __restrict__ m,
                                              float* __restrict__ v, int64_t n, float b1, float omb1, float b2, float omb2,
                                              float eps, const int32_t* __restrict__ state) {
  const float step = __int_as_float(state[1]);
  const int64_t stride = (int64_t)gridDim.x * blockDim.x, n4 = n >> 2;
  auto upd = [&](float& xv, float gv, float& mv, float& vv) { vf_adam_upd(xv, gv, mv, vv, b1, omb1, b2, omb2, eps, step); };
  auto ld = [&](const float* p, int64_t i) {
    if constexpr (NT) return __builtin_nontemporal_load((const f32x4*)p + i);
    else return ((const f32x4*)p)[i];
  };
  auto st = [&](float* p, int64_t i, f32x4 val) {
    if constexpr (NT) __builtin_nontemporal_store(val, (f32x4*)p + i);
    else ((f32x4*)p)[i] = val;
  };
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += stride) {
    f32x4 xv = ld(x, i), mv = ld(m, i), vv = ld(v, i);
    const f32x4 gv = ld(g, i);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float xe = xv[e], me = mv[e], ve = vv[e];
      upd(xe, gv[e], me, ve);
      xv[e] = xe;
      mv[e] = me;
      vv[e] = ve;
    }
    st(x, i, xv);
    st(m, i, mv);
    st(v, i, vv);
  }
  for (int64_t i = (n4 << 2) + blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += stride) upd(x[i], g[i], m[i], v[i]);
}
static int adam_apply(vf_ctx* ctx, float* x, const float* g, float* m, float* v, int64_t n, double beta1, double beta2, double eps,
                      const int32_t* t_dev) {
  VF_REQUIRE((((uintptr_t)x | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0, "adam operands must be 16-byte aligned");
  static const int tune_nt = getenv("VF_ADAM_NT") ? atoi(getenv("VF_ADAM_NT")) : 0;
  static const int tune_blocks = getenv("VF_ADAM_BLOCKS") ? atoi(getenv("VF_ADAM_BLOCKS")) : 2048;
  const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>(vf_cdiv(n, 1024), tune_blocks));
  if (tune_nt)
    VF_LAUNCH_TIMED(ctx, "adam", 0.0, 28.0 * (double)n, k_adam<true>, dim3(blocks), dim3(256), x, g, m, v, n, (float)beta1,
                    (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps, t_dev);
  else
    VF_LAUNCH_TIMED(ctx, "adam", 0.0, 28.0 * (double)n, k_adam<false>, dim3(blocks), dim3(256), x, g, m, v, n, (float)beta1,
                    (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps, t_dev);
  VF_LAUNCH_CHECK();
  return 0;
}
// the same pass over up to VF_ADAM_RANGES element ranges of the flat vectors in ONE launch (the generator's vector minus the slices
// the fused bottleneck update takes: three ranges): a thread's float4 index walks the ranges laid end to end
#define VF_ADAM_RANGES 8
struct VfAdamRanges {
  int n;
  int64_t off4[VF_ADAM_RANGES];       // range start / 4
  int64_t end4[VF_ADAM_RANGES + 1];   // running total of float4s: range r covers virtual indices [end4[r], end4[r + 1])
};
__global__ __launch_bounds__(256) void k_adam_ranges(float* __restrict__ x, const float* __restrict__ g, float* __restrict__ m,
                                                     float* __restrict__ v, const VfAdamRanges R, float b1, float omb1, float b2, float omb2,
                                                     float eps, const int32_t* __restrict__ state) {
  const float step = __int_as_float(state[1]);
  const int64_t stride = (int64_t)gridDim.x * blockDim.x, total4 = R.end4[R.n];
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total4; i += stride) {
    int r = 0;
    while (r + 1 < R.n && i >= R.end4[r + 1]) ++r;
    const int64_t j = R.off4[r] + (i - R.end4[r]);
    f32x4 xv = ((const f32x4*)x)[j], mv = ((const f32x4*)m)[j], vv = ((const f32x4*)v)[j];
    const f32x4 gv = ((const f32x4*)g)[j];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float xe = xv[e], me = mv[e], ve = vv[e];
      vf_adam_upd(xe, gv[e], me, ve, b1, omb1, b2, omb2, eps, step);
      xv[e] = xe;
      mv[e] = me;
      vv[e] = ve;
    }
    ((f32x4*)x)[j] = xv;
    ((f32x4*)m)[j] = mv;
    ((f32x4*)v)[j] = vv;
  }
}
VF_API int vf_adam_apply_ranges(vf_ctx* ctx, float* x, const float* g, float* m, float* v, const int64_t* offsets, const int64_t* lengths,
                                int nranges, double beta1, double beta2, double eps, const int32_t* t_dev) {
  VF_REQUIRE(ctx && x && g && m && v && offsets && lengths && t_dev, "vf_adam_apply_ranges: NULL argument");
  VF_REQUIRE(nranges >= 1 && nranges <= VF_ADAM_RANGES, "vf_adam_apply_ranges: %d ranges (1 .. %d)", nranges, VF_ADAM_RANGES);
  VF_REQUIRE((((uintptr_t)x | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0, "adam operands must be 16-byte aligned");
  VfAdamRanges R;
  memset(&R, 0, sizeof(R));
  int k = 0;
  for (int i = 0; i < nranges; ++i) {
    VF_REQUIRE(offsets[i] >= 0 && lengths[i] >= 0 && offsets[i] % 4 == 0 && lengths[i] % 4 == 0,
               "vf_adam_apply_ranges: range %d (offset %lld, length %lld) must be whole float4s", i, (long long)offsets[i], (long long)lengths[i]);
    if (lengths[i] == 0) continue;
    R.off4[k] = offsets[i] / 4;
    R.end4[k + 1] = R.end4[k] + lengths[i] / 4;
    ++k;
  }
  R.n = k;
  if (k == 0) return 0;
  const int64_t n = 4 * R.end4[k];
  static const int tune_blocks = getenv("VF_ADAM_BLOCKS") ? atoi(getenv("VF_ADAM_BLOCKS")) : 2048;
  const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>(vf_cdiv(n, 1024), tune_blocks));
  VF_LAUNCH_TIMED(ctx, "adam", 0.0, 28.0 * (double)n, k_adam_ranges, dim3(blocks), dim3(256), x, g, m, v, R, (float)beta1,
                  (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps, t_dev);
  VF_LAUNCH_CHECK();
  return 0;
}
VF_API int vf_adam_prep(vf_ctx* ctx, double lr, double beta1, double beta2, int32_t* t_dev) {
  hipLaunchKernelGGL(k_adam_prep, dim3(1), dim3(1), 0, ctx->stream, t_dev, lr, beta1, beta2);
  VF_LAUNCH_CHECK();
  return 0;
}
VF_API int vf_adam_apply(vf_ctx* ctx, float* x, const float* g, float* m, float* v, int64_t n, double beta1, double beta2,
                         double eps, const int32_t* t_dev) {
  return adam_apply(ctx, x, g, m, v, n, beta1, beta2, eps, t_dev);
}
VF_API int vf_adam_step(vf_ctx* ctx, float* x, const float* g, float* m, float* v, int64_t n, double lr, double beta1,
                        double beta2, double eps, int32_t* t_dev) {
  if (int rc = vf_adam_prep(ctx, lr, beta1, beta2, t_dev)) return rc;
  return adam_apply(ctx, x, g, m, v, n, beta1, beta2, eps, t_dev);
}
