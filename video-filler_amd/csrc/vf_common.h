// vf_common.h — shared host-side plumbing for the gfx950 backend (context, error reporting, launch checks).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>

#include "../../include/vf_hip.h"

#define VF_API extern "C" __attribute__((visibility("default")))

// BatchNorm statistics as a by-product of the GEMM that produces the tensor (vf_bn_fuse_next_fwd / _bwd): the epilogue (or
// the split-K slab reduce) of the NEXT conv-like launch leaves per-tile partial sums that vf_bn_train_fwd_pre /
// vf_bn_bwd_pre finalize, so the tensor is not read again just to be summed.
struct VfBnSt {
  int mode;               // 0 none; 1 forward: s1 = sum(v - shift), s2 = sum((v - shift)^2); 2 backward: s1 = sum(g), s2 = sum(g*(x - mean)),
                          //   g = the launch's output masked by the activation derivative (the output IS stored masked)
  const float* vec;       // mode 1: [C] shift (running_mean); mode 2: [groups][C] save_mean
  const float* x;         // mode 2: the BatchNorm input at the output's index
  double* part;           // [groups][rows_per_group][2][C]
  int rows_per_group;     // partial rows each group ends up with (what the finalize kernels walk)
  int tiles_per_group;    // GEMM epilogue: row tiles per batch group; slab reduce: blocks per group
  int zpar;               // GEMM epilogue: output-parity classes sharing a row tile (1 or 4)
};

struct vf_ctx {
  int device;
  hipStream_t stream;
  void* ws;         // caller-owned scratch (split-K slabs, reduction partials)
  size_t ws_bytes;
  size_t ws_front;  // bytes at the front of ws currently held by an im2col / column buffer (thin-channel passes)
  int mfma_bf16;    // 0: native f32 MFMA; 1: operands rounded to bf16; 3 (default): exact three-plane bf16 split
  int wg_active;    // a weight-gradient group is being recorded (vf_wgrad_group_begin .. _end)
  void* wg_rec;     // the recorder (vf_conv.hip)
  // one-shot attachment for the next conv-like launch (vf_bn_fuse_next_*), and what that launch made of it
  VfBnSt bnf;
  const float* bnf_yact;  // mode 2: activated BatchNorm output (the derivative mask), its activation
  int bnf_act;
  float bnf_slope;
  int bnf_groups, bnf_rows_cap;
  int bnf_result_rows;    // rows_per_group of the fused launch; 0: not fused
  // sign bits of an activated tensor (vf_net.hip): one-shot attachments like bnf.  act_bits_out: the next thin-input conv forward
  // (vf_conv_thin.hip) also writes, per output pixel and 64-channel group, two words — bit j of word h = (channel 2j + h) > 0;
  // act_bits_written says whether it did.  dmask_bits: the next planes-fed transposed pass (vf_pgemm.hip) reads its activation-
  // derivative mask from such bits (a broadcast word per pixel) instead of the fp32 activation — 2 MB instead of 67 MB for E2's
  // data-gradient (train.lua:90: the LeakyReLU below the second conv)
  unsigned* act_bits_out;
  int act_bits_written;
  const unsigned* dmask_bits;
  // one-shot (vf_net.hip): the next 512 -> 1 head pass of vf_conv.hip (k_dot_bwd_data / k_dot_bwd_weight: netD's last conv,
  // train.lua:195-196) multiplies its gradOutput by the derivative of the Sigmoid fused into that conv, evaluated from the
  // activated output — instead of a pass of its own over B values in front of it
  const float* dot_act_y;
  int dot_act;
  float dot_act_slope;
};
void vf_internal_wg_free(vf_ctx* ctx);

static inline char* vf_ws_ptr(vf_ctx* c) { return (char*)c->ws + c->ws_front; }
static inline size_t vf_ws_avail(vf_ctx* c) { return c->ws_bytes > c->ws_front ? c->ws_bytes - c->ws_front : 0; }

void vf_set_error(const char* fmt, ...);

#define VF_CHECK_HIP(expr)                                                              \
  do {                                                                                  \
    hipError_t _e = (expr);                                                             \
    if (_e != hipSuccess) {                                                             \
      vf_set_error("%s:%d: %s failed: %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
      return 1;                                                                         \
    }                                                                                   \
  } while (0)

#define VF_REQUIRE(cond, ...)      \
  do {                             \
    if (!(cond)) {                 \
      vf_set_error(__VA_ARGS__);   \
      return 2;                    \
    }                              \
  } while (0)

#define VF_LAUNCH_CHECK() VF_CHECK_HIP(hipGetLastError())

// per-launch profiling scope (active only between vf_prof_begin / vf_prof_end)
bool vf_prof_enabled();
void vf_prof_push(vf_ctx* ctx, const char* name, double flops, double bytes, bool begin);
// roctx range around a launch site (vf_trace.hip): active with vf_trace_enable(1) / VF_ROCTX=1, free otherwise
bool vf_trace_enabled();
extern "C" int vf_range_push(const char* name);
extern "C" int vf_range_pop(void);
struct VfRange {
  bool on;
  explicit VfRange(const char* name) : on(vf_trace_enabled()) {
    if (on) vf_range_push(name);
  }
  ~VfRange() {
    if (on) vf_range_pop();
  }
};
struct VfProf {
  vf_ctx* c;
  bool on;
  VfRange range;
  VfProf(vf_ctx* ctx, const char* name, double flops, double bytes) : c(ctx), on(vf_prof_enabled()), range(name) {
    if (on) vf_prof_push(c, name, flops, bytes, true);
  }
  ~VfProf() {
    if (on) vf_prof_push(c, nullptr, 0, 0, false);
  }
};

// Single-kernel launches are timed with hipExtLaunchKernelGGL's start/stop events: the elapsed time is the kernel's own
// execution (what rocprofv3's kernel trace reports), without the launch gap an event pair around the call would add.
bool vf_prof_ext(const char* name, double flops, double bytes, hipEvent_t* e0, hipEvent_t* e1);
#define VF_LAUNCH_TIMED(ctx, name, flops, bytes, kernel, grid, block, ...)                                   \
  do {                                                                                                       \
    VfRange _vr(name);                                                                                       \
    hipEvent_t _e0, _e1;                                                                                     \
    if (vf_prof_ext(name, flops, bytes, &_e0, &_e1))                                                         \
      hipExtLaunchKernelGGL(kernel, grid, block, 0, (ctx)->stream, _e0, _e1, 0, __VA_ARGS__);                \
    else                                                                                                     \
      hipLaunchKernelGGL(kernel, grid, block, 0, (ctx)->stream, __VA_ARGS__);                                \
  } while (0)

// compile-time loop: f(VfIntC<0>{}) ... f(VfIntC<N-1>{}) — the index is a constant inside the body
template <int I> struct VfIntC { static constexpr int value = I; };
template <int N, int I = 0, typename F>
__device__ __forceinline__ void vf_static_for(F&& f) {
  if constexpr (I < N) {
    f(VfIntC<I>{});
    vf_static_for<N, I + 1>(f);
  }
}

// Weight gradient from PRE-SPLIT operands (vf_pgemm.hip: k_pwgrad_group; recorded by vf_conv.hip's group recorder).
//   dW[n][tap][c] = sum_p U[p][n] * V[(b, 2 my - 1 + kh, 2 mx - 1 + kw)][c]      p = (b, my, mx) on the low-resolution grid
// Up / Vp: bf16 planes [3][pixels][channels] of the two operands (conv: U = gradOutput, V = input; full-conv: U = input,
// V = gradOutput).  out: dW, or the split-K slabs (ksplit > 1: slab s at s * Nu * 16 * Cv elements).
struct VfPWGrad {
  const void* Up;               // NULL: the fp32-fed form below
  const void* Vp;
  // fp32-fed form (the two bottleneck layers riding in the same launch: their planes do not exist, K = batch): plain matrices
  // Uf [P][Nu], Vf [P][16 * Cv]; the block splits them on their way into LDS
  const float* Uf;
  const float* Vf;
  float* out;
  unsigned u_ps, v_ps;          // plane strides in bytes
  int P, lgMh, lgMw;            // low-resolution pixels = B << (lgMh + lgMw)
  int Nu, Cv, Hv, Wv;
  int gx, gy, gz;               // column tiles (16 * Cv / 128), row tiles (Nu / 128), split-K
  int ksplit, nk;               // nk = P / 32 K steps
  float beta;                   // ksplit == 1: dW = beta * dW + sum
};
#define VF_PWG_MAX 16
struct VfPWGradGroup {
  int n;
  int blk_off[VF_PWG_MAX + 1];  // multiples of 8 (the XCD-aware tile order of a layer assumes blockIdx % 8 == local id % 8)
  VfPWGrad d[VF_PWG_MAX];
};
int vf_internal_pwgrad_group(vf_ctx* ctx, const VfPWGradGroup& G, int blocks, const char* name, double flops);

static inline int vf_ilog2(int v) {  // v must be a power of two
  int l = 0;
  while ((1 << l) < v) ++l;
  return l;
}
static inline bool vf_is_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }
static inline int64_t vf_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// optim/adam.lua's element update, fp32 in the reference's operation order (one definition for k_adam and for the weight-gradient
// kernel that applies it in its epilogue, vf_wgrad_small.hip): step = lr * sqrt(1 - b2^t) / (1 - b1^t)
__device__ __forceinline__ void vf_adam_upd(float& xv, float gv, float& mv, float& vv, float b1, float omb1, float b2, float omb2,
                                            float eps, float step) {
  float mi = mv * b1;
  mi = mi + omb1 * gv;
  float vi = vv * b2;
  vi = vi + (omb2 * gv) * gv;
  float d = sqrtf(vi);
  d = d + eps;
  mv = mi;
  vv = vi;
  xv = xv - (step * mi) / d;
}

// One layer of the fused bottleneck update (vf_wgrad_small.hip k_adam_fused_wgrad; built by vf_wgrad_adam_outer* and by vf_net.hip):
//   g = gscale * sum_k U[k][:]^T V[k][:]  consumed by optim.adam on x, m, v [Nu][Ncols]; g_out (may be NULL) also receives it.
// Batch row k lives in segment k / kps at row k % kps, the segments `seg` floats apart (seg == 0: one segment, plain rows).
// Rows [row0, row0 + Nu) of a weight matrix of ldu rows (ldu == 0: all of it — row0 = 0, ldu = Nu): U's batch rows are ldu floats
// long; x, m, v, g_out point at the matrix's first row, the kernel offsets them (data parallel: every rank forms and applies ITS rows).
struct VfFusedLayer {
  const float *U, *V;
  float *x, *m, *v, *g_out;
  int K, Nu, Ncols;
  int kps;
  int64_t seg;
  float gscale;
  int row0, ldu;
};
#define VF_FUSED_MAX 4
int vf_internal_adam_fused_multi(vf_ctx* ctx, const VfFusedLayer* layers, int nl, double beta1, double beta2, double eps,
                                 const int32_t* t_dev);

// fused activation
__device__ __forceinline__ float vf_act_apply(float v, int act, float slope) {
  switch (act) {
    case VF_ACT_LRELU: return v > 0.f ? v : v * slope;
    case VF_ACT_RELU: return v > 0.f ? v : 0.f;
    case VF_ACT_TANH: return tanhf(v);
    case VF_ACT_SIGMOID: return 1.f / (1.f + expf(-v));
    default: return v;
  }
}
// derivative factor evaluated from the ACTIVATED value y
__device__ __forceinline__ float vf_act_grad(float y, float g, int act, float slope) {
  switch (act) {
    case VF_ACT_LRELU: return y > 0.f ? g : g * slope;
    case VF_ACT_RELU: return y > 0.f ? g : 0.f;
    case VF_ACT_TANH: return g * (1.f - y * y);
    case VF_ACT_SIGMOID: return g * (1.f - y) * y;
    default: return g;
  }
}
