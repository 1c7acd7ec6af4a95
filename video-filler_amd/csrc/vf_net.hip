// vf_net.hip — nn.Sequential behind the C-ABI: the FAST path of the training iteration as one library object.
//
// The reference plugs a GPU backend in at one place: util.cudnn(net) walks netG / netD and swaps their modules
// (util.lua:108-131, called at train.lua:245-258 / train_vid_weighted.lua:330-355); from then on the drivers only call
//     net:forward(input)                      -> vf_net_forward
//     net:backward(input, gradOutput)         -> vf_net_backward           (updateGradInput + accGradParameters, scale 1)
//     net:updateGradInput(input, gradOutput)  -> vf_net_update_grad_input  (train.lua:366: netD in fGx)
//     net:getParameters()                     -> vf_net_parameters / vf_net_bind_parameters
//     net:apply(bias:zero())                  -> vf_net_zero_conv_biases   (train.lua:279-280)
//     net:zeroGradParameters(), :training(), :evaluate()
// This file is that net: the layer list the reference builds with netG:add(...) / netD:add(...) (train.lua:87-199,
// train_vid_weighted.lua:112-236) executed with every cross-layer shortcut of the hot path, on the host side of the same
// entry points a module-by-module host would call (vf_conv2d_*, vf_pconv_*, vf_bn_*, ...):
//   * an in-place LeakyReLU / ReLU (and a Tanh / Sigmoid right behind a convolution) is applied in its producer's epilogue
//     and undone in the producer's backward — or, for conv -> LeakyReLU -> conv, in the NEXT conv's data-gradient epilogue;
//   * BatchNorm statistics are a by-product of the GEMM that produces the tensor (forward) and of the data-gradient GEMM
//     above it (backward): vf_bn_fuse_next_* / vf_bn_*_pre — no statistics pass over the tensor;
//   * operands of the 4x4 stride-2 passes are bf16 PLANES split once by their producer (BatchNorm apply / backward, the
//     thin-input convolution) and handed to the consumer (vf_pconv_*, k_pwgrad_group); weight planes of the whole net are
//     refreshed in one launch per parameter update;
//   * every weight gradient of a backward walk runs in one grouped launch, every conv bias gradient in two;
//   * gradParameters:zero() is lazy (the next accumulation overwrites);
//   * netD's real and fake passes as ONE batch of 2B (train.lua:331-349): vf_net_set_batch_groups(2) makes every BatchNorm
//     keep the halves apart (statistics, running averages, backward sums per half, in order), and the generator's
//     third pass runs over the fake half only (vf_net_update_grad_input_group);
//   * a backward walk can be cut where a gradient bucket is complete (vf_net_backward_range, vf_net_bucket_split) so that
//     the data-parallel exchange of that bucket overlaps the rest of the walk; SyncBN through vf_comm_* (vf_net_set_sync_bn).
// video-filler_amd/nn.py's nn.Sequential is the module-by-module mirror of the same plan (it also serves the table modules of
// the option branches); video-filler_amd/cnet.py and lua/hipnn.lua (hipnn.Net) are the thin hosts of THIS object.
//
// Memory: activations, gradInputs, planes, BatchNorm save / partial buffers and the weight planes belong to the net; the
// flat parameter / gradient buffers and the BatchNorm running statistics are the net's own unless the host binds its own
// storage (vf_net_bind_parameters / vf_net_bind_bn_running: Torch7 keeps them in Lua-owned tensors).  Buffers that depend on
// which path a pass takes (planes) are allocated at their first use: run one iteration before capturing a hipGraph.
// Layout: activations NHWC, weights channels-last (include/vf_hip.h); the flat buffers hold {weight, bias} ({gamma, beta})
// module by module, every segment padded to 64 floats — the layout of nn.py's getParameters().
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "vf_common.h"

int vf_internal_colsum(vf_ctx* ctx, const float* g, float* gb, int64_t P, int C, float beta);      // vf_bn.hip

namespace {

// where the planes kernels pay: from 3 GFLOP per pass and 1024 GEMM rows; process-wide, like nn.py's gate
double g_gate_gflop = getenv("VF_PCONV_MIN_GFLOP") ? atof(getenv("VF_PCONV_MIN_GFLOP")) : 3.0;
int g_gate_rows = getenv("VF_PCONV_MIN_ROWS") ? atoi(getenv("VF_PCONV_MIN_ROWS")) : 1024;
const bool g_no_pconv = getenv("VF_NO_PCONV") != nullptr;
const bool g_no_act_bits = getenv("VF_NO_ACT_BITS") != nullptr;
const bool g_no_head_fuse = getenv("VF_NO_HEAD_FUSE") != nullptr;     // (A/B switch: the head's Sigmoid backward as a launch of its own)      // (A/B switch: derivative masks from the fp32 activations only)
const bool g_no_bn_fuse = getenv("VF_NO_BN_FUSE") != nullptr;
const bool g_pwgrad = !(getenv("VF_PWGRAD") && strcmp(getenv("VF_PWGRAD"), "1") != 0);

struct VfColsumDescH {      // = VfColsumDesc of vf_bn.hip (64 bytes)
  const float* g;
  float* gb;
  double* part;
  int64_t P;
  int C, cq, rows_per_block, gx, gy;
  int blk1_off, blk2_off;
  float beta;
};
static_assert(sizeof(VfColsumDescH) == 64, "descriptor layout is shared with vf_bn.hip");
struct VfWpDescH {          // = VfWpDesc of vf_pgemm.hip (48 bytes)
  const float* w;
  void* nat;
  void* tr;
  int d0, d1, gx, gz, blk_off, pad;
};
static_assert(sizeof(VfWpDescH) == 48, "descriptor layout is shared with vf_pgemm.hip");

struct Layer {
  vf_layer_desc d;
  int C = 0, H = 0, W = 0;       // input shape of the layer
  int Co = 0, Ho = 0, Wo = 0;    // output shape
  int64_t w_off = -1, b_off = -1, w_n = 0, b_n = 0;      // offsets into the flat buffers
  float* y = nullptr;            // output [B][Ho][Wo][Co] (NULL: in place on the producer's output / a view)
  float* y_own = nullptr;        // the net's own allocation behind y (y may be re-bound: vf_net_bind_output)
  float* gx = nullptr;           // gradInput [B][H][W][C]
  float* gtmp = nullptr;         // fused activation: its updateGradInput when it cannot run in place on the incoming gradient
  // BatchNorm state: running statistics [C] (own or bound), save_mean / save_invstd [G][C], sums [G][2C], partial rows
  float *rm = nullptr, *rv = nullptr, *sm = nullptr, *si = nullptr;
  double* sums = nullptr;
  double* part = nullptr;
  int part_rows_cap = 0;
  // planes (bf16 [3][n]), allocated at first use
  void* xp = nullptr;            // conv: split of its input when the producer did not write planes
  void* gp = nullptr;            // conv: split of its gradOutput for the data-gradient pass
  void* yp = nullptr;            // BatchNorm / thin-input conv: planes of the own output, written beside it
  void* bgp = nullptr;           // BatchNorm: planes of its gradInput, written by its backward
  unsigned* ybits = nullptr;     // thin-input conv + (Leaky)ReLU: sign bits of its output (vf_common.h act_bits_out), for the data-gradient above
  bool bits_live = false;        // ... written by the last forward (and nobody has edited y since)
  const void* out_planes = nullptr;     // set by the last forward: planes of y (or NULL)
  const void* grad_planes = nullptr;    // set by the last backward: planes of gx (or NULL)
  void *wp_nat = nullptr, *wp_tr = nullptr;      // weight planes, native and transposed
  bool wp_live = false;
  int wp_mode = 0;               // product mode the weight planes were written in (3: three exact planes; 1: one rounded plane)
  // operands the two GEMM passes of this layer were fed with: the weight gradient takes the same planes
  const float* x_seen = nullptr;
  const void* xp_seen = nullptr;
  const float* g_seen = nullptr;
  const void* gp_seen = nullptr;
  int fused_act = VF_ACT_NONE;   // activation applied in this layer's epilogue (the next layer is an absorbed VF_L_ACT)
  float fused_slope = 0.f;
  bool absorbed = false;         // a VF_L_ACT that its producer applies
  bool fresh = true;             // zeroGradParameters(): the next accGradParameters overwrites (beta = 0)
  // optim.adam in the weight-gradient kernel (vf_net_set_fused_adam): the operands of the gradient the last backward walk left out
  bool fa = false;               // a layer the fused kernel takes
  const float *fa_u = nullptr, *fa_v = nullptr;
  int fa_k = 0;                  // 0: nothing pending (the gradient was accumulated the plain way, or no backward ran)
};

struct Entry {
  int main;       // layer index
  int act;        // absorbed activation layer index, or -1
};

inline bool is_conv(const Layer& l) { return l.d.kind == VF_L_CONV || l.d.kind == VF_L_FULLCONV; }
inline bool is_full(const Layer& l) { return l.d.kind == VF_L_FULLCONV; }
inline bool is_s2(const Layer& l) { return l.d.k == 4 && l.d.stride == 2 && l.d.pad == 1; }
inline bool relu_like(int act) { return act == VF_ACT_LRELU || act == VF_ACT_RELU; }

}  // namespace

struct vf_net {
  vf_ctx* ctx = nullptr;
  int B = 0, C0 = 0, H0 = 0, W0 = 0;
  std::vector<Layer> L;
  std::vector<Entry> plan;
  float *params = nullptr, *grads = nullptr;          // own or bound
  float *params_own = nullptr, *grads_own = nullptr;
  int64_t nparams = 0;
  bool train = true;
  int groups = 1;                  // batch groups of the next forward (BatchNorm statistics per group)
  int gcap = 0;                    // groups the BatchNorm state buffers are sized for
  bool skip_input_grad = false;    // vf_net_backward leaves out the first layer's gradInput (the drivers never read it)
  bool wp_managed = false;         // the host calls vf_net_refresh_weight_planes after every parameter update
  int act_done_at = -1;            // a cut walk resumes: the activation backward of entry `at`'s output is already applied
  int bn_pre_at = -1, bn_pre_rows = 0;
  vf_comm* comm = nullptr;         // SyncBN: sums all-reduced over the communicator
  int sync_world = 1;
  bool sync_force = false;
  vf_net_act_observer observer = nullptr;
  void* observer_user = nullptr;
  // device tables built once per combination and kept (the same walk recurs every iteration)
  std::map<std::string, std::pair<void*, std::vector<int>>> colsum_plans;
  void* wp_table = nullptr;
  int wp_n = 0, wp_blocks = 0;
  std::vector<int> wp_key;
  int64_t* bias_offs = nullptr;    // device: conv bias segments of the flat buffer (train.lua:279)
  int64_t* bias_lens = nullptr;
  int nbias = 0;
  struct BothTable {
    int64_t *offs, *lens;
    const float *base, *other_base;      // the flat buffers the offsets were computed between
  };
  std::map<const vf_net*, BothTable> both_tables;
  bool fused_adam = false;         // vf_net_set_fused_adam
  std::vector<int> fa_layers;
  vf_comm* fwd_wait_comm = nullptr;     // vf_net_forward_wait_fused: collectives the next forward waits for in front of its first
  std::vector<int> fwd_wait_tickets;    // bottleneck layer (one-shot)
  bool split_pending = false;      // vf_net_backward_split ran: the gradients of the entries below the cut are still recorded
  std::string pending_bias;        // (their deferred bias gradients, as a byte image of Deferred[])
  std::vector<void*> owned;        // parameter-lifetime allocations
  std::vector<void*> act_owned;    // shape-lifetime allocations (freed by vf_net_reshape)
};

namespace {

int net_alloc(std::vector<void*>& pool, void** out, size_t bytes) {
  void* p = nullptr;
  VF_CHECK_HIP(hipMalloc(&p, bytes ? bytes : 4));
  pool.push_back(p);
  *out = p;
  return 0;
}

bool sync_on(const vf_net* n) { return (n->comm && n->sync_world > 1) || n->sync_force; }
bool bn_fusable(const vf_net* n, const Layer& l) { return n->train && !sync_on(n) && l.C % 4 == 0 && !g_no_bn_fuse; }

// could any pass of this layer use the planes kernels (4x4 stride 2, wide enough)?
bool pconv_layer(const Layer& l) {
  return is_conv(l) && is_s2(l) && std::min(l.d.nin, l.d.nout) >= 32 && l.d.nin % 4 == 0 && l.d.nout % 4 == 0;
}
bool pconv_ok(const vf_net* n, const Layer& l, int Bn, int Hg, int Wg, int Cgather, int Cout, bool transposed) {
  const int mode = n->ctx->mfma_bf16;      // 3: three exact planes; 1: one plane rounded to bf16 by the producer
  if (g_no_pconv || (mode != 3 && mode != 1) || !is_s2(l)) return false;
  const int64_t rows = (int64_t)Bn * Hg * Wg / (transposed ? 1 : 4);
  const double gflop = 2.0 * (double)rows * 16.0 * Cgather * Cout * 1e-9;
  if (rows < g_gate_rows || gflop < g_gate_gflop) return false;
  return vf_pconv_supported_in_mode(mode, Bn, Hg, Wg, Cgather, Cout, 4, 2, 1, transposed ? 1 : 0) != 0;
}

int ensure_planes(vf_net* n, void** slot, int64_t numel) {
  if (*slot) return 0;
  return net_alloc(n->act_owned, slot, (size_t)numel * 6);
}

// the entry's output tensor (what the next entry reads): the nearest layer at or before it that owns one
const float* entry_out(const vf_net* n, int idx, const float* x_in) {
  for (int e = idx; e >= 0; --e) {
    const Layer& l = n->L[n->plan[e].main];
    if (l.y) return l.y;
  }
  return x_in;
}

int run_observer(vf_net* n, int act_layer, float* y, int64_t numel, const void* planes) {
  if (!n->observer) return 0;
  VF_CHECK_HIP(hipGetLastError());
  const int edited = n->observer(n->observer_user, act_layer, y, numel);
  if (edited < 0) {
    vf_set_error("vf_net: the activation observer reported an error at layer %d", act_layer);
    return 2;
  }
  // the producer's planes stay the consumer's operand (that hand-off is what ships); only an EDITED tensor is re-split
  if (edited > 0 && planes) return vf_planes_split(n->ctx, y, const_cast<void*>(planes), numel);
  return 0;
}

// ---- weight planes -------------------------------------------------------------------------------------------------
int weight_planes(vf_net* n, Layer& l, bool transposed, const void** out) {
  const float* w = n->params + l.w_off;
  const int d0 = is_full(l) ? l.d.nin : l.d.nout, d1 = is_full(l) ? l.d.nout : l.d.nin;
  if (!l.wp_live || l.wp_mode != n->ctx->mfma_bf16) {
    // first use (or the product mode changed: another plane format): split now; from here on the net's one-launch refresh keeps this layer's planes current
    if (!l.wp_nat) {
      if (int rc = net_alloc(n->owned, &l.wp_nat, (size_t)l.w_n * 6)) return rc;
      if (int rc = net_alloc(n->owned, &l.wp_tr, (size_t)l.w_n * 6)) return rc;
    }
    if (int rc = vf_weight_planes(n->ctx, w, l.wp_nat, l.wp_tr, d0, d1)) return rc;
    l.wp_live = true;
    l.wp_mode = n->ctx->mfma_bf16;
  }
  *out = transposed ? l.wp_tr : l.wp_nat;
  return 0;
}

int refresh_weight_planes(vf_net* n) {
  if (g_no_pconv || (n->ctx->mfma_bf16 != 3 && n->ctx->mfma_bf16 != 1)) return 0;
  std::vector<int> live;
  for (size_t i = 0; i < n->L.size(); ++i)
    if (n->L[i].wp_live && n->L[i].wp_mode == n->ctx->mfma_bf16) live.push_back((int)i);
  if (live.empty()) return 0;
  if (live != n->wp_key || !n->wp_table) {
    std::vector<VfWpDescH> desc(live.size());
    int blocks = 0;
    for (size_t j = 0; j < live.size(); ++j) {
      Layer& l = n->L[live[j]];
      const int d0 = is_full(l) ? l.d.nin : l.d.nout, d1 = is_full(l) ? l.d.nout : l.d.nin;
      const int gx = (d0 + 31) / 32, gz = (d1 + 31) / 32;
      desc[j] = VfWpDescH{n->params + l.w_off, l.wp_nat, l.wp_tr, d0, d1, gx, gz, blocks, 0};
      blocks += gx * 16 * gz;
    }
    void* dev = nullptr;       // a fresh table: a launch still in flight may be reading the old one (kept until destroy)
    if (int rc = net_alloc(n->owned, &dev, desc.size() * sizeof(VfWpDescH))) return rc;
    VF_CHECK_HIP(hipMemcpy(dev, desc.data(), desc.size() * sizeof(VfWpDescH), hipMemcpyHostToDevice));
    n->wp_table = dev;
    n->wp_n = (int)live.size();
    n->wp_blocks = blocks;
    n->wp_key = live;
  }
  return vf_weight_planes_multi(n->ctx, n->wp_table, n->wp_n, n->wp_blocks);
}

// ---- forward pieces ------------------------------------------------------------------------------------------------
int conv_forward(vf_net* n, Layer& l, const float* x, const void* in_planes, int Bn, int act, float slope, bool want_planes) {
  vf_ctx* ctx = n->ctx;
  const float* w = n->params + l.w_off;
  const float* b = n->params + l.b_off;
  const bool full = is_full(l);
  l.out_planes = nullptr;
  // every branch below rewrites l.y: sign bits a previous forward left (another planes gate, product mode or VF_NO_THIN then) no longer
  // describe it, whichever branch this forward takes
  l.bits_live = false;
  const bool simple_act = act == VF_ACT_NONE || relu_like(act);
  if (want_planes && !full && l.C == 3 && simple_act && is_s2(l) && l.Co % 64 == 0 && l.H % 16 == 0 && l.W % 16 == 0) {
    if (int rc = ensure_planes(n, &l.yp, (int64_t)n->B * l.Ho * l.Wo * l.Co)) return rc;
    // the sign bits of the activated output, for the derivative mask of the data-gradient pass above (2 MB read there instead of the
    // 67 MB activation); not under an activation observer, whose edits of y the bits would not follow (tests/helpers.py KinkSync)
    if (relu_like(act) && !n->observer && !g_no_act_bits) {
      if (!l.ybits)
        if (int rc = net_alloc(n->act_owned, (void**)&l.ybits, sizeof(unsigned) * (size_t)n->B * l.Ho * l.Wo * (l.Co / 32))) return rc;
      ctx->act_bits_out = l.ybits;
    }
    const int frc = vf_conv2d_fwd_planes(ctx, x, w, b, l.y, l.yp, Bn, l.H, l.W, l.C, l.Co, l.d.k, l.d.stride, l.d.pad, act, slope);
    l.bits_live = ctx->act_bits_out == nullptr && ctx->act_bits_written && relu_like(act) && !n->observer && !g_no_act_bits;
    ctx->act_bits_out = nullptr;
    ctx->act_bits_written = 0;
    if (frc) return frc;
    l.out_planes = l.yp;
    l.x_seen = nullptr;
    return 0;
  }
  if (simple_act && pconv_ok(n, l, Bn, l.H, l.W, l.C, l.Co, full)) {
    const void* xp = in_planes;
    if (!xp) {
      if (int rc = ensure_planes(n, &l.xp, (int64_t)n->B * l.H * l.W * l.C)) return rc;
      if (int rc = vf_planes_split(ctx, x, l.xp, (int64_t)Bn * l.H * l.W * l.C)) return rc;
      xp = l.xp;
    }
    l.x_seen = x;
    l.xp_seen = xp;
    const void* wp = nullptr;
    if (int rc = weight_planes(n, l, full, &wp)) return rc;
    return full ? vf_pconv_scatter(ctx, xp, wp, b, l.y, Bn, l.H, l.W, l.C, l.Co, act, slope, nullptr, VF_ACT_NONE, 0.f)
                : vf_pconv_gather(ctx, xp, wp, b, l.y, Bn, l.H, l.W, l.C, l.Co, act, slope);
  }
  l.x_seen = nullptr;
  return full ? vf_deconv2d_fwd(ctx, x, w, b, l.y, Bn, l.H, l.W, l.C, l.Co, l.d.k, l.d.stride, l.d.pad, act, slope)
              : vf_conv2d_fwd(ctx, x, w, b, l.y, Bn, l.H, l.W, l.C, l.Co, l.d.k, l.d.stride, l.d.pad, act, slope);
}

int bn_forward(vf_net* n, Layer& l, const float* x, int Bn, int act, float slope, int pre_rows, bool want_planes) {
  vf_ctx* ctx = n->ctx;
  const float* gamma = n->params + l.w_off;
  const float* beta = n->params + l.b_off;
  const int G = n->groups;
  const int64_t npix = (int64_t)Bn * l.H * l.W;
  l.out_planes = nullptr;
  if (!n->train) return vf_bn_eval_fwd(ctx, x, l.y, gamma, beta, l.rm, l.rv, npix, l.C, l.d.eps, act, slope);
  VF_REQUIRE(Bn % G == 0, "vf_net_forward: the batch (%d) does not divide into %d groups", Bn, G);
  void* yp = nullptr;
  if (want_planes && !sync_on(n)) {
    if (int rc = ensure_planes(n, &l.yp, (int64_t)n->B * l.H * l.W * l.C)) return rc;
    yp = l.yp;
  }
  if (pre_rows > 0) {
    int rc = vf_bn_train_fwd_pre(ctx, l.part, pre_rows, x, l.y, gamma, beta, l.rm, l.rv, l.sm, l.si, l.sums, npix / G, l.C, G,
                                 l.d.momentum, l.d.eps, act, slope, yp);
    l.out_planes = yp;
    return rc;
  }
  if (!sync_on(n)) {
    int rc = yp ? vf_bn_train_fwd_planes(ctx, x, l.y, gamma, beta, l.rm, l.rv, l.sm, l.si, l.sums, npix / G, l.C, G, l.d.momentum,
                                         l.d.eps, act, slope, yp)
                : vf_bn_train_fwd_groups(ctx, x, l.y, gamma, beta, l.rm, l.rv, l.sm, l.si, l.sums, npix / G, l.C, G, l.d.momentum,
                                         l.d.eps, act, slope);
    l.out_planes = yp;
    return rc;
  }
  // SyncBN: the per-channel sums of all ranks, then statistics of the global batch
  VF_REQUIRE(G == 1, "SyncBN with batch groups is not part of the path");
  if (int rc = vf_bn_stats(ctx, x, l.rm, l.sums, npix, l.C)) return rc;
  if (n->comm)
    if (int rc = vf_comm_allreduce_inline(n->comm, ctx, l.sums, 2 * l.C, 1, 0)) return rc;
  if (int rc = vf_bn_finalize(ctx, l.sums, l.rm, l.rv, l.sm, l.si, npix * n->sync_world, l.C, l.d.momentum, l.d.eps)) return rc;
  return vf_bn_apply(ctx, x, l.y, gamma, beta, l.sm, l.si, npix, l.C, act, slope);
}

// ---- backward pieces -----------------------------------------------------------------------------------------------
// in_act != NONE: `x` is the in-place activated output of the module below; its updateGradInput rides in this epilogue
int conv_bwd_data(vf_net* n, Layer& l, const float* x, const float* go, int Bn, int in_act, float in_slope, const void* g_planes,
                  const unsigned* in_bits = nullptr) {
  vf_ctx* ctx = n->ctx;
  const float* w = n->params + l.w_off;
  const bool full = is_full(l);
  if (pconv_ok(n, l, Bn, l.Ho, l.Wo, l.Co, l.C, !full) && (in_act == VF_ACT_NONE || !full)) {
    const void* gp = g_planes;
    if (!gp) {
      if (int rc = ensure_planes(n, &l.gp, (int64_t)n->B * l.Ho * l.Wo * l.Co)) return rc;
      if (int rc = vf_planes_split(ctx, go, l.gp, (int64_t)Bn * l.Ho * l.Wo * l.Co)) return rc;
      gp = l.gp;
    }
    l.g_seen = go;
    l.gp_seen = gp;
    const void* wp = nullptr;
    if (int rc = weight_planes(n, l, !full, &wp)) return rc;
    if (full) return vf_pconv_gather(ctx, gp, wp, nullptr, l.gx, Bn, l.Ho, l.Wo, l.Co, l.C, VF_ACT_NONE, 0.f);
    ctx->dmask_bits = in_act != VF_ACT_NONE ? in_bits : nullptr;      // (one-shot: the mask `x` as sign bits, where its producer left them)
    const int rc = vf_pconv_scatter(ctx, gp, wp, nullptr, l.gx, Bn, l.Ho, l.Wo, l.Co, l.C, VF_ACT_NONE, 0.f,
                                    in_act != VF_ACT_NONE ? x : nullptr, in_act, in_slope);
    ctx->dmask_bits = nullptr;
    return rc;
  }
  l.g_seen = nullptr;
  if (in_act != VF_ACT_NONE)
    return vf_conv2d_bwd_data_act(ctx, go, w, l.gx, x, in_act, in_slope, Bn, l.H, l.W, l.C, l.Co, l.d.k, l.d.stride, l.d.pad);
  return full ? vf_deconv2d_bwd_data(ctx, go, w, l.gx, Bn, l.H, l.W, l.C, l.Co, l.d.k, l.d.stride, l.d.pad)
              : vf_conv2d_bwd_data(ctx, go, w, l.gx, Bn, l.H, l.W, l.C, l.Co, l.d.k, l.d.stride, l.d.pad);
}

struct Deferred {
  const float* g;
  float* gb;
  int64_t P;
  int C;
  float beta;
};

int conv_acc(vf_net* n, Layer& l, const float* x, const float* go, int Bn, std::vector<Deferred>* deferred) {
  vf_ctx* ctx = n->ctx;
  const bool full = is_full(l);
  float* gw = n->grads + l.w_off;
  float* gb = n->grads + l.b_off;
  const float beta = l.fresh ? 0.f : 1.f;
  l.fresh = false;
  if (deferred && l.Co % 4 == 0 && ((uintptr_t)go & 15) == 0) {
    deferred->push_back(Deferred{go, gb, (int64_t)Bn * l.Ho * l.Wo, l.Co, beta});
    gb = nullptr;
  }
  VF_REQUIRE(!(n->fused_adam && l.fa && l.fa_k > 0 && beta != 0.f),
             "fused Adam: a second backward pass accumulates onto a weight gradient that was never written (one backward per zeroGradParameters)");
  if (n->fused_adam && l.fa && beta == 0.f) {
    // the weight gradient is formed by vf_net_adam_fused, in the kernel that applies it: U = the 1x1-map side, V = the 4x4-map side
    l.fa_u = full ? x : go;
    l.fa_v = full ? go : x;
    l.fa_k = Bn;
    l.g_seen = nullptr;
    l.gp_seen = nullptr;
    if (!gb) return 0;
    return vf_internal_colsum(ctx, go, gb, (int64_t)Bn * l.Ho * l.Wo, l.Co, beta);
  }
  l.fa_k = 0;
  const bool planes = g_pwgrad && !g_no_pconv && l.x_seen == x && l.xp_seen && l.g_seen == go && l.gp_seen;
  const void *xp = l.xp_seen, *gp = l.gp_seen;
  l.g_seen = nullptr;        // single use: only a data-gradient pass of THIS walk may hand its planes over
  l.gp_seen = nullptr;
  if (planes)
    return full ? vf_deconv2d_bwd_weight_planes(ctx, x, go, xp, gp, gw, gb, Bn, l.H, l.W, l.C, l.Co, l.d.k, l.d.stride, l.d.pad, beta)
                : vf_conv2d_bwd_weight_planes(ctx, x, go, xp, gp, gw, gb, Bn, l.H, l.W, l.C, l.Co, l.d.k, l.d.stride, l.d.pad, beta);
  return full ? vf_deconv2d_bwd_weight(ctx, x, go, gw, gb, Bn, l.H, l.W, l.C, l.Co, l.d.k, l.d.stride, l.d.pad, beta)
              : vf_conv2d_bwd_weight(ctx, x, go, gw, gb, Bn, l.H, l.W, l.C, l.Co, l.d.k, l.d.stride, l.d.pad, beta);
}

// gsel >= 0: x / gy / y_act hold batch group gsel only (a pass over one of the concatenated batches): its saved statistics.
// pre_rows > 0: gy arrives ALREADY MASKED by the activation's derivative and its sums sit in the partial buffer.
int bn_backward(vf_net* n, Layer& l, const float* x, const float* gy, int Bn, bool want_gx, bool want_gp, int act, float slope,
                const float* y_act, int gsel, int pre_rows, bool want_planes) {
  vf_ctx* ctx = n->ctx;
  VF_REQUIRE(n->train, "vf_net_backward: BatchNorm backward in evaluate mode is not part of the path");
  const float* gamma = n->params + l.w_off;
  float* gg = want_gp ? n->grads + l.w_off : nullptr;
  float* gbt = want_gp ? n->grads + l.b_off : nullptr;
  float* gx = want_gx ? l.gx : nullptr;
  float pbeta = 1.f;
  if (want_gp) {
    pbeta = l.fresh ? 0.f : 1.f;
    l.fresh = false;
  }
  l.grad_planes = nullptr;
  const int G = gsel >= 0 ? 1 : n->groups;
  const int row = gsel >= 0 ? gsel : 0;
  float *sm = l.sm + (int64_t)row * l.C, *si = l.si + (int64_t)row * l.C;
  double* su = l.sums + (int64_t)row * 2 * l.C;
  const int64_t npix_g = (int64_t)Bn * l.H * l.W / G;
  void* gp = nullptr;
  if (want_planes && gx && !sync_on(n)) {
    if (int rc = ensure_planes(n, &l.bgp, (int64_t)n->B * l.H * l.W * l.C)) return rc;
    gp = l.bgp;
  }
  if (pre_rows > 0) {
    int rc = vf_bn_bwd_pre(ctx, l.part, pre_rows, x, gy, gx, gg, gbt, gamma, sm, si, su, npix_g, l.C, G, pbeta, gp);
    l.grad_planes = gp;
    return rc;
  }
  if (!sync_on(n)) {
    int rc = gp ? vf_bn_bwd_planes(ctx, x, y_act, gy, gx, gg, gbt, gamma, sm, si, su, npix_g, l.C, G, act, slope, pbeta, gp)
                : vf_bn_bwd_groups(ctx, x, y_act, gy, gx, gg, gbt, gamma, sm, si, su, npix_g, l.C, G, act, slope, pbeta);
    l.grad_planes = gp;
    return rc;
  }
  // SyncBN: gamma / beta gradients come from THIS rank's sums (the flat-gradient exchange adds the other ranks' shares
  // later); gradInput needs the sums over the whole global batch
  VF_REQUIRE(G == 1 && gsel < 0, "SyncBN with batch groups is not part of the path");
  const int64_t npix = (int64_t)Bn * l.H * l.W, n_total = npix * n->sync_world;
  if (int rc = vf_bn_bwd_stats(ctx, x, y_act, gy, sm, su, npix, l.C, act, slope)) return rc;
  if (want_gp)
    if (int rc = vf_bn_bwd_apply(ctx, x, y_act, gy, nullptr, gg, gbt, gamma, sm, si, su, npix, n_total, l.C, act, slope, pbeta)) return rc;
  if (n->comm)
    if (int rc = vf_comm_allreduce_inline(n->comm, ctx, su, 2 * l.C, 1, 0)) return rc;
  if (want_gx) return vf_bn_bwd_apply(ctx, x, y_act, gy, gx, nullptr, nullptr, gamma, sm, si, su, npix, n_total, l.C, act, slope, 1.f);
  return 0;
}

// every deferred gradBias of a walk: two launches; the descriptor table is built once per combination and kept on the device
int bias_grad_flush(vf_net* n, const std::vector<Deferred>& items) {
  if (items.empty()) return 0;
  std::string key((const char*)items.data(), items.size() * sizeof(Deferred));
  auto it = n->colsum_plans.find(key);
  if (it == n->colsum_plans.end()) {
    std::vector<VfColsumDescH> desc(items.size());
    char* ws = vf_ws_ptr(n->ctx);
    size_t off = 0;
    int b1 = 0, b2 = 0;
    for (size_t i = 0; i < items.size(); ++i) {
      const Deferred& d = items[i];
      int cq, rpb, gx, gy;
      if (int rc = vf_bias_grad_plan(d.P, d.C, &cq, &rpb, &gx, &gy)) return rc;
      desc[i] = VfColsumDescH{d.g, d.gb, (double*)(ws + off), d.P, d.C, cq, rpb, gx, gy, b1, b2, d.beta};
      off += ((size_t)gx * d.C * 8 + 255) / 256 * 256;
      b1 += gx * gy;
      b2 += (d.C + 3) / 4;
    }
    VF_REQUIRE(off <= vf_ws_avail(n->ctx), "vf_net: workspace too small for the bias-gradient partials");
    void* dev = nullptr;
    if (int rc = net_alloc(n->owned, &dev, desc.size() * sizeof(VfColsumDescH))) return rc;
    VF_CHECK_HIP(hipMemcpy(dev, desc.data(), desc.size() * sizeof(VfColsumDescH), hipMemcpyHostToDevice));
    it = n->colsum_plans.emplace(key, std::make_pair(dev, std::vector<int>{(int)items.size(), b1, b2})).first;
  }
  const std::vector<int>& g = it->second.second;
  return vf_bias_grad_multi(n->ctx, it->second.first, g[0], g[1], g[2]);
}

// Backward over plan entries hi-1 .. lo.  gi >= 0: the pass covers batch group gi of the G the last forward ran
// concatenated — saved activations are sliced to that group's samples; x_in / gy hold that group only.
// defer_cut >= 0: one uninterrupted walk; the gradients of entries >= defer_cut are launched at its end, the rest stay recorded
// (vf_net_backward_finish)
int net_walk_back(vf_net* n, const float* x_in, const float* gy, const float** gx_out, bool want_gp, bool need_input_grad, int hi,
                  int lo, int gi, int G, int defer_cut = -1) {
  vf_ctx* ctx = n->ctx;
  const int np = (int)n->plan.size();
  if (hi < 0 || hi > np) hi = np;
  VF_REQUIRE(lo >= 0 && lo <= hi, "vf_net backward: bad plan range [%d, %d)", lo, hi);
  VF_REQUIRE(gi < 0 || (!want_gp && G >= 1 && gi < G && n->B % G == 0), "vf_net: bad batch group %d of %d", gi, G);
  VF_REQUIRE(!(want_gp && n->split_pending), "vf_net: a split backward is pending — call vf_net_backward_finish first");
  const int Bn = gi >= 0 ? n->B / G : n->B;
  int cut_wg = -1;
  size_t cut_bias = 0;
  const float* g = gy;
  const void* g_pl = nullptr;
  std::vector<Deferred> deferred;
  bool act_done = hi < np ? (n->act_done_at == hi) : false;
  int bn_pre = 0, pre = 0;
  if (!n->wp_managed)
    if (int rc = refresh_weight_planes(n)) return rc;
  if (hi < np && n->bn_pre_at == hi) bn_pre = n->bn_pre_rows;      // a walk cut between a convolution and the BatchNorm below it
  n->bn_pre_at = -1;
  int rc = 0;
  if (want_gp && (rc = vf_wgrad_group_begin(ctx))) return rc;
  auto slice = [&](const float* p, int64_t per_sample) { return (gi >= 0 && p) ? p + (int64_t)gi * Bn * per_sample : p; };
  for (int idx = hi - 1; idx >= lo && !rc; --idx) {
    if (want_gp && defer_cut >= 0 && idx == defer_cut - 1) {      // everything recorded so far belongs to the finished bucket
      if ((rc = vf_wgrad_group_count(ctx, &cut_wg))) break;
      cut_bias = deferred.size();
    }
    const Entry& e = n->plan[idx];
    Layer& l = n->L[e.main];
    const float* x = idx == 0 ? x_in : slice(entry_out(n, idx - 1, nullptr), (int64_t)l.H * l.W * l.C);
    const float* mout = slice(entry_out(n, idx, nullptr), (int64_t)l.Ho * l.Wo * l.Co);
    const bool want_gx = need_input_grad || idx > 0;
    const int64_t out_e = (int64_t)Bn * l.Ho * l.Wo * l.Co;
    if (l.d.kind == VF_L_BN) {
      const int rows = bn_pre;
      bn_pre = 0;
      // the convolution below consumes this gradient in its data-gradient pass: write its planes here
      bool want_pl = false;
      if (want_gx && idx > 0) {
        const Layer& below = n->L[n->plan[idx - 1].main];
        want_pl = is_conv(below) && pconv_layer(below) && (idx - 1 > 0 || need_input_grad) &&
                  pconv_ok(n, below, Bn, l.H, l.W, below.Co, below.C, !is_full(below));
      }
      rc = bn_backward(n, l, x, g, Bn, want_gx, want_gp, l.fused_act, l.fused_slope, l.fused_act != VF_ACT_NONE ? mout : nullptr,
                       gi, rows, want_pl);
      g = l.gx;
      g_pl = l.grad_planes;
      act_done = false;
      continue;
    }
    if (is_conv(l)) {
      const float* go = g;
      // netD's 512 -> 1 head (train.lua:195-196: conv 4x4 on the 4x4 map + Sigmoid): the Sigmoid's derivative rides in the two dot
      // kernels of its backward pass (vf_ctx::dot_act_y) instead of a launch of its own over B values
      const bool dot_head = l.d.kind == VF_L_CONV && l.Co == 1 && l.d.k == 4 && l.d.stride == 1 && l.d.pad == 0 && l.H == 4 && l.W == 4 &&
                            l.fused_act != VF_ACT_NONE && !relu_like(l.fused_act) && !act_done && !g_no_head_fuse;
      if (l.fused_act != VF_ACT_NONE && !act_done && !dot_head) {
        // undo the activation applied in this layer's epilogue: in place on the incoming gradient where that is ours (an
        // in-place module does exactly that), else — the caller's gradOutput, Tanh / Sigmoid — into a buffer of its own
        const bool ours = g != gy;
        if (relu_like(l.fused_act) && ours) {
          rc = vf_act_bwd(ctx, mout, g, const_cast<float*>(g), out_e, l.fused_act, l.fused_slope);
          g_pl = nullptr;      // (planes a BatchNorm above wrote hold the UNMASKED gradient: conv + ReLU -> BN chains)
        } else {
          if (!l.gtmp && (rc = net_alloc(n->act_owned, (void**)&l.gtmp, sizeof(float) * (size_t)n->B * l.Ho * l.Wo * l.Co))) break;
          rc = vf_act_bwd(ctx, mout, g, l.gtmp, out_e, l.fused_act, l.fused_slope);
          go = l.gtmp;
          g_pl = nullptr;
        }
        if (rc) break;
      }
      // the module below is a bare conv + in-place (leaky) ReLU: its activation backward rides in this module's
      // data-gradient epilogue (x IS that activated output)
      int in_act = VF_ACT_NONE;
      float in_slope = 0.f;
      const unsigned* in_bits = nullptr;
      if (want_gx && idx > 0 && l.d.kind == VF_L_CONV && is_s2(l)) {
        const Layer& pm = n->L[n->plan[idx - 1].main];
        if (is_conv(pm) && relu_like(pm.fused_act)) {
          in_act = pm.fused_act;
          in_slope = pm.fused_slope;
          if (pm.bits_live && pm.ybits && !n->observer)
            in_bits = pm.ybits + (gi >= 0 ? (int64_t)gi * Bn * pm.Ho * pm.Wo * (pm.Co / 32) : 0);
        }
      }
      // the module below is a BatchNorm (+ activation): this module's data-gradient pass also sums what that BatchNorm's
      // backward needs, and stores its output masked by the activation's derivative
      bool fuse_below = false;
      if (want_gx && idx > 1) {
        Layer& pm = n->L[n->plan[idx - 1].main];
        if (pm.d.kind == VF_L_BN && bn_fusable(n, pm) && (pm.fused_act == VF_ACT_NONE || relu_like(pm.fused_act)) && pm.C == l.C &&
            (gi < 0 || n->groups == G)) {
          const float* xbn = slice(entry_out(n, idx - 2, nullptr), (int64_t)pm.H * pm.W * pm.C);      // the BatchNorm's input
          const float* yact = pm.fused_act != VF_ACT_NONE ? x : nullptr;                              // (x IS pm's output)
          const float* smv = gi >= 0 ? pm.sm + (int64_t)gi * pm.C : pm.sm;
          if ((rc = vf_bn_fuse_next_bwd(ctx, xbn, yact, pm.fused_act, pm.fused_slope, smv, pm.part, pm.part_rows_cap,
                                        gi >= 0 ? 1 : n->groups)))
            break;
          fuse_below = true;
        }
      }
      auto arm_head = [&]() {
        ctx->dot_act_y = dot_head ? mout : nullptr;
        ctx->dot_act = l.fused_act;
        ctx->dot_act_slope = l.fused_slope;
      };
      if (want_gx) {
        arm_head();
        rc = conv_bwd_data(n, l, x, go, Bn, fuse_below ? VF_ACT_NONE : in_act, in_slope, g_pl, fuse_below ? nullptr : in_bits);
        ctx->dot_act_y = nullptr;
        if (fuse_below && !rc) rc = vf_bn_fuse_result(ctx, &pre);
        if (rc) break;
      }
      if (want_gp) {
        arm_head();
        rc = conv_acc(n, l, x, go, Bn, &deferred);
        ctx->dot_act_y = nullptr;
        if (rc) break;
      }
      g = want_gx ? l.gx : nullptr;
      g_pl = nullptr;
      act_done = !fuse_below && in_act != VF_ACT_NONE && want_gx;
      bn_pre = pre;
      pre = 0;
      continue;
    }
    if (l.d.kind == VF_L_ACT && !l.absorbed) {
      rc = vf_act_bwd(ctx, mout, g, l.gx, out_e, l.d.act, l.d.slope);
      g = l.gx;
      g_pl = nullptr;
    }
    // (nn.View, an absorbed activation: same storage, same gradient)
    act_done = false;
  }
  if (want_gp) {
    if (rc) {
      (void)vf_wgrad_group_abort(ctx);      // a failure in mid-walk must not leave the group open
      return rc;
    }
    if (defer_cut >= 0 && cut_wg >= 0) {
      if ((rc = vf_wgrad_group_end_partial(ctx, cut_wg))) return rc;
      std::vector<Deferred> first(deferred.begin(), deferred.begin() + cut_bias);
      if ((rc = bias_grad_flush(n, first))) return rc;
      n->pending_bias.assign((const char*)(deferred.data() + cut_bias), (deferred.size() - cut_bias) * sizeof(Deferred));
      n->split_pending = true;
    } else {
      if ((rc = vf_wgrad_group_end(ctx))) return rc;
      if ((rc = bias_grad_flush(n, deferred))) return rc;
    }
  }
  if (rc) return rc;
  n->act_done_at = act_done ? lo : -1;      // a partial walk resumes at `lo`
  if (bn_pre && lo > 0) {
    n->bn_pre_at = lo;
    n->bn_pre_rows = bn_pre;
  }
  if (gx_out) *gx_out = g;
  return 0;
}

void free_pool(std::vector<void*>& pool) {
  for (void* p : pool) (void)hipFree(p);
  pool.clear();
}

// shapes, activation buffers and BatchNorm scratch for input [B][H][W][C]; parameters are untouched
int net_shape(vf_net* n, int B, int C, int H, int W) {
  free_pool(n->act_owned);
  n->colsum_plans.clear();      // (the tables name activation buffers that no longer exist; the device copies live in `owned`)
  n->B = B; n->C0 = C; n->H0 = H; n->W0 = W;
  n->act_done_at = n->bn_pre_at = -1;
  int c = C, h = H, w = W;
  for (size_t i = 0; i < n->L.size(); ++i) {
    Layer& l = n->L[i];
    l.C = c; l.H = h; l.W = w;
    l.Co = c; l.Ho = h; l.Wo = w;
    switch (l.d.kind) {
      case VF_L_CONV:
        VF_REQUIRE(l.d.nin == c, "vf_net: layer %d expects %d input planes, gets %d", (int)i, l.d.nin, c);
        l.Co = l.d.nout;
        l.Ho = (h + 2 * l.d.pad - l.d.k) / l.d.stride + 1;
        l.Wo = (w + 2 * l.d.pad - l.d.k) / l.d.stride + 1;
        break;
      case VF_L_FULLCONV:
        VF_REQUIRE(l.d.nin == c, "vf_net: layer %d expects %d input planes, gets %d", (int)i, l.d.nin, c);
        l.Co = l.d.nout;
        l.Ho = (h - 1) * l.d.stride - 2 * l.d.pad + l.d.k;
        l.Wo = (w - 1) * l.d.stride - 2 * l.d.pad + l.d.k;
        break;
      case VF_L_BN:
        VF_REQUIRE(l.d.nout == c, "vf_net: BatchNorm layer %d has %d channels, gets %d", (int)i, l.d.nout, c);
        break;
      default:
        break;
    }
    VF_REQUIRE(l.Ho > 0 && l.Wo > 0, "vf_net: layer %d has an empty output", (int)i);
    c = l.Co; h = l.Ho; w = l.Wo;
    l.y = l.y_own = l.gx = l.gtmp = nullptr;
    l.xp = l.gp = l.yp = l.bgp = nullptr;
    l.ybits = nullptr;
    l.bits_live = false;
    l.out_planes = l.grad_planes = nullptr;
    l.x_seen = l.g_seen = nullptr;
    l.xp_seen = l.gp_seen = nullptr;
    l.part = nullptr;
    l.sm = l.si = nullptr;
    l.sums = nullptr;
  }
  n->gcap = std::max(n->groups, 2);
  for (size_t i = 0; i < n->L.size(); ++i) {
    Layer& l = n->L[i];
    const size_t in_e = (size_t)B * l.H * l.W * l.C, out_e = (size_t)B * l.Ho * l.Wo * l.Co;
    const bool own_out = is_conv(l) || l.d.kind == VF_L_BN || (l.d.kind == VF_L_ACT && !l.absorbed);
    if (own_out) {
      if (int rc = net_alloc(n->act_owned, (void**)&l.y_own, sizeof(float) * out_e)) return rc;
      l.y = l.y_own;
    }
    if (is_conv(l) || l.d.kind == VF_L_BN || (l.d.kind == VF_L_ACT && !l.absorbed))
      if (int rc = net_alloc(n->act_owned, (void**)&l.gx, sizeof(float) * in_e)) return rc;
    if (l.d.kind == VF_L_BN) {
      const size_t gc = (size_t)n->gcap * l.C;
      if (int rc = net_alloc(n->act_owned, (void**)&l.sm, 4 * gc)) return rc;
      if (int rc = net_alloc(n->act_owned, (void**)&l.si, 4 * gc)) return rc;
      if (int rc = net_alloc(n->act_owned, (void**)&l.sums, 16 * gc)) return rc;
      // [rows][2C] double partials: one row per 64-pixel output tile of the producing GEMM (or per block of its split-K
      // combine, at most ~512)
      const int64_t npix = (int64_t)B * l.H * l.W;
      l.part_rows_cap = (int)std::max<int64_t>(npix / 64, 512) + 8 * n->gcap;
      if (int rc = net_alloc(n->act_owned, (void**)&l.part, sizeof(double) * (size_t)l.part_rows_cap * 2 * l.C)) return rc;
      (void)hipMemsetAsync(l.sm, 0, 4 * gc, n->ctx->stream);
      (void)hipMemsetAsync(l.si, 0, 4 * gc, n->ctx->stream);
      (void)hipMemsetAsync(l.sums, 0, 16 * gc, n->ctx->stream);
    }
  }
  return 0;
}

}  // namespace

VF_API int vf_net_destroy(vf_net* n) {
  if (!n) return 0;
  if (n->ctx) (void)hipStreamSynchronize(n->ctx->stream);
  free_pool(n->act_owned);
  free_pool(n->owned);
  delete n;
  return 0;
}

// layers[0..nlayers): the modules in order; the input is [B][H][W][C] (NHWC)
VF_API int vf_net_create(vf_ctx* ctx, vf_net** out, const vf_layer_desc* layers, int nlayers, int B, int C, int H, int W) {
  VF_REQUIRE(ctx && out && layers && nlayers > 0 && B > 0 && C > 0 && H > 0 && W > 0, "vf_net_create: bad arguments");
  *out = nullptr;
  vf_net* n = new vf_net();
  n->ctx = ctx;
  int64_t off = 0;
  auto seg = [&](int64_t len) {
    const int64_t o = off;
    off += (len + 63) & ~(int64_t)63;
    return o;
  };
  int rc = 0;
  for (int i = 0; i < nlayers && !rc; ++i) {
    Layer l;
    l.d = layers[i];
    switch (l.d.kind) {
      case VF_L_CONV:
        l.w_n = (int64_t)l.d.nout * l.d.k * l.d.k * l.d.nin;
        l.b_n = l.d.nout;
        break;
      case VF_L_FULLCONV:
        l.w_n = (int64_t)l.d.nin * l.d.k * l.d.k * l.d.nout;
        l.b_n = l.d.nout;
        break;
      case VF_L_BN:
        if (l.d.eps == 0.f) l.d.eps = 1e-5f;
        if (l.d.momentum == 0.f) l.d.momentum = 0.1f;
        l.w_n = l.b_n = l.d.nout;
        break;
      case VF_L_ACT:
      case VF_L_VIEW:
        break;
      default:
        vf_set_error("vf_net_create: unknown layer kind %d at %d", l.d.kind, i);
        rc = 2;
        break;
    }
    if (rc) break;
    if ((l.d.kind == VF_L_CONV || l.d.kind == VF_L_FULLCONV) && (l.d.nin <= 0 || l.d.nout <= 0 || l.d.k <= 0 || l.d.stride <= 0 || l.d.pad < 0)) {
      vf_set_error("vf_net_create: layer %d: bad convolution geometry", i);
      rc = 2;
      break;
    }
    if (l.w_n) {
      l.w_off = seg(l.w_n);
      l.b_off = seg(l.b_n);
    }
    n->L.push_back(l);
  }
  if (rc) {
    vf_net_destroy(n);
    return rc;
  }
  // execution plan: an in-place (Leaky)ReLU behind a conv / full-conv / BatchNorm and a Tanh / Sigmoid right behind a
  // convolution are applied by that layer (nn.Sequential(fuse=True)'s rule; the reference's activations are in-place modules)
  for (size_t i = 0; i < n->L.size();) {
    Layer& p = n->L[i];
    Entry e{(int)i, -1};
    if (i + 1 < n->L.size() && n->L[i + 1].d.kind == VF_L_ACT) {
      Layer& a = n->L[i + 1];
      const bool ok = (is_conv(p) || p.d.kind == VF_L_BN) ? (relu_like(a.d.act) || (is_conv(p) && (a.d.act == VF_ACT_TANH || a.d.act == VF_ACT_SIGMOID)))
                                                          : false;
      if (ok) {
        p.fused_act = a.d.act;
        p.fused_slope = a.d.slope;
        a.absorbed = true;
        e.act = (int)i + 1;
      }
    }
    n->plan.push_back(e);
    i += e.act >= 0 ? 2 : 1;
  }
  n->nparams = off;
  if ((rc = net_alloc(n->owned, (void**)&n->params_own, sizeof(float) * (size_t)off)) ||
      (rc = net_alloc(n->owned, (void**)&n->grads_own, sizeof(float) * (size_t)off))) {
    vf_net_destroy(n);
    return rc;
  }
  n->params = n->params_own;
  n->grads = n->grads_own;
  (void)hipMemsetAsync(n->params, 0, sizeof(float) * (size_t)off, ctx->stream);
  (void)hipMemsetAsync(n->grads, 0, sizeof(float) * (size_t)off, ctx->stream);
  // BatchNorm running statistics (mean 0, variance 1) and the conv-bias segment table of the sweep (train.lua:279)
  std::vector<int64_t> boffs, blens;
  for (Layer& l : n->L) {
    if (l.d.kind == VF_L_BN) {
      const int Cc = l.d.nout;
      if ((rc = net_alloc(n->owned, (void**)&l.rm, 4 * (size_t)Cc)) || (rc = net_alloc(n->owned, (void**)&l.rv, 4 * (size_t)Cc))) break;
      std::vector<float> ones((size_t)Cc, 1.f);
      (void)hipMemsetAsync(l.rm, 0, 4 * (size_t)Cc, ctx->stream);
      if (hipMemcpy(l.rv, ones.data(), 4 * (size_t)Cc, hipMemcpyHostToDevice) != hipSuccess) {
        vf_set_error("vf_net_create: hipMemcpy failed");
        rc = 1;
        break;
      }
    } else if (is_conv(l)) {
      boffs.push_back(l.b_off);
      blens.push_back(l.b_n);
    }
  }
  if (!rc && !boffs.empty()) {
    n->nbias = (int)boffs.size();
    if (!(rc = net_alloc(n->owned, (void**)&n->bias_offs, 8 * boffs.size())) && !(rc = net_alloc(n->owned, (void**)&n->bias_lens, 8 * boffs.size()))) {
      if (hipMemcpy(n->bias_offs, boffs.data(), 8 * boffs.size(), hipMemcpyHostToDevice) != hipSuccess ||
          hipMemcpy(n->bias_lens, blens.data(), 8 * blens.size(), hipMemcpyHostToDevice) != hipSuccess) {
        vf_set_error("vf_net_create: hipMemcpy failed");
        rc = 1;
      }
    }
  }
  if (!rc) rc = net_shape(n, B, C, H, W);
  if (rc) {
    vf_net_destroy(n);
    return rc;
  }
  *out = n;
  return 0;
}

// a new input shape (Torch7 modules resize their outputs on the fly; this object sizes them once per shape): activations,
// planes and BatchNorm scratch are re-planned, parameters / running statistics / weight planes stay.  Not during a capture.
VF_API int vf_net_reshape(vf_net* n, int B, int C, int H, int W) {
  VF_REQUIRE(n && B > 0 && C > 0 && H > 0 && W > 0, "vf_net_reshape: bad arguments");
  if (B == n->B && C == n->C0 && H == n->H0 && W == n->W0) return 0;
  VF_CHECK_HIP(hipStreamSynchronize(n->ctx->stream));
  return net_shape(n, B, C, H, W);
}

// net:getParameters(): the flat buffers (device pointers) and their length in floats.  Segment `which` of module `layer`
// starts at vf_net_param_offset (0 = weight / gamma, 1 = bias / beta); the host fills them (weights_init, a checkpoint).
VF_API int vf_net_parameters(vf_net* n, float** params, float** grads, int64_t* count) {
  VF_REQUIRE(n != nullptr, "vf_net_parameters: NULL net");
  if (params) *params = n->params;
  if (grads) *grads = n->grads;
  if (count) *count = n->nparams;
  return 0;
}
// Host-owned flat storage instead of the net's own (Torch7 keeps parameters in Lua-owned tensors; after getParameters()
// every weight / bias is a view of ONE storage — this is that storage): count floats each, the layout of
// vf_net_param_offset, 16-byte aligned.  Nothing is copied.  NULL, NULL goes back to the net's own buffers.
VF_API int vf_net_bind_parameters(vf_net* n, float* params, float* grads, int64_t count) {
  VF_REQUIRE(n != nullptr, "vf_net_bind_parameters: NULL net");
  if (!params && !grads) {
    n->params = n->params_own;
    n->grads = n->grads_own;
  } else {
    VF_REQUIRE(params && grads && count >= n->nparams, "vf_net_bind_parameters: need both buffers of >= %lld floats", (long long)n->nparams);
    VF_REQUIRE((((uintptr_t)params | (uintptr_t)grads) & 15) == 0, "vf_net_bind_parameters: buffers must be 16-byte aligned");
    n->params = params;
    n->grads = grads;
  }
  for (Layer& l : n->L) l.wp_live = false;      // the planes were split from the other storage
  n->wp_key.clear();
  n->colsum_plans.clear();
  n->both_tables.clear();
  return 0;
}
VF_API int64_t vf_net_param_offset(const vf_net* n, int layer, int which, int64_t* length) {
  if (!n || layer < 0 || layer >= (int)n->L.size() || n->L[layer].w_off < 0) return -1;
  const Layer& l = n->L[layer];
  if (length) *length = which ? l.b_n : l.w_n;
  return which ? l.b_off : l.w_off;
}
// running statistics of BatchNorm layer `layer` (device pointers, C floats each)
VF_API int vf_net_bn_running(vf_net* n, int layer, float** running_mean, float** running_var) {
  VF_REQUIRE(n && layer >= 0 && layer < (int)n->L.size() && n->L[layer].d.kind == VF_L_BN, "vf_net_bn_running: layer %d is not a BatchNorm", layer);
  if (running_mean) *running_mean = n->L[layer].rm;
  if (running_var) *running_var = n->L[layer].rv;
  return 0;
}
VF_API int vf_net_bind_bn_running(vf_net* n, int layer, float* running_mean, float* running_var) {
  VF_REQUIRE(n && layer >= 0 && layer < (int)n->L.size() && n->L[layer].d.kind == VF_L_BN, "vf_net_bind_bn_running: layer %d is not a BatchNorm", layer);
  VF_REQUIRE(running_mean && running_var, "vf_net_bind_bn_running: NULL buffer");
  n->L[layer].rm = running_mean;
  n->L[layer].rv = running_var;
  return 0;
}
// save_mean / save_invstd of the last training forward ([groups][C]; THNN's save_mean / save_std)
VF_API int vf_net_bn_saved(vf_net* n, int layer, float** save_mean, float** save_invstd) {
  VF_REQUIRE(n && layer >= 0 && layer < (int)n->L.size() && n->L[layer].d.kind == VF_L_BN, "vf_net_bn_saved: layer %d is not a BatchNorm", layer);
  if (save_mean) *save_mean = n->L[layer].sm;
  if (save_invstd) *save_invstd = n->L[layer].si;
  return 0;
}
VF_API int vf_net_training(vf_net* n, int train) {
  VF_REQUIRE(n != nullptr, "vf_net_training: NULL net");
  n->train = train != 0;
  return 0;
}
VF_API int vf_net_zero_grad(vf_net* n) {      // lazily: the next accGradParameters of each module overwrites
  VF_REQUIRE(n != nullptr, "vf_net_zero_grad: NULL net");
  for (Layer& l : n->L) {
    l.fresh = true;
    l.fa_k = 0;
  }
  return 0;
}
// netX:apply(function(m) if torch.type(m):find('Convolution') then m.bias:zero() end end) (train.lua:279-280): one launch
// for this net — and for `other` too when given (both closures sweep BOTH nets), its segments addressed from this net's base
VF_API int vf_net_zero_conv_biases(vf_net* n, vf_net* other) {
  VF_REQUIRE(n != nullptr, "vf_net_zero_conv_biases: NULL net");
  if (!other || other == n || other->nbias == 0) {
    if (n->nbias == 0) return 0;
    return vf_zero_segments(n->ctx, n->params, n->bias_offs, n->bias_lens, n->nbias);
  }
  auto it = n->both_tables.find(other);
  if (it != n->both_tables.end() && (it->second.base != n->params || it->second.other_base != other->params)) {
    n->both_tables.erase(it);       // one of the nets was re-bound to other storage since
    it = n->both_tables.end();
  }
  if (it == n->both_tables.end()) {
    std::vector<int64_t> offs((size_t)n->nbias + other->nbias), lens(offs.size());
    if (n->nbias) {
      VF_CHECK_HIP(hipMemcpy(offs.data(), n->bias_offs, 8 * (size_t)n->nbias, hipMemcpyDeviceToHost));
      VF_CHECK_HIP(hipMemcpy(lens.data(), n->bias_lens, 8 * (size_t)n->nbias, hipMemcpyDeviceToHost));
    }
    VF_CHECK_HIP(hipMemcpy(offs.data() + n->nbias, other->bias_offs, 8 * (size_t)other->nbias, hipMemcpyDeviceToHost));
    VF_CHECK_HIP(hipMemcpy(lens.data() + n->nbias, other->bias_lens, 8 * (size_t)other->nbias, hipMemcpyDeviceToHost));
    const int64_t delta = other->params - n->params;      // in floats
    for (int i = 0; i < other->nbias; ++i) offs[n->nbias + i] += delta;
    int64_t *d_offs = nullptr, *d_lens = nullptr;
    if (int rc = net_alloc(n->owned, (void**)&d_offs, 8 * offs.size())) return rc;
    if (int rc = net_alloc(n->owned, (void**)&d_lens, 8 * lens.size())) return rc;
    VF_CHECK_HIP(hipMemcpy(d_offs, offs.data(), 8 * offs.size(), hipMemcpyHostToDevice));
    VF_CHECK_HIP(hipMemcpy(d_lens, lens.data(), 8 * lens.size(), hipMemcpyHostToDevice));
    it = n->both_tables.emplace(other, vf_net::BothTable{d_offs, d_lens, n->params, other->params}).first;
  }
  return vf_zero_segments(n->ctx, n->params, it->second.offs, it->second.lens, n->nbias + other->nbias);
}

// output of module `layer` after a forward call (net.modules[i].output), gradInput after a backward call; device pointers
VF_API int vf_net_layer_output(vf_net* n, int layer, const float** y) {
  VF_REQUIRE(n && y && layer >= 0 && layer < (int)n->L.size(), "vf_net_layer_output: bad layer");
  const float* p = nullptr;
  for (int i = 0; i <= layer; ++i)
    if (n->L[i].y) p = n->L[i].y;
  *y = p;
  return 0;
}
VF_API int vf_net_layer_grad_input(vf_net* n, int layer, const float** gx) {
  VF_REQUIRE(n && gx && layer >= 0 && layer < (int)n->L.size(), "vf_net_layer_grad_input: bad layer");
  *gx = n->L[layer].gx;
  return 0;
}
// shape of module `layer`: input (C, H, W) and output (Co, Ho, Wo) per sample, and the batch size
VF_API int vf_net_layer_shape(const vf_net* n, int layer, int* B, int* C, int* H, int* W, int* Co, int* Ho, int* Wo) {
  VF_REQUIRE(n && layer >= 0 && layer < (int)n->L.size(), "vf_net_layer_shape: bad layer");
  const Layer& l = n->L[layer];
  if (B) *B = n->B;
  if (C) *C = l.C;
  if (H) *H = l.H;
  if (W) *W = l.W;
  if (Co) *Co = l.Co;
  if (Ho) *Ho = l.Ho;
  if (Wo) *Wo = l.Wo;
  return 0;
}
// Let module `layer` (a convolution / BatchNorm) write its output into the host's buffer instead of the net's own — e.g.
// the generator's last convolution straight into the fake half of netD's [real; fake] input (no copy per iteration).
// NULL restores the net's buffer.
VF_API int vf_net_bind_output(vf_net* n, int layer, float* y) {
  VF_REQUIRE(n && layer >= 0 && layer < (int)n->L.size() && n->L[layer].y_own, "vf_net_bind_output: layer %d owns no output", layer);
  n->L[layer].y = y ? y : n->L[layer].y_own;
  return 0;
}

// the next forwards carry G concatenated, independent batches (BatchNorm statistics, running averages and backward sums
// per group, in group order — what G separate calls would compute)
VF_API int vf_net_set_batch_groups(vf_net* n, int G) {
  VF_REQUIRE(n && G >= 1 && G <= 64, "vf_net_set_batch_groups: bad group count %d", G);
  if (G > n->gcap) {
    n->groups = G;
    VF_CHECK_HIP(hipStreamSynchronize(n->ctx->stream));
    return net_shape(n, n->B, n->C0, n->H0, n->W0);
  }
  n->groups = G;
  return 0;
}
VF_API int vf_net_set_skip_input_grad(vf_net* n, int on) {
  VF_REQUIRE(n != nullptr, "vf_net_set_skip_input_grad: NULL net");
  n->skip_input_grad = on != 0;
  return 0;
}
VF_API int vf_net_set_weight_planes_managed(vf_net* n, int on) {
  VF_REQUIRE(n != nullptr, "vf_net_set_weight_planes_managed: NULL net");
  n->wp_managed = on != 0;
  return 0;
}
VF_API int vf_net_refresh_weight_planes(vf_net* n) {
  VF_REQUIRE(n != nullptr, "vf_net_refresh_weight_planes: NULL net");
  return refresh_weight_planes(n);
}
VF_API int vf_net_set_planes_gate(double min_gflop, int min_rows) {
  g_gate_gflop = min_gflop;
  g_gate_rows = min_rows;
  return 0;
}
VF_API int vf_net_set_sync_bn(vf_net* n, vf_comm* comm, int world, int force) {
  VF_REQUIRE(n && world >= 1, "vf_net_set_sync_bn: bad arguments");
  // (a world of several ranks without a communicator would silently normalise with rank-local statistics)
  VF_REQUIRE(world == 1 || comm != nullptr, "vf_net_set_sync_bn: world = %d needs a communicator (vf_comm_init)", world);
  // bn_forward / bn_backward all-reduce the sums over EVERY rank of `comm` and divide by npix * world: the two must be the same number
  // (a forced world-1 SyncBN over a communicator of N ranks would normalise with N-times-too-large sums: ADVICE r4)
  VF_REQUIRE(comm == nullptr || vf_comm_world(comm) == world,
             "vf_net_set_sync_bn: world = %d but the communicator spans %d ranks", world, comm ? vf_comm_world(comm) : 0);
  n->comm = comm;
  n->sync_world = world;
  n->sync_force = force != 0;
  return 0;
}
VF_API int vf_net_set_act_observer(vf_net* n, vf_net_act_observer fn, void* user) {
  VF_REQUIRE(n != nullptr, "vf_net_set_act_observer: NULL net");
  n->observer = fn;
  n->observer_user = user;
  return 0;
}
// did the last forward leave the sign bits of this layer's activated output (the derivative mask of the data-gradient above)?
VF_API int vf_net_layer_has_act_bits(const vf_net* n, int layer) {
  return n && layer >= 0 && layer < (int)n->L.size() && n->L[layer].bits_live && !n->observer ? 1 : 0;
}
VF_API int vf_net_plan_size(const vf_net* n) { return n ? (int)n->plan.size() : -1; }
// (plan index k, flat offset): the shortest tail plan[k:] that owns at least `frac` of the parameters.  After a backward walk
// over plan[k:] the flat gradient [offset, end) is final: its exchange can start while the rest of the walk runs.
VF_API int vf_net_bucket_split(const vf_net* n, double frac, int* plan_index, int64_t* flat_offset) {
  VF_REQUIRE(n && plan_index && flat_offset, "vf_net_bucket_split: NULL argument");
  double total = 0;
  for (const Layer& l : n->L) total += (double)(l.w_n + l.b_n);
  double acc = 0;
  *plan_index = 0;
  *flat_offset = 0;
  for (int idx = (int)n->plan.size() - 1; idx >= 0; --idx) {
    const Layer& l = n->L[n->plan[idx].main];
    if (l.w_off < 0) continue;
    acc += (double)(l.w_n + l.b_n);
    *plan_index = idx;
    *flat_offset = l.w_off;
    if (acc >= frac * total) break;
  }
  return 0;
}

static bool fused_adam_shape(const Layer& l);
VF_API int vf_net_forward(vf_net* n, const float* x, const float** y) {
  VF_REQUIRE(n && x, "vf_net_forward: NULL argument");
  vf_ctx* ctx = n->ctx;
  const float* cur = x;
  const void* cur_pl = nullptr;      // bf16 planes of `cur`, when its producer wrote them
  int pre_rows = 0;
  if (!n->wp_managed)
    if (int rc = refresh_weight_planes(n)) return rc;
  const int np = (int)n->plan.size();
  for (int idx = 0; idx < np; ++idx) {
    const Entry& e = n->plan[idx];
    Layer& l = n->L[e.main];
    Layer* nxt = idx + 1 < np ? &n->L[n->plan[idx + 1].main] : nullptr;
    int rc = 0;
    if (!n->fwd_wait_tickets.empty() && (fused_adam_shape(l) || idx + 1 == np)) {
      // vf_net_forward_wait_fused: the parameter rows other ranks updated must have arrived before the first layer that reads them
      for (int t : n->fwd_wait_tickets)
        if ((rc = vf_comm_wait(n->fwd_wait_comm, ctx, t))) return rc;
      n->fwd_wait_tickets.clear();
      n->fwd_wait_comm = nullptr;
    }
    if (is_conv(l)) {
      if (l.fused_act == VF_ACT_NONE && nxt && nxt->d.kind == VF_L_BN && bn_fusable(n, *nxt) && nxt->d.nout == l.Co) {
        // the BatchNorm behind this convolution gets its statistics from the convolution's own epilogue
        if ((rc = vf_bn_fuse_next_fwd(ctx, nxt->rm, nxt->part, nxt->part_rows_cap, n->groups))) return rc;
        if ((rc = conv_forward(n, l, cur, cur_pl, n->B, VF_ACT_NONE, 0.f, false))) return rc;
        if ((rc = vf_bn_fuse_result(ctx, &pre_rows))) return rc;
        cur = l.y;
        cur_pl = nullptr;
        continue;
      }
      const bool want_pl = nxt && is_conv(*nxt) && pconv_layer(*nxt) &&
                           pconv_ok(n, *nxt, n->B, l.Ho, l.Wo, nxt->d.nin, nxt->d.nout, is_full(*nxt));
      if ((rc = conv_forward(n, l, cur, cur_pl, n->B, l.fused_act, l.fused_slope, want_pl))) return rc;
      cur = l.y;
      cur_pl = l.out_planes;
      if (relu_like(l.fused_act) && (rc = run_observer(n, e.act, l.y, (int64_t)n->B * l.Ho * l.Wo * l.Co, cur_pl))) return rc;
      continue;
    }
    if (l.d.kind == VF_L_BN) {
      const int rows = pre_rows;
      pre_rows = 0;
      const bool want_pl = n->train && nxt && is_conv(*nxt) && pconv_layer(*nxt) && (l.fused_act == VF_ACT_NONE || relu_like(l.fused_act)) &&
                           pconv_ok(n, *nxt, n->B, l.H, l.W, nxt->d.nin, nxt->d.nout, is_full(*nxt));
      if ((rc = bn_forward(n, l, cur, n->B, l.fused_act, l.fused_slope, rows, want_pl))) return rc;
      cur = l.y;
      cur_pl = l.out_planes;
      if (relu_like(l.fused_act) && (rc = run_observer(n, e.act, l.y, (int64_t)n->B * l.H * l.W * l.C, cur_pl))) return rc;
      continue;
    }
    cur_pl = nullptr;
    if (l.d.kind == VF_L_ACT && !l.absorbed) {
      if ((rc = vf_act_fwd(ctx, cur, l.y, (int64_t)n->B * l.H * l.W * l.C, l.d.act, l.d.slope))) return rc;
      cur = l.y;
      if (relu_like(l.d.act) && (rc = run_observer(n, e.main, l.y, (int64_t)n->B * l.H * l.W * l.C, nullptr))) return rc;
    }
    // nn.View: same storage
  }
  if (y) *y = cur;
  return 0;
}

// net:backward(input, gradOutput): gradInput + accumulated parameter gradients (every weight gradient in one grouped launch).
// With vf_net_set_skip_input_grad the first layer's gradInput is left out and *gx is NULL.
VF_API int vf_net_backward(vf_net* n, const float* x, const float* gy, const float** gx) {
  VF_REQUIRE(n && x && gy, "vf_net_backward: NULL argument");
  return net_walk_back(n, x, gy, gx, true, !n->skip_input_grad, -1, 0, -1, 1);
}
// backward() restricted to plan entries hi-1 .. lo (hi < 0: from the top); gy is what the entry above `hi` returned
VF_API int vf_net_backward_range(vf_net* n, const float* x, const float* gy, int hi, int lo, int need_input_grad, const float** gx) {
  VF_REQUIRE(n && x && gy, "vf_net_backward_range: NULL argument");
  return net_walk_back(n, x, gy, gx, true, need_input_grad != 0, hi, lo, -1, 1);
}
VF_API int vf_net_backward_split(vf_net* n, const float* x, const float* gy, int k, int need_input_grad, const float** gx) {
  VF_REQUIRE(n && x && gy && k >= 0 && k <= (int)n->plan.size(), "vf_net_backward_split: bad arguments");
  return net_walk_back(n, x, gy, gx, true, need_input_grad != 0, -1, 0, -1, 1, k);
}
VF_API int vf_net_backward_finish(vf_net* n) {
  VF_REQUIRE(n != nullptr, "vf_net_backward_finish: NULL net");
  if (!n->split_pending) return 0;
  n->split_pending = false;
  if (int rc = vf_wgrad_group_end(n->ctx)) return rc;
  std::vector<Deferred> rest(n->pending_bias.size() / sizeof(Deferred));
  memcpy(rest.data(), n->pending_bias.data(), n->pending_bias.size());
  n->pending_bias.clear();
  return bias_grad_flush(n, rest);
}
// ---- optim.adam inside the bottleneck pair's weight gradients (vf_wgrad_adam_outer, vf_wgrad_small.hip)
static bool fused_adam_shape(const Layer& l) {
  if (!is_conv(l) || l.d.k != 4 || l.d.stride != 1 || l.d.pad != 0 || l.w_off < 0 || (l.w_off & 3)) return false;
  const bool full = is_full(l);
  const int Nu = full ? l.C : l.Co, Cv = full ? l.Co : l.C;
  const bool maps = full ? (l.H == 1 && l.W == 1 && l.Ho == 4 && l.Wo == 4) : (l.H == 4 && l.W == 4 && l.Ho == 1 && l.Wo == 1);
  return maps && vf_wgrad_adam_outer_supported(1, Nu, 16 * Cv) && (int64_t)Nu * 16 * Cv == l.w_n;
}
VF_API int vf_net_set_fused_adam(vf_net* n, int on, int* count) {
  VF_REQUIRE(n != nullptr, "vf_net_set_fused_adam: NULL net");
  n->fa_layers.clear();
  for (size_t i = 0; i < n->L.size(); ++i) {
    Layer& l = n->L[i];
    l.fa = on && fused_adam_shape(l);
    l.fa_k = 0;
    if (l.fa) n->fa_layers.push_back((int)i);
  }
  n->fused_adam = !n->fa_layers.empty();
  if (count) *count = (int)n->fa_layers.size();
  return 0;
}
VF_API int vf_net_fused_adam_range(const vf_net* n, int i, int64_t* offset, int64_t* length) {
  VF_REQUIRE(n && offset && length && i >= 0 && i < (int)n->fa_layers.size(), "vf_net_fused_adam_range: layer %d of %d", i,
             n ? (int)n->fa_layers.size() : 0);
  const Layer& l = n->L[n->fa_layers[i]];
  *offset = l.w_off;
  *length = l.w_n;
  return 0;
}
static int64_t pad4(int64_t n) { return (n + 3) & ~(int64_t)3; }
// the update sharded by weight rows: rank r of `world` owns rows [r * bs, min(Nu, (r + 1) * bs)), bs = 2 * ceil(Nu / (2 * world)) — even
// blocks (the fused kernel's lanes own row pairs), equal wherever Nu splits, a shorter last block where it does not (4000 rows over
// 3 ranks: 1334, 1334, 1332)
static void fused_row_block(int Nu, int rank, int world, int* row0, int* nrows) {
  const int bs = 2 * ((Nu + 2 * world - 1) / (2 * world));
  const int r0 = std::min(Nu, rank * bs), r1 = std::min(Nu, r0 + bs);
  *row0 = r0;
  *nrows = r1 - r0;
}
// row_world > 1: this rank forms and applies rows [row_rank * Nu / row_world, (row_rank + 1) * Nu / row_world) of every fused tensor only
static int fused_layers_launch(vf_net* n, const float* all, int world, int64_t seg_stride, float* m, float* v, double beta1, double beta2,
                               double eps, const int32_t* t_dev, int keep_grad, int row_rank = 0, int row_world = 1) {
  VfFusedLayer F[VF_FUSED_MAX];
  int nf = 0;
  int64_t off = 0;
  for (int i : n->fa_layers) {
    Layer& l = n->L[i];
    const bool full = is_full(l);
    const int Nu = full ? l.C : l.Co, Cv = full ? l.Co : l.C;
    float *x = n->params + l.w_off, *g = n->grads + l.w_off;
    if (all) VF_REQUIRE(l.fa_k == n->B, "vf_net_adam_fused_gathered: layer %d has no pending gradient", i);
    if (l.fa_k <= 0) {       // accumulated the plain way (a backward pass onto a gradient that was not fresh): the plain update
      if (int rc = vf_adam_apply(n->ctx, x, g, m + l.w_off, v + l.w_off, l.w_n, beta1, beta2, eps, t_dev)) return rc;
      continue;
    }
    if (nf == VF_FUSED_MAX) {
      if (int rc = vf_internal_adam_fused_multi(n->ctx, F, nf, beta1, beta2, eps, t_dev)) return rc;
      nf = 0;
    }
    VfFusedLayer& L = F[nf++];
    if (all) {
      L.U = all + off;
      off += pad4((int64_t)l.fa_k * Nu);
      L.V = all + off;
      off += pad4((int64_t)l.fa_k * 16 * Cv);
      VF_REQUIRE(off <= seg_stride || world == 1, "vf_net_adam_fused_gathered: segments of %lld floats overlap", (long long)seg_stride);
    } else {
      L.U = l.fa_u;
      L.V = l.fa_v;
    }
    L.x = x; L.m = m + l.w_off; L.v = v + l.w_off; L.g_out = keep_grad ? g : nullptr;
    L.K = world * l.fa_k; L.Nu = Nu; L.Ncols = 16 * Cv;
    L.kps = l.fa_k; L.seg = all ? seg_stride : 0; L.gscale = 1.f / (float)world;
    L.row0 = 0; L.ldu = 0;
    if (row_world > 1) {
      int r0, nr;
      fused_row_block(Nu, row_rank, row_world, &r0, &nr);
      VF_REQUIRE(nr >= 64, "vf_net_adam_fused_gathered_rows: %d rows do not split over %d ranks (rank %d would hold %d; at least 64)", Nu,
                 row_world, row_rank, nr);
      L.ldu = Nu;
      L.Nu = nr;
      L.row0 = r0;
    }
    l.fa_k = 0;
  }
  return nf ? vf_internal_adam_fused_multi(n->ctx, F, nf, beta1, beta2, eps, t_dev) : 0;
}
VF_API int vf_net_adam_fused(vf_net* n, float* m, float* v, double beta1, double beta2, double eps, const int32_t* t_dev, int keep_grad) {
  VF_REQUIRE(n && m && v && t_dev, "vf_net_adam_fused: NULL argument");
  return fused_layers_launch(n, nullptr, 1, 0, m, v, beta1, beta2, eps, t_dev, keep_grad);
}
// Data parallel: instead of all-reducing the two weight gradients (262 MB of train.lua's 284), every rank all-gathers the OPERANDS
// they are the product of — K = batch rows of (Nu + Ncols) floats per layer, 6 MB at batchSize 64 — and forms the gradient of the
// global batch itself, inside the fused kernel (K = world * batch rows, gscale = 1 / world: the mean over ranks, identical on every
// rank).  vf_net_fused_adam_pack copies this rank's operands, layer after layer (U then V, each padded to 4 floats), into its
// segment of the gather buffer; vf_net_adam_fused_gathered consumes `world` such segments, seg_stride floats apart.
VF_API int vf_net_fused_adam_pack_size(const vf_net* n, int64_t* floats) {
  VF_REQUIRE(n && floats, "vf_net_fused_adam_pack_size: NULL argument");
  int64_t t = 0;
  for (int i : n->fa_layers) {
    const Layer& l = n->L[i];
    const bool full = is_full(l);
    const int Nu = full ? l.C : l.Co, Cv = full ? l.Co : l.C;
    t += pad4((int64_t)n->B * Nu) + pad4((int64_t)n->B * 16 * Cv);
  }
  *floats = t;
  return 0;
}
VF_API int vf_net_fused_adam_pack(vf_net* n, float* seg) {
  VF_REQUIRE(n && seg, "vf_net_fused_adam_pack: NULL argument");
  for (int i : n->fa_layers) {
    const Layer& l = n->L[i];
    const bool full = is_full(l);
    const int Nu = full ? l.C : l.Co, Cv = full ? l.Co : l.C;
    VF_REQUIRE(l.fa_k == n->B, "vf_net_fused_adam_pack: layer %d has no pending gradient of the net's batch (a backward pass with "
               "vf_net_set_fused_adam on comes first)", i);
    VF_CHECK_HIP(hipMemcpyAsync(seg, l.fa_u, (size_t)l.fa_k * Nu * sizeof(float), hipMemcpyDeviceToDevice, n->ctx->stream));
    seg += pad4((int64_t)l.fa_k * Nu);
    VF_CHECK_HIP(hipMemcpyAsync(seg, l.fa_v, (size_t)l.fa_k * 16 * Cv * sizeof(float), hipMemcpyDeviceToDevice, n->ctx->stream));
    seg += pad4((int64_t)l.fa_k * 16 * Cv);
  }
  return 0;
}
VF_API int vf_net_adam_fused_gathered(vf_net* n, const float* all, int world, int64_t seg_stride, float* m, float* v, double beta1,
                                      double beta2, double eps, const int32_t* t_dev, int keep_grad) {
  VF_REQUIRE(n && all && m && v && t_dev && world >= 1, "vf_net_adam_fused_gathered: bad argument");
  return fused_layers_launch(n, all, world, seg_stride, m, v, beta1, beta2, eps, t_dev, keep_grad);
}
// The same with the update SHARDED BY WEIGHT ROWS: rank row_rank of row_world forms the global-batch gradient of ITS
// 1 / row_world of every fused tensor's rows and applies optim.adam to them — 2 B flops per weight and 24 B / row_world of Adam traffic
// per rank again, instead of every rank redoing all rows — then the ranks all-gather the updated rows (the host's collective over
// vf_net_fused_adam_range's slice: row blocks are contiguous and equal).  m and v are touched in this rank's rows only.
VF_API int vf_net_fused_adam_rows_ok(const vf_net* n, int row_world) {
  if (!n || row_world < 1) return 0;
  int cnt = 0;
  for (const Layer& l : n->L) {       // (the layers vf_net_set_fused_adam would mark, whether or not they are marked right now)
    if (!fused_adam_shape(l)) continue;
    const int Nu = is_full(l) ? l.C : l.Co;
    int r0, nr;
    fused_row_block(Nu, row_world - 1, row_world, &r0, &nr);      // (the last rank's block is the shortest)
    if (nr < 64) return 0;
    ++cnt;
  }
  return cnt > 0;
}
// rank `row_rank`'s row block of fused layer i as an element range of the flat vectors (what the host all-gathers after the update)
VF_API int vf_net_fused_adam_row_range(const vf_net* n, int i, int row_rank, int row_world, int64_t* offset, int64_t* length) {
  VF_REQUIRE(n && offset && length && i >= 0 && i < (int)n->fa_layers.size() && row_world >= 1 && row_rank >= 0 && row_rank < row_world,
             "vf_net_fused_adam_row_range: bad argument");
  const Layer& l = n->L[n->fa_layers[i]];
  const bool full = is_full(l);
  const int Nu = full ? l.C : l.Co, Cv = full ? l.Co : l.C;
  int r0, nr;
  fused_row_block(Nu, row_rank, row_world, &r0, &nr);
  *offset = l.w_off + (int64_t)r0 * 16 * Cv;
  *length = (int64_t)nr * 16 * Cv;
  return 0;
}
// The rows another rank updated arrive by a collective the host issued on the communicator's stream (vf_comm_allgather_async /
// vf_comm_broadcast_async tickets).  Nothing in front of the bottleneck conv reads them (train.lua:89-104: E1 ... E5), so the next
// vf_net_forward runs up to there beside the transfer and waits for the tickets in front of the first layer the fused update takes.
VF_API int vf_net_forward_wait_fused(vf_net* n, vf_comm* comm, int ticket) {
  VF_REQUIRE(n && comm && ticket >= 0, "vf_net_forward_wait_fused: bad argument");
  VF_REQUIRE(n->fwd_wait_comm == nullptr || n->fwd_wait_comm == comm, "vf_net_forward_wait_fused: tickets of two communicators");
  n->fwd_wait_comm = comm;
  n->fwd_wait_tickets.push_back(ticket);
  return 0;
}
VF_API int vf_net_adam_fused_gathered_rows(vf_net* n, const float* all, int world, int64_t seg_stride, float* m, float* v, double beta1,
                                           double beta2, double eps, const int32_t* t_dev, int keep_grad, int row_rank, int row_world) {
  VF_REQUIRE(n && all && m && v && t_dev && world >= 1 && row_world >= 1 && row_rank >= 0 && row_rank < row_world,
             "vf_net_adam_fused_gathered_rows: bad argument");
  return fused_layers_launch(n, all, world, seg_stride, m, v, beta1, beta2, eps, t_dev, keep_grad, row_rank, row_world);
}
// net:updateGradInput(input, gradOutput): gradInput only (parameter gradients untouched; train.lua:366)
VF_API int vf_net_update_grad_input(vf_net* n, const float* x, const float* gy, const float** gx) {
  VF_REQUIRE(n && x && gy, "vf_net_update_grad_input: NULL argument");
  return net_walk_back(n, x, gy, gx, false, true, -1, 0, -1, 1);
}
// the same over batch group g of the G the last forward ran concatenated: x / gy hold that group's samples only; saved
// activations and BatchNorm statistics are that group's (fGx's pass over the fake half of netD's 2B batch)
VF_API int vf_net_update_grad_input_group(vf_net* n, const float* x, const float* gy, int g, int G, const float** gx) {
  VF_REQUIRE(n && x && gy, "vf_net_update_grad_input_group: NULL argument");
  return net_walk_back(n, x, gy, gx, false, true, -1, 0, g, G);
}
