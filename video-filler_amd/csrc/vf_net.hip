// vf_net.hip — nn.Sequential behind the C-ABI (graph-level net_{create, forward, backward, update_grad_input,
// parameters}).  A host that does not want to mirror the module protocol (video-filler_amd/nn.py does) can hand the library a
// flat list of layers — exactly what the reference builds with netG:add(...) / netD:add(...) (train.lua:87-199) — and drive the
// whole net with one call per Torch7 method:
//     net:forward(input)                      -> vf_net_forward
//     net:backward(input, gradOutput)         -> vf_net_backward           (updateGradInput + accGradParameters, scale 1)
//     net:updateGradInput(input, gradOutput)  -> vf_net_update_grad_input  (train.lua:366: netD in fGx)
//     net:getParameters()                     -> vf_net_parameters         (ONE flat fp32 buffer + one for the gradients)
//     net:zeroGradParameters(), :training(), :evaluate()
// Pure host code over the library's own entry points (vf_conv2d_*, vf_deconv2d_*, vf_bn_*, vf_act_*): the same kernels, the same
// in-place activation semantics (an activation that follows a convolution or a BatchNorm is applied in its producer's
// epilogue and undone in its backward, as nn.Sequential(fuse=True) does), every weight gradient of a backward call in one
// grouped launch.  What it does NOT carry is the mirror's cross-layer plumbing (BatchNorm statistics out of the GEMM epilogues,
// planes handed from producer to consumer): this is the simple protocol surface, nn.py the fast one.
// Layout: activations NHWC, weights channels-last as everywhere in this library (include/vf_hip.h); the flat parameter buffer
// holds, module by module, {weight, bias} ({gamma, beta} for BatchNorm), every segment padded to 64 floats.
#include <cstring>
#include <vector>

#include "vf_common.h"

// (vf_common.h brings include/vf_hip.h: the entry points used below, vf_layer_desc and the VF_L_* kinds)

namespace {

struct Layer {
  vf_layer_desc d;
  int C, H, W;            // input shape of the layer
  int Co, Ho, Wo;         // output shape
  int64_t w_off = -1, b_off = -1, w_n = 0, b_n = 0;      // offsets into the flat buffers
  float* y = nullptr;     // output [B][Ho][Wo][Co] (NULL: in place on the producer's output / a view)
  float* gx = nullptr;    // gradInput [B][H][W][C]
  // BatchNorm state
  float *rm = nullptr, *rv = nullptr, *sm = nullptr, *si = nullptr;
  double* sums = nullptr;
  int fused_act = VF_ACT_NONE;      // activation applied in this layer's epilogue (the next layer is a VF_L_ACT)
  float fused_slope = 0.f;
  bool absorbed = false;            // a VF_L_ACT that its producer applies
  bool fresh = true;                // zeroGradParameters() was called: the next accGradParameters overwrites (beta = 0)
};

}  // namespace

struct vf_net {
  vf_ctx* ctx = nullptr;
  int B = 0;
  std::vector<Layer> L;
  float *params = nullptr, *grads = nullptr;
  int64_t nparams = 0;
  bool train = true;
  std::vector<void*> owned;
};

static int net_alloc(vf_net* n, void** out, size_t bytes) {
  void* p = nullptr;
  VF_CHECK_HIP(hipMalloc(&p, bytes ? bytes : 4));
  n->owned.push_back(p);
  *out = p;
  return 0;
}

VF_API int vf_net_destroy(vf_net* n) {
  if (!n) return 0;
  for (void* p : n->owned) (void)hipFree(p);
  delete n;
  return 0;
}

// layers[0..nlayers): the modules in order; the input is [B][H][W][C] (NHWC).  Buffers are sized once, here.
VF_API int vf_net_create(vf_ctx* ctx, vf_net** out, const vf_layer_desc* layers, int nlayers, int B, int C, int H, int W) {
  VF_REQUIRE(ctx && out && layers && nlayers > 0 && B > 0 && C > 0 && H > 0 && W > 0, "vf_net_create: bad arguments");
  vf_net* n = new vf_net();
  n->ctx = ctx;
  n->B = B;
  int c = C, h = H, w = W;
  int64_t off = 0;
  auto seg = [&](int64_t len) {
    const int64_t o = off;
    off += (len + 63) & ~(int64_t)63;
    return o;
  };
  for (int i = 0; i < nlayers; ++i) {
    Layer l;
    l.d = layers[i];
    l.C = c; l.H = h; l.W = w;
    l.Co = c; l.Ho = h; l.Wo = w;
    switch (l.d.kind) {
      case VF_L_CONV:
        if (l.d.nin != c) { vf_set_error("vf_net_create: layer %d expects %d input planes, gets %d", i, l.d.nin, c); vf_net_destroy(n); return 2; }
        l.Co = l.d.nout;
        l.Ho = (h + 2 * l.d.pad - l.d.k) / l.d.stride + 1;
        l.Wo = (w + 2 * l.d.pad - l.d.k) / l.d.stride + 1;
        l.w_n = (int64_t)l.d.nout * l.d.k * l.d.k * l.d.nin;
        l.b_n = l.d.nout;
        break;
      case VF_L_FULLCONV:
        if (l.d.nin != c) { vf_set_error("vf_net_create: layer %d expects %d input planes, gets %d", i, l.d.nin, c); vf_net_destroy(n); return 2; }
        l.Co = l.d.nout;
        l.Ho = (h - 1) * l.d.stride - 2 * l.d.pad + l.d.k;
        l.Wo = (w - 1) * l.d.stride - 2 * l.d.pad + l.d.k;
        l.w_n = (int64_t)l.d.nin * l.d.k * l.d.k * l.d.nout;
        l.b_n = l.d.nout;
        break;
      case VF_L_BN:
        if (l.d.nout != c) { vf_set_error("vf_net_create: BatchNorm layer %d has %d channels, gets %d", i, l.d.nout, c); vf_net_destroy(n); return 2; }
        if (l.d.eps == 0.f) l.d.eps = 1e-5f;
        if (l.d.momentum == 0.f) l.d.momentum = 0.1f;
        l.w_n = c;
        l.b_n = c;
        break;
      case VF_L_ACT:
      case VF_L_VIEW:
        break;
      default:
        vf_set_error("vf_net_create: unknown layer kind %d at %d", l.d.kind, i);
        vf_net_destroy(n);
        return 2;
    }
    if (l.Ho <= 0 || l.Wo <= 0) { vf_set_error("vf_net_create: layer %d has an empty output", i); vf_net_destroy(n); return 2; }
    if (l.w_n) { l.w_off = seg(l.w_n); l.b_off = seg(l.b_n); }
    c = l.Co; h = l.Ho; w = l.Wo;
    n->L.push_back(l);
  }
  // an activation directly behind a conv / full-conv / BatchNorm is applied by that layer (in place, as the reference's
  // nn.LeakyReLU(0.2, true) / nn.ReLU(true) are; Tanh / Sigmoid own no state either)
  for (size_t i = 0; i + 1 < n->L.size(); ++i) {
    Layer& p = n->L[i];
    Layer& a = n->L[i + 1];
    if (a.d.kind == VF_L_ACT && (p.d.kind == VF_L_CONV || p.d.kind == VF_L_FULLCONV || p.d.kind == VF_L_BN)) {
      p.fused_act = a.d.act;
      p.fused_slope = a.d.slope;
      a.absorbed = true;
    }
  }
  n->nparams = off;
  int rc = 0;
  if ((rc = net_alloc(n, (void**)&n->params, sizeof(float) * (size_t)off)) || (rc = net_alloc(n, (void**)&n->grads, sizeof(float) * (size_t)off))) {
    vf_net_destroy(n);
    return rc;
  }
  (void)hipMemsetAsync(n->params, 0, sizeof(float) * (size_t)off, ctx->stream);
  (void)hipMemsetAsync(n->grads, 0, sizeof(float) * (size_t)off, ctx->stream);
  for (Layer& l : n->L) {
    const size_t in_e = (size_t)B * l.H * l.W * l.C, out_e = (size_t)B * l.Ho * l.Wo * l.Co;
    const bool own_out = l.d.kind == VF_L_CONV || l.d.kind == VF_L_FULLCONV || l.d.kind == VF_L_BN || (l.d.kind == VF_L_ACT && !l.absorbed);
    if (own_out && (rc = net_alloc(n, (void**)&l.y, sizeof(float) * out_e))) break;
    // gradInput: every computing module; an absorbed activation behind a (full-)convolution lends its slot to that
    // convolution's backward (the activation's own updateGradInput, written there: the caller's gradOutput is not ours to edit)
    const size_t li = (size_t)(&l - &n->L[0]);
    const bool lend = l.d.kind == VF_L_ACT && l.absorbed && li > 0 && n->L[li - 1].d.kind != VF_L_BN;
    if (((l.d.kind != VF_L_VIEW && !(l.d.kind == VF_L_ACT && l.absorbed)) || lend) && (rc = net_alloc(n, (void**)&l.gx, sizeof(float) * in_e))) break;
    if (l.d.kind == VF_L_BN) {
      if ((rc = net_alloc(n, (void**)&l.rm, 4 * (size_t)l.C)) || (rc = net_alloc(n, (void**)&l.rv, 4 * (size_t)l.C)) ||
          (rc = net_alloc(n, (void**)&l.sm, 4 * (size_t)l.C)) || (rc = net_alloc(n, (void**)&l.si, 4 * (size_t)l.C)) ||
          (rc = net_alloc(n, (void**)&l.sums, 16 * (size_t)l.C)))
        break;
      (void)hipMemsetAsync(l.rm, 0, 4 * (size_t)l.C, ctx->stream);
      std::vector<float> ones((size_t)l.C, 1.f);
      (void)hipMemcpyAsync(l.rv, ones.data(), 4 * (size_t)l.C, hipMemcpyHostToDevice, ctx->stream);
      (void)hipStreamSynchronize(ctx->stream);      // (`ones` leaves scope)
    }
  }
  if (rc) {
    vf_net_destroy(n);
    return rc;
  }
  *out = n;
  return 0;
}

// net:getParameters(): the flat buffers (device pointers) and their length in floats.  Segment i of module m starts at
// vf_net_param_offset(net, m, which) (which: 0 = weight / gamma, 1 = bias / beta); the host fills them (weights_init, a checkpoint).
VF_API int vf_net_parameters(vf_net* n, float** params, float** grads, int64_t* count) {
  VF_REQUIRE(n != nullptr, "vf_net_parameters: NULL net");
  if (params) *params = n->params;
  if (grads) *grads = n->grads;
  if (count) *count = n->nparams;
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
VF_API int vf_net_training(vf_net* n, int train) {
  VF_REQUIRE(n != nullptr, "vf_net_training: NULL net");
  n->train = train != 0;
  return 0;
}
VF_API int vf_net_zero_grad(vf_net* n) {      // lazily, like the mirror: the next accGradParameters of each module overwrites
  VF_REQUIRE(n != nullptr, "vf_net_zero_grad: NULL net");
  for (Layer& l : n->L) l.fresh = true;
  return 0;
}
// output of module `layer` after a forward call (net.modules[i].output), device pointer
VF_API int vf_net_layer_output(vf_net* n, int layer, const float** y) {
  VF_REQUIRE(n && y && layer >= 0 && layer < (int)n->L.size(), "vf_net_layer_output: bad layer");
  const float* p = nullptr;
  for (int i = 0; i <= layer; ++i)
    if (n->L[i].y) p = n->L[i].y;
  *y = p;
  return 0;
}

VF_API int vf_net_forward(vf_net* n, const float* x, const float** y) {
  VF_REQUIRE(n && x, "vf_net_forward: NULL argument");
  vf_ctx* ctx = n->ctx;
  const float* cur = x;
  for (Layer& l : n->L) {
    const float* w = l.w_off >= 0 ? n->params + l.w_off : nullptr;
    const float* b = l.b_off >= 0 ? n->params + l.b_off : nullptr;
    int rc = 0;
    switch (l.d.kind) {
      case VF_L_CONV:
        rc = vf_conv2d_fwd(ctx, cur, w, b, l.y, n->B, l.H, l.W, l.C, l.Co, l.d.k, l.d.stride, l.d.pad, l.fused_act, l.fused_slope);
        cur = l.y;
        break;
      case VF_L_FULLCONV:
        rc = vf_deconv2d_fwd(ctx, cur, w, b, l.y, n->B, l.H, l.W, l.C, l.Co, l.d.k, l.d.stride, l.d.pad, l.fused_act, l.fused_slope);
        cur = l.y;
        break;
      case VF_L_BN: {
        const int64_t npix = (int64_t)n->B * l.H * l.W;
        if (n->train)
          rc = vf_bn_train_fwd(ctx, cur, l.y, w, b, l.rm, l.rv, l.sm, l.si, l.sums, npix, l.C, l.d.momentum, l.d.eps, l.fused_act, l.fused_slope);
        else
          rc = vf_bn_eval_fwd(ctx, cur, l.y, w, b, l.rm, l.rv, npix, l.C, l.d.eps, l.fused_act, l.fused_slope);
        cur = l.y;
        break;
      }
      case VF_L_ACT:
        if (!l.absorbed) {
          rc = vf_act_fwd(ctx, cur, l.y, (int64_t)n->B * l.H * l.W * l.C, l.d.act, l.d.slope);
          cur = l.y;
        }
        break;
      default:
        break;      // nn.View: same storage
    }
    if (rc) return rc;
  }
  if (y) *y = cur;
  return 0;
}

// the walk shared by backward (acc = true) and updateGradInput (acc = false)
static int net_walk_back(vf_net* n, const float* x, const float* gy, const float** gx_out, bool acc) {
  vf_ctx* ctx = n->ctx;
  const float* g = gy;
  int rc = 0;
  if (acc && (rc = vf_wgrad_group_begin(ctx))) return rc;
  for (int i = (int)n->L.size() - 1; i >= 0 && !rc; --i) {
    Layer& l = n->L[i];
    const float* in = x;                    // input of layer i = output of the nearest earlier layer that owns one
    for (int j = i - 1; j >= 0; --j)
      if (n->L[j].y) { in = n->L[j].y; break; }
    const float* w = l.w_off >= 0 ? n->params + l.w_off : nullptr;
    float* gw = l.w_off >= 0 ? n->grads + l.w_off : nullptr;
    float* gb = l.b_off >= 0 ? n->grads + l.b_off : nullptr;
    const float beta = l.fresh ? 0.f : 1.f;
    const int64_t out_e = (int64_t)n->B * l.Ho * l.Wo * l.Co;
    switch (l.d.kind) {
      case VF_L_CONV:
      case VF_L_FULLCONV: {
        const bool full = l.d.kind == VF_L_FULLCONV;
        const float* go = g;
        if (l.fused_act != VF_ACT_NONE) {      // undo the activation applied in this layer's epilogue (needs its own buffer:
          float* tmp = n->L[i + 1].gx;         // the caller's gradOutput is not ours to overwrite) — the absorbed module's slot
          if ((rc = vf_act_bwd(ctx, l.y, g, tmp, out_e, l.fused_act, l.fused_slope))) break;
          go = tmp;
        }
        if (acc) {
          rc = full ? vf_deconv2d_bwd_weight(ctx, in, go, gw, gb, n->B, l.H, l.W, l.C, l.Co, l.d.k, l.d.stride, l.d.pad, beta)
                    : vf_conv2d_bwd_weight(ctx, in, go, gw, gb, n->B, l.H, l.W, l.C, l.Co, l.d.k, l.d.stride, l.d.pad, beta);
          l.fresh = false;
          if (rc) break;
        }
        rc = full ? vf_deconv2d_bwd_data(ctx, go, w, l.gx, n->B, l.H, l.W, l.C, l.Co, l.d.k, l.d.stride, l.d.pad)
                  : vf_conv2d_bwd_data(ctx, go, w, l.gx, n->B, l.H, l.W, l.C, l.Co, l.d.k, l.d.stride, l.d.pad);
        g = l.gx;
        break;
      }
      case VF_L_BN: {
        const int64_t npix = (int64_t)n->B * l.H * l.W;
        if (!n->train) { vf_set_error("vf_net_backward: BatchNorm backward in evaluate mode is not part of the path"); rc = 2; break; }
        rc = vf_bn_bwd(ctx, in, l.fused_act != VF_ACT_NONE ? l.y : nullptr, g, l.gx, acc ? gw : nullptr, acc ? gb : nullptr, w, l.sm, l.si,
                       l.sums, npix, l.C, l.fused_act, l.fused_slope, beta);
        if (acc) l.fresh = false;
        g = l.gx;
        break;
      }
      case VF_L_ACT:
        if (!l.absorbed) {
          rc = vf_act_bwd(ctx, l.y, g, l.gx, out_e, l.d.act, l.d.slope);
          g = l.gx;
        }
        break;
      default:
        break;
    }
  }
  if (acc) {
    if (rc) {
      (void)vf_wgrad_group_abort(ctx);
      return rc;
    }
    if ((rc = vf_wgrad_group_end(ctx))) return rc;
  }
  if (rc) return rc;
  if (gx_out) *gx_out = g;
  return 0;
}

// net:backward(input, gradOutput): gradInput + accumulated parameter gradients (every weight gradient in one grouped launch)
VF_API int vf_net_backward(vf_net* n, const float* x, const float* gy, const float** gx) {
  VF_REQUIRE(n && x && gy, "vf_net_backward: NULL argument");
  return net_walk_back(n, x, gy, gx, true);
}
// net:updateGradInput(input, gradOutput): gradInput only (parameter gradients untouched; train.lua:366)
VF_API int vf_net_update_grad_input(vf_net* n, const float* x, const float* gy, const float** gx) {
  VF_REQUIRE(n && x && gy, "vf_net_update_grad_input: NULL argument");
  return net_walk_back(n, x, gy, gx, false);
}
