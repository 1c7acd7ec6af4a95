/* vf_hip.h — C ABI of the MI355X (gfx950) backend for the context-encoder GAN hot path.
 *
 * The reference (MKimiSH/video-filler, Lua/Torch7) has no FFI of its own: its compute boundary is
 * Torch7's nn.Module / nn.Criterion object protocol plus ONE backend-swap hook, util.cudnn(net)
 * (util.lua:108-131, called at train.lua:245-258 / train_vid_weighted.lua:330-347).  Each entry point
 * below is what a Torch7-side module class bound through LuaJIT ffi.cdef would call in place of the
 * THNN / THCUNN / cudnn native it replaces; the replaced native and the reference call site are cited
 * per function.  INTEGRATION.md shows the ffi.cdef + nn.Module stubs.
 *
 * Conventions
 *  - every function returns 0 on success, non-zero on error; vf_last_error() gives the message
 *    (Torch7 natives raise THError -> Lua error(); the binding turns non-zero into error()).
 *  - all tensor pointers are DEVICE pointers to fp32 unless stated; nothing is allocated or freed
 *    by the library except through vf_malloc / vf_free; the caller owns every buffer.
 *  - all work is enqueued on the context's stream; no call synchronises the host except
 *    vf_stream_synchronize and vf_memcpy_d2h.
 *  - activations are NHWC ("channels-last": logical B x C x H x W as the reference indexes it,
 *    physical [B][H][W][C]).  Convolution weights are the reference tensors in channels-last too:
 *      nn.SpatialConvolution      logical [Cout][Cin][kH][kW]  physical [Cout][kH][kW][Cin]
 *      nn.SpatialFullConvolution  logical [Cin][Cout][kH][kW]  physical [Cin][kH][kW][Cout]
 *    vf_nchw_to_nhwc / vf_nhwc_to_nchw convert reference-layout buffers at the boundary.
 *  - convolutions: any square kernel / stride / padding is accepted.  The matrix-core kernels serve k = 4 with
 *    (stride, pad) in {(2,1), (1,0)} on power-of-two maps — the shapes of the reference's main nets
 *    (train.lua:89-146,183-196; fineSize 64 / 128 / 256) — everything else runs on the general kernels of
 *    vf_conv_generic.hip behind the same entry points (see "Shapes" below; vf_conv_is_fast tells which).
 */
#ifndef VF_HIP_H
#define VF_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct vf_ctx vf_ctx;

/* activation codes for fused epilogues and vf_act_* */
enum { VF_ACT_NONE = 0, VF_ACT_LRELU = 1, VF_ACT_RELU = 2, VF_ACT_TANH = 3, VF_ACT_SIGMOID = 4 };

/* ---- context / memory / errors ------------------------------------------------------------ */
const char* vf_last_error(void);
int vf_version(void);
/* cutorch.setDevice(opt.gpu) (train.lua:249).  stream = hipStream_t or NULL for the null stream. */
int vf_ctx_create(vf_ctx** out, int device, void* stream);
int vf_ctx_destroy(vf_ctx* ctx);
/* How the conv / full-conv passes of this context form their products (inputs, outputs and accumulators are fp32 in
 * every mode; activations, weights, BatchNorm, criteria and Adam stay fp32 in HBM):
 *   3 (default): fp32-grade on the bf16 matrix pipe — every fp32 operand element is split EXACTLY into three bf16 planes
 *      (x = hi + mid + lo: 3 x 8 bits = the 24-bit significand) on its way into LDS and the six largest cross terms run
 *      on v_mfma_f32_32x32x16_bf16 with fp32 accumulation; the three dropped terms are below 2^-24 of a product, i.e.
 *      one fp32 rounding.  Passes the fp32 parity tolerances unchanged (tests/test_gpu_bf16.py and the whole -m gpu
 *      suite); 1.1-1.4x faster than mode 0 because the bf16 pipe is 16x the f32 one.
 *      Outside the normal range mode 3 is NOT mode 0 (tests/test_gpu_bf16.py covers normals 1e-18 .. 1e18 only):
 *        - a +-inf operand element gives NaN in every output it reaches (the split forms inf - inf), where modes 0 / 1 give
 *          +-inf; NaN operands give NaN in all modes;
 *        - operand elements below ~2^-110 in magnitude lose their low planes (the residuals are bf16 subnormals, which the
 *          matrix pipe flushes): such an element is carried with 8-16 significant bits instead of 24, and fp32 subnormal
 *          operands are treated as zero.  Products that small are far below fp32's own rounding of any sum they join.
 *      Training never produces either (activations and weights of the reference nets are O(1e-3 .. 1e2)); a host that can
 *      feed infinities and needs IEEE behaviour for them selects mode 0.
 *   0: native fp32 operands, v_mfma_f32_32x32x2_f32 (an fmaf chain, bit for bit).
 *   1: operands ROUNDED to bf16 (round-to-nearest-even), one product term — opt-in, ~2^-9 relative operand rounding. */
int vf_ctx_set_mfma_mode(vf_ctx* ctx, int mode);
int vf_ctx_set_stream(vf_ctx* ctx, void* stream);
/* scratch for split-K slabs and reduction partials; caller-owned, >= vf_workspace_bytes_hint(). */
int vf_ctx_set_workspace(vf_ctx* ctx, void* ptr, size_t bytes);
size_t vf_workspace_bytes_hint(void);
int vf_stream_synchronize(vf_ctx* ctx);
int vf_malloc(void** out, size_t bytes);
int vf_free(void* ptr);
int vf_memcpy_h2d(vf_ctx* ctx, void* dst, const void* src, size_t bytes);
int vf_memcpy_d2h(vf_ctx* ctx, void* dst, const void* src, size_t bytes); /* synchronises */
int vf_zero(vf_ctx* ctx, void* ptr, size_t bytes);                       /* Tensor:zero() */
/* bias zeroing sweep `m.bias:zero()` over every *Convolution* module (train.lua:279-280):
 * zero nseg segments base[offs[i] .. offs[i]+lens[i]) ; offs/lens are DEVICE int64 arrays. */
int vf_zero_segments(vf_ctx* ctx, float* base, const int64_t* offs, const int64_t* lens, int nseg);
int vf_nchw_to_nhwc(vf_ctx* ctx, const float* src, float* dst, int B, int C, int H, int W);
int vf_nhwc_to_nchw(vf_ctx* ctx, const float* src, float* dst, int B, int C, int H, int W);

/* ---- nn.SpatialConvolution (THNN SpatialConvolutionMM / cudnn.SpatialConvolution) ----------
 * Shapes.  The matrix-core kernels (vf_conv.hip, vf_pgemm.hip) serve the reference's main nets: kernel 4x4 with
 * stride 2 pad 1 on maps whose H and W are POWERS OF TWO (pixel coordinates are split with shifts and masks), and the
 * 4x4 stride 1 pad 0 bottleneck pair on a 4x4 / 1x1 map (train.lua:89-146,183-199).  Every other geometry — the option
 * branches' 5x5 / 1x1 convolutions, or 4x4 stride 2 on a map that is not a power of two — is served by the general
 * kernels of vf_conv_generic.hip through the same entry points: same results to the fp32 tolerances, native fp32
 * arithmetic whatever the context's product mode, and several times slower (they are not on the training hot path:
 * fineSize is 64 / 128 / 256 in every recipe of the reference).  vf_conv_is_fast reports which path a geometry takes. */
/* updateOutput.  y[B][Ho][Wo][Cout] = act(conv(x[B][H][W][Cin], w) + bias).  bias may be NULL.
 * act/slope fuse a following in-place nn.LeakyReLU / nn.ReLU / nn.Tanh / nn.Sigmoid (train.lua:90,196). */
int vf_conv_is_fast(int H, int W, int k, int stride, int pad);
int vf_conv2d_fwd(vf_ctx* ctx, const float* x, const float* w, const float* bias, float* y, int B, int H,
                  int W, int Cin, int Cout, int k, int stride, int pad, int act, float slope);
/* updateGradInput.  gx[B][H][W][Cin] from gy[B][Ho][Wo][Cout]. */
int vf_conv2d_bwd_data(vf_ctx* ctx, const float* gy, const float* w, float* gx, int B, int H, int W,
                       int Cin, int Cout, int k, int stride, int pad);
/* accGradParameters(scale = 1).  gw = beta*gw + dW ; gb = beta*gb + db (gb may be NULL).
 * beta = 1 is Torch's accumulate; beta = 0 overwrites (saves the gradParameters:zero() pass). */
/* updateGradInput of a conv whose INPUT is the in-place activated output of the module before it (train.lua:89-91:
 * conv -> LeakyReLU(0.2, true) -> conv): gx = (W^T * gy) .* act'(x_act), i.e. this conv's updateGradInput and the
 * nn.LeakyReLU:updateGradInput pass over the same tensor in one epilogue.  x_act is this conv's own input. */
int vf_conv2d_bwd_data_act(vf_ctx* ctx, const float* gy, const float* w, float* gx, const float* x_act, int act,
                           float slope, int B, int H, int W, int Cin, int Cout, int k, int stride, int pad);
int vf_conv2d_bwd_weight(vf_ctx* ctx, const float* x, const float* gy, float* gw, float* gb, int B, int H,
                         int W, int Cin, int Cout, int k, int stride, int pad, float beta);

/* ---- nn.SpatialFullConvolution (THNN SpatialFullConvolution; train.lua:134-146) ------------- */
/* x[B][H][W][Cin] -> y[B][Ho][Wo][Cout], Ho = (H-1)*stride - 2*pad + k. */
int vf_deconv2d_fwd(vf_ctx* ctx, const float* x, const float* w, const float* bias, float* y, int B, int H,
                    int W, int Cin, int Cout, int k, int stride, int pad, int act, float slope);
int vf_deconv2d_bwd_data(vf_ctx* ctx, const float* gy, const float* w, float* gx, int B, int H, int W,
                         int Cin, int Cout, int k, int stride, int pad);
int vf_deconv2d_bwd_weight(vf_ctx* ctx, const float* x, const float* gy, float* gw, float* gb, int B, int H,
                           int W, int Cin, int Cout, int k, int stride, int pad, float beta);

/* ---- the same passes from PRE-SPLIT operands (vf_pgemm.hip) ----------------------------------
 * The default product mode forms every fp32 product from three bf16 planes per operand (x = hi + mid + lo, exact: see
 * vf_ctx_set_mfma_mode).  The entry points above split their fp32 operands inside the GEMM, in every pass that reads
 * them; these take operands split ONCE by whoever produced them — `planes` = bf16 [3][n], plane q at q * n elements:
 *   vf_planes_split    any fp32 tensor of n elements (n % 4 == 0)
 *   vf_weight_planes   a conv / full-conv weight, physical [d0][16][d1]: native planes [3][d0][16][d1] and (optional)
 *                      transposed planes [3][d1][16][d0] — the data-gradient / full-conv forward passes walk the transpose
 *   vf_bn_train_fwd_pre / vf_bn_bwd_pre write the planes of their output beside it (y_planes / gx_planes, may be NULL)
 * and compute bit for bit the six-term products of mode 3 (another summation order over K: same parity tolerances).
 * 4x4, stride 2, pad 1 only, channel counts of the gathered operand % 32 == 0, output channels >= 32 and % 4 == 0, more
 * than 64 GEMM rows (vf_pconv_supported says; everything else stays with the entry points above).
 *   vf_pconv_gather    conv forward (ap = x planes [B][H][W][Cin], wp = native planes of w [Cout][16][Cin]) and full-conv
 *                      data-gradient (ap = gy planes, wp = native planes of the full-conv weight [Cin_full][16][Cout_full])
 *   vf_pconv_scatter   conv data-gradient (ap = gy planes on the LOW-res grid H x W, wp = transposed planes [Cin][16][Cout];
 *                      dmask/dact/dslope as vf_conv2d_bwd_data_act) and full-conv forward (ap = x planes, wp = transposed
 *                      planes of the full-conv weight [Cout_full][16][Cin_full], bias, act); output is 2H x 2W
 * Both honour a pending vf_bn_fuse_next_* attachment like the entry points above. */
int vf_planes_split(vf_ctx* ctx, const float* x, void* planes, int64_t n);
/* vf_conv2d_fwd that also leaves the planes of y (in its epilogue for the 3-channel image-side layers, train.lua:89,183) */
int vf_conv2d_fwd_planes(vf_ctx* ctx, const float* x, const float* w, const float* bias, float* y, void* y_planes, int B,
                         int H, int W, int Cin, int Cout, int k, int stride, int pad, int act, float slope);
int vf_weight_planes(vf_ctx* ctx, const float* w, void* planes_native, void* planes_transposed, int d0, int d1);
/* the same for n weights in one launch: desc_dev = device array of n records {const float* w; void* native; void* transposed;
 * int d0, d1, gx = ceil(d0/32), gz = ceil(d1/32), blk_off (first block of the record: running sum of gx*16*gz), pad;} (48 bytes
 * each); blocks = the total.  A net refreshes its weight planes once per parameter update (optim.adam, train.lua:421-424). */
int vf_weight_planes_multi(vf_ctx* ctx, const void* desc_dev, int n, int blocks);
int vf_pconv_supported(int B, int H, int W, int Cin, int Cout, int k, int stride, int pad, int transposed);
/* The planes path in the bf16-operand mode (vf_ctx_set_mfma_mode 1): `planes` is then ONE plane, bf16 [1][n], the operand rounded
 * to nearest-even by its producer — every producer above (vf_planes_split, vf_weight_planes[_multi], the y_planes / gx_planes
 * of the BatchNorm entries, vf_conv2d_fwd_planes) writes that form when the context is in mode 1 — and the GEMMs issue one MFMA
 * per product instead of six (k_pconv_dma<.., NPL = 1>, k_pwgrad_group<.., NPL = 1>).  Served shapes are narrower (whole
 * 64-channel K steps, whole 64 x 64 tiles): vf_pconv_supported_in_mode(1, ...); mode 3 = vf_pconv_supported; mode 0: never.
 * Results equal the in-kernel-rounding kernels' (same products, another summation order). */
int vf_pconv_supported_in_mode(int mfma_mode, int B, int H, int W, int Cin, int Cout, int k, int stride, int pad, int transposed);
/* Which kernel serves a planes pass is a tiling decision of the library (kernel map).  Two of the choices can be switched
 * per process, for same-box A/B runs and for the tests that compare the kernels bit for bit: gather_patch (0 / 1: k_pconv_patch_g for
 * the gather passes on whole 128-row tiles; default 1, env VF_PG_GPATCH) and scatter_patch (0 off, 1 auto, 2 / 4 parity classes per
 * block of k_pconv_patch_tr; default 1, env VF_PG_PATCH).  A negative value leaves a setting as it is.  Results do not depend on it:
 * every choice forms the same six-term products in the same K order per output element. */
int vf_pconv_set_routing(int gather_patch, int scatter_patch);
int vf_pconv_gather(vf_ctx* ctx, const void* ap, const void* wp, const float* bias, float* y, int B, int H, int W, int Cin,
                    int Cout, int act, float slope);
int vf_pconv_scatter(vf_ctx* ctx, const void* ap, const void* wp, const float* bias, float* y, int B, int H, int W, int Cin,
                     int Cout, int act, float slope, const float* dmask, int dact, float dslope);
/* accGradParameters with the planes of BOTH operands at hand (x_planes: bf16 [3][B*H*W*Cin], gy_planes: bf16 [3][B*Ho*Wo*Cout],
 * the layout vf_planes_split writes; x and gy themselves are still needed: bias gradient, shapes the planes kernel does not
 * take).  The weight gradient runs on k_pwgrad_group — both operands DMA-staged as [pixel][channel] tiles, fragments by the
 * transposing LDS read — when the product mode is 3, stride 2 pad 1, the low-resolution operand has a multiple of 128
 * channels, the gathered one a multiple of 64 and the pixel count is a multiple of 32; otherwise exactly
 * vf_conv2d_bwd_weight / vf_deconv2d_bwd_weight.  Same six-term products, another summation order over pixels. */
int vf_conv2d_bwd_weight_planes(vf_ctx* ctx, const float* x, const float* gy, const void* x_planes, const void* gy_planes, float* gw,
                                float* gb, int B, int H, int W, int Cin, int Cout, int k, int stride, int pad, float beta);
int vf_deconv2d_bwd_weight_planes(vf_ctx* ctx, const float* x, const float* gy, const void* x_planes, const void* gy_planes,
                                  float* gw, float* gb, int B, int H, int W, int Cin, int Cout, int k, int stride, int pad,
                                  float beta);

/* ---- nn.SpatialBatchNormalization (THNN/THCUNN BatchNormalization; train.lua:92) ------------ */
/* Training forward in two phases so a data-parallel caller can all-reduce `sums` in between:
 *   vf_bn_stats     : sums[0..C) = sum(x - shift), sums[C..2C) = sum((x - shift)^2)  (DOUBLE), shift = running_mean
 *   vf_bn_finalize  : mean/invstd from sums over n_total rows; updates running stats (momentum, unbiased var)
 *   vf_bn_apply     : y = act((x - mean) * invstd * gamma + beta)     (y may alias x)
 * vf_bn_train_fwd runs the three back to back.  npix = B*H*W rows of C channels. */
int vf_bn_stats(vf_ctx* ctx, const float* x, const float* shift, double* sums, int64_t npix, int C);
int vf_bn_finalize(vf_ctx* ctx, const double* sums, float* running_mean, float* running_var, float* save_mean,
                   float* save_invstd, int64_t n_total, int C, float momentum, float eps);
int vf_bn_apply(vf_ctx* ctx, const float* x, float* y, const float* gamma, const float* beta, const float* mean,
                const float* invstd, int64_t npix, int C, int act, float slope);
int vf_bn_train_fwd(vf_ctx* ctx, const float* x, float* y, const float* gamma, const float* beta,
                    float* running_mean, float* running_var, float* save_mean, float* save_invstd, double* sums,
                    int64_t npix, int C, float momentum, float eps, int act, float slope);
/* evaluate() mode (test_vid.lua:48): running statistics. */
int vf_bn_eval_fwd(vf_ctx* ctx, const float* x, float* y, const float* gamma, const float* beta,
                   const float* running_mean, const float* running_var, int64_t npix, int C, float eps, int act,
                   float slope);
/* Backward, also two-phase: vf_bn_bwd_stats gives sums[0..C) = sum(g), sums[C..2C) = sum(g*(x-mean)) with
 * g = gy masked by the fused activation's derivative (y_act = the activated output; NULL when act = NONE);
 * vf_bn_bwd_apply writes gx (may be NULL) and ggamma/gbeta (may be NULL) = pbeta*old + new. */
int vf_bn_bwd_stats(vf_ctx* ctx, const float* x, const float* y_act, const float* gy, const float* save_mean,
                    double* sums, int64_t npix, int C, int act, float slope);
int vf_bn_bwd_apply(vf_ctx* ctx, const float* x, const float* y_act, const float* gy, float* gx, float* ggamma,
                    float* gbeta, const float* gamma, const float* save_mean, const float* save_invstd,
                    const double* sums, int64_t npix, int64_t n_total, int C, int act, float slope, float pbeta);
int vf_bn_bwd(vf_ctx* ctx, const float* x, const float* y_act, const float* gy, float* gx, float* ggamma,
              float* gbeta, const float* gamma, const float* save_mean, const float* save_invstd, double* sums,
              int64_t npix, int C, int act, float slope, float pbeta);
/* The same two calls over a batch that is the concatenation of `groups` independent batches of npix_per_group pixels
 * each (netD's real and fake passes of one closure run as one tensor, train.lua:300-349): statistics, running-average
 * updates and backward sums are per group, in group order — exactly what `groups` separate calls would compute, in one
 * launch per stage.  save_mean / save_invstd are [groups][C], sums is [groups][2C]; gamma/beta gradients accumulate the
 * groups' contributions in order (pbeta applies to the first). */
int vf_bn_train_fwd_groups(vf_ctx* ctx, const float* x, float* y, const float* gamma, const float* beta,
                           float* running_mean, float* running_var, float* save_mean, float* save_invstd,
                           double* sums, int64_t npix_per_group, int C, int groups, float momentum, float eps,
                           int act, float slope);
int vf_bn_bwd_groups(vf_ctx* ctx, const float* x, const float* y_act, const float* gy, float* gx, float* ggamma,
                     float* gbeta, const float* gamma, const float* save_mean, const float* save_invstd,
                     double* sums, int64_t npix_per_group, int C, int groups, int act, float slope, float pbeta);

/* BatchNorm statistics as a by-product of the convolution that produces the tensor (train.lua:91-92: every BatchNorm of the
 * nets follows a convolution).  vf_bn_fuse_next_fwd / _bwd attach a request to the context; the NEXT vf_conv2d_fwd /
 * vf_deconv2d_fwd (fwd) or vf_conv2d_bwd_data / vf_deconv2d_bwd_data (bwd) on it leaves per-tile partial sums in `part`
 * ([groups][rows][2][C] doubles, rows <= part_rows_cap / groups) from its epilogue (or from its split-K combine), and
 * vf_bn_fuse_result says how many rows per group it wrote — 0 when that launch could not (thin or generic shapes, tiles
 * that straddle a batch group): the caller then runs the plain vf_bn_train_fwd / vf_bn_bwd.
 *   fwd: sums of (v - shift), (v - shift)^2 of the conv output v, shift = running_mean[C]  -> vf_bn_train_fwd_pre
 *   bwd: the data-gradient pass stores its output ALREADY MASKED by the derivative of the activation fused behind the
 *        BatchNorm (act, y_act = the activated BatchNorm output; nn.LeakyReLU / nn.ReLU:updateGradInput) and sums
 *        g and g * (x - save_mean), x = the BatchNorm's input                                   -> vf_bn_bwd_pre
 * The _pre calls are vf_bn_train_fwd_groups / vf_bn_bwd_groups without their statistics pass (g_masked: no activation);
 * y_planes / gx_planes (may be NULL): the three bf16 planes of the output, [3][groups * npix_per_group * C], written beside
 * it for a vf_pconv_* consumer. */
int vf_bn_train_fwd_planes(vf_ctx* ctx, const float* x, float* y, const float* gamma, const float* beta,
                           float* running_mean, float* running_var, float* save_mean, float* save_invstd, double* sums,
                           int64_t npix_per_group, int C, int groups, float momentum, float eps, int act, float slope,
                           void* y_planes);      /* vf_bn_train_fwd_groups + the planes of y */
int vf_bn_bwd_planes(vf_ctx* ctx, const float* x, const float* y_act, const float* gy, float* gx, float* ggamma,
                     float* gbeta, const float* gamma, const float* save_mean, const float* save_invstd, double* sums,
                     int64_t npix_per_group, int C, int groups, int act, float slope, float pbeta,
                     void* gx_planes);           /* vf_bn_bwd_groups + the planes of gx */
int vf_bn_fuse_next_fwd(vf_ctx* ctx, const float* shift, double* part, int part_rows_cap, int groups);
int vf_bn_fuse_next_bwd(vf_ctx* ctx, const float* x, const float* y_act, int act, float slope, const float* save_mean,
                        double* part, int part_rows_cap, int groups);
int vf_bn_fuse_result(vf_ctx* ctx, int* rows_per_group);
int vf_bn_train_fwd_pre(vf_ctx* ctx, const double* part, int rows_per_group, const float* x, float* y, const float* gamma,
                        const float* beta, float* running_mean, float* running_var, float* save_mean, float* save_invstd,
                        double* sums, int64_t npix_per_group, int C, int groups, float momentum, float eps, int act,
                        float slope, void* y_planes);
int vf_bn_bwd_pre(vf_ctx* ctx, const double* part, int rows_per_group, const float* x, const float* g_masked, float* gx,
                  float* ggamma, float* gbeta, const float* gamma, const float* save_mean, const float* save_invstd,
                  double* sums, int64_t npix_per_group, int C, int groups, float pbeta, void* gx_planes);

/* ---- pointwise modules (nn.LeakyReLU / ReLU / Tanh / Sigmoid; train.lua:90,146,196) --------- */
int vf_act_fwd(vf_ctx* ctx, const float* x, float* y, int64_t n, int act, float slope); /* y may alias x */
/* gx = gy * act'(.) evaluated from the ACTIVATED output y (in-place semantics); gx may alias gy */
int vf_act_bwd(vf_ctx* ctx, const float* y, const float* gy, float* gx, int64_t n, int act, float slope);
/* y = a*x + b*y  (Tensor:mul / :add(alpha, src), train.lua:382) */
int vf_axpby(vf_ctx* ctx, float a, const float* x, float b, float* y, int64_t n);
/* y = y .* x  (Tensor:cmul, train_vid_weighted.lua:497) */
int vf_cmul(vf_ctx* ctx, const float* x, float* y, int64_t n);
/* y = a*y + b   (input_mask:mul(1-lambda):add(lambda), train_vid_weighted.lua:494) */
int vf_scale_shift(vf_ctx* ctx, float* y, float a, float b, int64_t n);
/* out = mask != 0 ? fake : real   (maskedSelect + maskedCopy, train_vid_weighted.lua:430-432) */
int vf_masked_compose(vf_ctx* ctx, float* out, const float* real, const float* fake, const float* mask,
                      int64_t n);

/* ---- criteria: loss scalars are written to DEVICE doubles (no host sync) -------------------- */
/* nn.BCECriterion (eps 1e-12, sizeAverage; train.lua:204).  target = constant label (label:fill). */
int vf_bce_fwd(vf_ctx* ctx, const float* x, float label, int n, double* loss);
int vf_bce_bwd(vf_ctx* ctx, const float* x, float label, float* gx, int n);
/* criterion:forward and criterion:backward of the same scores in ONE launch, for one group of n scores or for two consecutive
 * groups with their own labels (netD's real and fake halves, train.lua:331-349): loss0 / loss1 get the group means, gx the
 * gradient of all groups * n scores.  Element for element vf_bce_fwd + vf_bce_bwd. */
int vf_bce_fwd_bwd(vf_ctx* ctx, const float* x, float label0, float label1, int n_per_group, int groups, double* loss0,
                   double* loss1, float* gx);
/* nn.MSECriterion (train.lua:207).  loss = mean((x-t)^2); gx = (2/n)(x-t). */
int vf_mse_fwd(vf_ctx* ctx, const float* x, const float* t, int64_t n, double* loss);
int vf_mse_bwd(vf_ctx* ctx, const float* x, const float* t, float* gx, int64_t n);
/* Fused generator reconstruction gradient (train_vid_weighted.lua:489-503,523-528 / train.lua:377-399):
 *   loss  = mean((x-t)^2)
 *   df_dg = alpha*df_dg + (2/n)(x-t) * wgt,   wgt = c0 + c1*mask[i]            (mask != NULL)
 *                                             wgt = inside band ? c0 : c0 + c1  (mask == NULL, band > 0:
 *                                                   rows/cols [band, HW-band) of an HW x HW image are "inside")
 * n = B*HW*HW*C elements, NHWC. */
int vf_recon_grad_mix(vf_ctx* ctx, float* df_dg, const float* x, const float* t, const float* mask, float alpha,
                      float c0, float c1, int band, int HW, int C, int64_t n, double* loss);
/* nn.GDLCriterion(1):forward (gdl_criterion.lua:38-45) incl. the flattened-pairing quirk. */
int vf_gdl_fwd(vf_ctx* ctx, const float* yhat, const float* y, int B, int H, int W, int C, double* loss);
/* updateGradInput (gdl_criterion.lua:47-53): gyhat = d loss / d yhat, same shapes.  No driver of the reference calls it
 * (train_vid_weighted.lua:523-528 adds the forward VALUE to the loss and uses the MSE gradient); provided so that
 * nn.GDLCriterion is a complete nn.Criterion.  Derivative convention of THNN Abs / AbsCriterion: +1 at 0. */
int vf_gdl_bwd(vf_ctx* ctx, const float* yhat, const float* y, float* gyhat, int B, int H, int W, int C);
/* nn.MaskedMSECriterion(w) (MaskedMSECriterion.lua:29-41); mask is uint8 0/1. */
int vf_masked_mse_fwd(vf_ctx* ctx, const float* x, const float* xhat, const uint8_t* mask, float w, int64_t n,
                      double* loss);
int vf_masked_mse_bwd(vf_ctx* ctx, const float* x, const float* xhat, const uint8_t* mask, float w, float* gx,
                      int64_t n);

/* ---- optim.adam (train.lua:421-424) -------------------------------------------------------- */
/* One fused pass over the flat parameter vector (28 B/param).  t_dev points to TWO DEVICE int32 words:
 * t_dev[0] = state.t (the call increments it first), t_dev[1] = scratch for the fp32 step size
 * lr*sqrt(1-beta2^t)/(1-beta1^t), computed on the device in double so a captured graph replays correctly.
 * Hyper-parameters are doubles, as Lua numbers are. */
int vf_adam_step(vf_ctx* ctx, float* x, const float* g, float* m, float* v, int64_t n, double lr, double beta1,
                 double beta2, double eps, int32_t* t_dev);
/* The two halves of vf_adam_step, so that one update can be applied range by range (and on more than one stream):
 * vf_adam_prep advances the step count and derives the step size ONCE (t_dev[0] += 1, t_dev[1] = bits of
 * lr*sqrt(1-beta2^t)/(1-beta1^t)); vf_adam_apply updates x[0..n) (any 16-byte aligned sub-range of the flat
 * vectors) with that step size.  prep + apply over the whole vector == vf_adam_step. */
int vf_adam_prep(vf_ctx* ctx, double lr, double beta1, double beta2, int32_t* t_dev);
int vf_adam_apply(vf_ctx* ctx, float* x, const float* g, float* m, float* v, int64_t n, double beta1, double beta2,
                  double eps, const int32_t* t_dev);
/* the same pass over several element ranges of the flat vectors in ONE launch (offsets / lengths in floats, multiples of 4, at most 8
 * ranges; host arrays): the generator's vector around the slices vf_net_adam_fused takes */
int vf_adam_apply_ranges(vf_ctx* ctx, float* x, const float* g, float* m, float* v, const int64_t* offsets, const int64_t* lengths,
                         int nranges, double beta1, double beta2, double eps, const int32_t* t_dev);
/* optim.adam applied inside the weight-gradient kernel of a bottleneck layer (train.lua:104 conv nef*8 -> nBottleneck on a 4x4
 * map, :134 full-conv nBottleneck -> ngf*8 onto one; THNN accGradParameters with K = batch):
 *   g[n][col] = sum_{k < K} U[k][n] * V[k][col]        U = [K][Nu] (the 1x1-map side), V = [K][Ncols] (the 4x4-map side, Ncols = 16*C)
 * is formed in the accumulators and consumed there: x, m, v = [Nu][Ncols] slices of the flat parameter vector and of the optimiser
 * state are read and written once — 24 B per weight where accGradParameters + vf_adam_apply move 32 (those two tensors are 92 % of
 * train.lua's generator).  g (may be NULL) also receives the gradient (gradParameters stays complete; 28 B).  Requires a fresh
 * gradient (zeroGradParameters before the backward pass, as both closures do) and t_dev after vf_adam_prep; element for element
 * the update of vf_adam_apply.  vf_wgrad_adam_outer_supported: 1 when the shape is the kernel's (Ncols % 128 == 0, Nu even, >= 64). */
int vf_wgrad_adam_outer_supported(int K, int Nu, int Ncols);
int vf_wgrad_adam_outer(vf_ctx* ctx, const float* U, const float* V, int K, int Nu, int Ncols, float* x, float* m, float* v,
                        float* g, double beta1, double beta2, double eps, const int32_t* t_dev);
/* the same with the batch rows gathered from several ranks (data parallel): row k lives in segment k / rows_per_seg at row
 * k % rows_per_seg of it, the segments seg_stride floats apart (U and V both); g = gscale * sum (1 / world: the mean over ranks).
 * K % rows_per_seg == 0, seg_stride % 4 == 0; U, V 8-byte aligned. */
int vf_wgrad_adam_outer_gathered(vf_ctx* ctx, const float* U, const float* V, int K, int rows_per_seg, int64_t seg_stride, int Nu,
                                 int Ncols, float* x, float* m, float* v, float* g, float gscale, double beta1, double beta2,
                                 double eps, const int32_t* t_dev);
/* the same for rows [row0, row0 + nrows) of the tensors only (row0 even, nrows even and >= 64; x, m, v, g point at row 0): the
 * data-parallel update sharded by weight rows — bit for bit what the whole-tensor call gives those rows */
int vf_wgrad_adam_outer_rows(vf_ctx* ctx, const float* U, const float* V, int K, int rows_per_seg, int64_t seg_stride, int Nu, int Ncols,
                             int row0, int nrows, float* x, float* m, float* v, float* g, float gscale, double beta1, double beta2,
                             double eps, const int32_t* t_dev);

/* ---- batch preparation and the inference tile loop (the data formats either side of the closures) ---------
 * train.lua:284-298: from the loader's batch (B x C x fs x fs planar, [-1,1]) produce the NHWC generator input with
 * the centre hole [fs/4+ov, 3fs/4-ov)^2 painted with fill[c] (DEVICE float[C]; 2*{117,104,123}/255-1 in the reference)
 * and the NHWC clone of the centre crop [fs/4, 3fs/4)^2 taken BEFORE painting. */
int vf_center_prepare(vf_ctx* ctx, const float* batch_nchw, float* ctx_nhwc, float* center_nhwc, const float* fill,
                      int B, int C, int fs, int overlapPred);
/* datavid/donkey_folder.lua:135-189 (trainHook, withMask) for ONE sample: crop the decoded clip (C x iH x iW planar,
 * [0,1]) at 0-based (w1, h1), build the mask (nblocks == 0: crop of the mask image `mask` (iH x iW), non-zero =
 * masked, :162-166; nblocks > 0: randomBlockMask's squares of side block_size at 1-based HOST coordinates tlx/tly,
 * :114-129), fill masked pixels with mask_value, mirror horizontally if flip (:178-183), map [0,1] -> [-1,1]
 * (:185-187).  Outputs are fs x fs x C NHWC: full clip, masked clip, mask as 0/1 floats. */
int vf_clip_prepare(vf_ctx* ctx, const float* clip, const float* mask, float* full, float* masked, float* maskout,
                    int C, int iH, int iW, int fs, int w1, int h1, int flip, float mask_value, int nblocks,
                    int block_size, const int* tlx, const int* tly);
/* test_vid_wholeim.lua:159-205: the padded planar clip (groups*nc x H x W; H, W multiples of fs) cut into
 * (H/fs)*(W/fs) tiles, all gathered into ONE NHWC batch — row (tile*groups + g) is fs x fs x nc — so the generator
 * runs once instead of once per tile (evaluate-mode BatchNorm makes that exact); vflip (DEVICE uint8 per tile, or
 * NULL) flips a tile vertically on the way in (:167-170) and back on the way out (:191-197). */
int vf_tiles_gather(vf_ctx* ctx, const float* full, float* tiles, int groups, int nc, int H, int W, int fs,
                    const unsigned char* vflip);
int vf_tiles_scatter(vf_ctx* ctx, const float* tiles, float* out, int groups, int nc, int H, int W, int fs,
                     const unsigned char* vflip);

/* ---- option branches of train.lua: noiseGen (:109-124, 319-327) and conditionAdv (:158-180) -----------------------
 * nn.JoinTable(2) over NHWC tensors: dst[p][c_dst + c] = src[p][c_src + c] for c < Ccopy, p < npix (forward: one call
 * per table element into the joined tensor; updateGradInput: one call per element out of the joined gradient).
 * The branches' convolutions (5x5 stride 2 pad 2 / 2+32; 1x1) are served by vf_conv2d_* above: any kernel size,
 * stride and padding is accepted there, the 4x4 shapes of the main nets take the matrix-core path. */
int vf_channel_copy(vf_ctx* ctx, const float* src, int Csrc, int c_src, float* dst, int Cdst, int c_dst, int Ccopy,
                    int64_t npix);
/* noise:uniform(-1,1) (normal = 0) / noise:normal(0,1) (normal = 1), train.lua:319-323.  Counter-based: element i is a
 * function of (seed, counter, i) only (splitmix64; Torch7's generator stream is not reproducible).  counter_dev, if
 * not NULL, is a DEVICE int32 read at run time (e.g. Adam's step count, so a replayed HIP graph draws fresh noise
 * every iteration); otherwise `counter` is used. */
int vf_noise_fill(vf_ctx* ctx, float* out, int64_t n, uint64_t seed, const int32_t* counter_dev, uint64_t counter,
                  int normal);

/* ---- every weight gradient of one backward walk in one launch -----------------------------------------------------
 * Between vf_wgrad_group_begin and vf_wgrad_group_end, vf_conv2d_bwd_weight / vf_deconv2d_bwd_weight calls on this
 * context are RECORDED (the 16-byte-vectorised ones; others still launch at once) and the group — one grouped GEMM
 * launch per tile family plus one grouped split-K reduce — runs at _end.  Valid because accGradParameters of a layer
 * reads only that layer's input and gradOutput, which the nn protocol keeps in the modules' buffers until the walk is
 * over; the caller must not overwrite them, nor read gradWeight, before _end.  The workspace must hold the split-K
 * slabs of all recorded layers at once (vf_workspace_bytes_hint); if it does not, the group is flushed early. */
int vf_wgrad_group_begin(vf_ctx* ctx);
int vf_wgrad_group_end(vf_ctx* ctx);
/* Drop an open group without launching anything (a host-side error cut the backward walk short): the recorded GEMMs are
 * discarded and the context is back to immediate launches.  No-op when no group is open. */
int vf_wgrad_group_abort(vf_ctx* ctx);
/* Data parallel: a walk whose weight gradients leave in two launches, both at its END (a group launched in the middle of the
 * data-gradient chain slowed the passes behind it by 20-40 %: DESIGN.md 8).  vf_wgrad_group_count: gradients recorded so far (the
 * host reads it when the walk passes the bucket boundary); vf_wgrad_group_end_partial(count): launch the first `count` recorded
 * ones — the finished bucket, whose exchange can start — and keep the group open; vf_wgrad_group_end launches the rest. */
int vf_wgrad_group_count(vf_ctx* ctx, int* count);
int vf_wgrad_group_end_partial(vf_ctx* ctx, int count);

/* ---- every conv bias gradient of one backward walk in two launches ----------------------------------------------
 * gradBias = sum over pixels of gradOutput (THNN accGradParameters) is not needed before optim.adam, and each layer's
 * gradOutput stays in its module's buffer until the walk ends, so the column sums of all layers run as ONE stage-1 and
 * ONE stage-2 launch.  desc_dev: DEVICE array of n 64-byte descriptors
 *   { const float* g; float* gb; double* part; int64 P; int32 C, cq, rows_per_block, gx, gy, blk1_off, blk2_off; float beta }
 * (g: [P][C] gradOutput, C % 4 == 0, 16-byte aligned; part: gx*C doubles of scratch; geometry from vf_bias_grad_plan;
 * blk1_off / blk2_off: running sums of gx*gy / ceil(C/4) over the layers).  gb = beta*gb + column sums. */
int vf_bias_grad_plan(int64_t P, int C, int* cq, int* rows_per_block, int* gx, int* gy);
int vf_bias_grad_multi(vf_ctx* ctx, const void* desc_dev, int n, int blocks1, int blocks2);

/* ---- data-parallel exchange over RCCL (one process per GPU) --------------------------------------------------------
 * The reference trains on one device (train.lua:42 `gpu = 1`); BASELINE's N > 1 configuration is data parallelism over
 * the same closures: the flat gradient vectors of train.lua:240-241 are averaged over ranks before each optim.adam,
 * optionally BatchNorm's sums are added over ranks (statistics of the global batch).  These entries are that exchange, so
 * a Lua / FFI host needs nothing besides this library.  RCCL is bound at run time (dlopen "librccl.so", or $VF_RCCL_LIB)
 * at the first call; the rest of the library does not depend on it.
 *
 * vf_comm_available: 0 if RCCL can be bound in this process (local, non-collective: check it on every rank and let the ranks
 *   agree BEFORE entering the collective vf_comm_init, so that a rank without RCCL does not leave the others waiting).
 * vf_comm_unique_id: rank 0 fills 128 opaque bytes; the host hands them to every rank (file, socket, MPI ...).
 * vf_comm_init: on any failure the half-built communicator is released (RCCL comm aborted, stream / events destroyed).
 * vf_comm_init: collective over all ranks, on the CURRENT device (hipSetDevice first).
 * vf_comm_allreduce_async: in place, on the communicator's own stream, ordered after the work already given to ctx's
 *   stream; ctx's stream carries on (backward kernels overlap the bucket's flight).  dtype 0 = f32, 1 = f64;
 *   op 0 = sum, 1 = average, 2 = max, 3 = min.  *ticket (0..63, reused round-robin) names it for vf_comm_wait, which makes
 *   ctx's stream wait on the DEVICE for that collective and all earlier ones (the host does not block).
 * vf_comm_allreduce_avg_async: the gradient bucket case (f32, average).
 * vf_comm_allreduce_inline: the collective on ctx's stream itself (SyncBN sums; capturable into a hipGraph with the
 *   kernels around it).  vf_comm_broadcast: root's buffer to all ranks, on ctx's stream (initial parameters).
 * vf_comm_barrier: host-blocking; both streams of every rank have drained. */
typedef struct vf_comm vf_comm;
#define VF_COMM_ID_BYTES 128
int vf_comm_available(void);
int vf_comm_unique_id(void* id128);
int vf_comm_init(vf_comm** out, const void* id128, int world, int rank);
int vf_comm_world(const vf_comm* c);
int vf_comm_rank(const vf_comm* c);
int vf_comm_allreduce_async(vf_comm* c, vf_ctx* ctx, void* buf, int64_t count, int dtype, int op, int* ticket);
int vf_comm_allreduce_avg_async(vf_comm* c, vf_ctx* ctx, float* buf, int64_t n, int* ticket);
/* the two halves of an all-reduce (buf: world * shard_count floats; rank r's shard at buf + r * shard_count), for an optimiser
 * sharded over the ranks: reduce-scatter the gradient (mean), update 1 / world of the parameters, all-gather them */
int vf_comm_reduce_scatter_avg_async(vf_comm* c, vf_ctx* ctx, float* buf, int64_t shard_count, int* ticket);
int vf_comm_allgather_async(vf_comm* c, vf_ctx* ctx, float* buf, int64_t shard_count, int* ticket);
/* rank `root`'s buf[0..count) to every rank on the exchange stream (ticketed like the collectives above): the row blocks of a tensor
 * whose rows do not split evenly over the ranks travel as one broadcast per rank (vf_net_fused_adam_row_range) */
int vf_comm_broadcast_async(vf_comm* c, vf_ctx* ctx, float* buf, int64_t count, int root, int* ticket);
int vf_comm_wait(vf_comm* c, vf_ctx* ctx, int ticket);
int vf_comm_allreduce_inline(vf_comm* c, vf_ctx* ctx, void* buf, int64_t count, int dtype, int op);
int vf_comm_broadcast(vf_comm* c, vf_ctx* ctx, void* buf, int64_t count, int dtype, int root);
int vf_comm_barrier(vf_comm* c, vf_ctx* ctx);
int vf_comm_destroy(vf_comm* c);

/* ---- per-kernel timers (the reference has three torch.Timers, train.lua:241-243; these are finer) ----
 * Between vf_prof_begin and vf_prof_end every kernel launch of the library is bracketed by HIP events on the
 * context's stream.  vf_prof_end synchronises and aggregates per kernel name; vf_prof_get reads entry i:
 * launches, total milliseconds, and the ALGORITHMIC flops / bytes the launches were asked to do. */
int vf_prof_begin(vf_ctx* ctx);
int vf_prof_end(vf_ctx* ctx);
int vf_prof_count(void);
int vf_prof_get(int i, char* name, int name_cap, int64_t* launches, double* ms, double* flops, double* bytes);

/* ---- nn.Sequential as one object: the FAST path behind the boundary (net_{create, forward, backward,
 * update_grad_input, parameters}) ---------------------------------------------------------------------------------------------
 * util.cudnn(net) (util.lua:108-131) is where the reference swaps a net onto a GPU backend; from then on its drivers only call
 * net:forward / :backward / :updateGradInput / :getParameters / :apply(bias:zero()) / :zeroGradParameters / :evaluate
 * (train.lua:245-263, 278-410; train_vid_weighted.lua:330-355, 373-537).  vf_net is that net: hand over the layer list the
 * reference builds with netG:add(...) / netD:add(...) (train.lua:87-199) and drive it with one call per Torch7 method.  It
 * executes the plan with every cross-layer shortcut of the hot path (csrc/vf_net.hip): activations in their producer's
 * epilogue, BatchNorm statistics as a by-product of the neighbouring GEMMs (vf_bn_fuse_next_* / vf_bn_*_pre), operands as bf16
 * planes handed from producer to consumer (vf_pconv_*), weight planes of the whole net refreshed in one launch per parameter
 * update, every weight gradient of a walk in one grouped launch and every conv bias gradient in two, lazy gradient zeroing,
 * netD's real + fake passes as one batch of 2B with BatchNorm per half (vf_net_set_batch_groups), backward walks cut at a
 * gradient bucket (vf_net_backward_range), SyncBN over vf_comm_*.  The benched iteration runs through exactly these calls
 * (bench.py --host cabi; video-filler_amd/cnet.py and lua/hipnn.lua's hipnn.Net are thin hosts of it).
 * Chain nets only (the table modules of train.lua's option branches stay with a module-by-module host).
 * Activations NHWC, weights channels-last; the flat parameter buffer holds {weight, bias} ({gamma, beta}) module by module,
 * segments padded to 64 floats (vf_net_param_offset).  All calls enqueue on the context's stream; buffers whose need depends on
 * the path a pass takes (planes) are allocated at first use, so run one iteration before capturing a hipGraph. */
enum { VF_L_CONV = 1, VF_L_FULLCONV = 2, VF_L_BN = 3, VF_L_ACT = 4, VF_L_VIEW = 5 };
typedef struct vf_layer_desc {
  int kind;             /* VF_L_* */
  int nin, nout;        /* conv / full-conv planes; BatchNorm: nout = channels */
  int k, stride, pad;   /* conv / full-conv */
  int act;              /* VF_L_ACT: VF_ACT_LRELU / RELU / TANH / SIGMOID */
  float slope;          /* LeakyReLU negval */
  float eps, momentum;  /* BatchNorm (0 -> 1e-5 / 0.1) */
} vf_layer_desc;
typedef struct vf_net vf_net;
int vf_net_create(vf_ctx* ctx, vf_net** out, const vf_layer_desc* layers, int nlayers, int B, int C, int H, int W);
int vf_net_destroy(vf_net* net);
/* a new input shape: activations / planes / BatchNorm scratch re-planned, parameters and running statistics kept (Torch7
 * modules resize their outputs on the fly).  Synchronises; not inside a capture. */
int vf_net_reshape(vf_net* net, int B, int C, int H, int W);
int vf_net_parameters(vf_net* net, float** params, float** grads, int64_t* count);   /* net:getParameters() */
/* host-owned flat storage (count >= vf_net_parameters' count floats each, 16-byte aligned, same layout) instead of the net's
 * own: Torch7 keeps parameters in Lua-owned tensors, whose weight / bias views all alias ONE storage after getParameters().
 * Nothing is copied.  NULL, NULL: back to the net's own buffers. */
int vf_net_bind_parameters(vf_net* net, float* params, float* grads, int64_t count);
int64_t vf_net_param_offset(const vf_net* net, int layer, int which, int64_t* length); /* which: 0 weight/gamma, 1 bias/beta; -1 if none */
int vf_net_bn_running(vf_net* net, int layer, float** running_mean, float** running_var);
int vf_net_bind_bn_running(vf_net* net, int layer, float* running_mean, float* running_var);   /* host-owned running statistics */
int vf_net_bn_saved(vf_net* net, int layer, float** save_mean, float** save_invstd);          /* [groups][C] of the last forward */
int vf_net_training(vf_net* net, int train);                                          /* net:training() / net:evaluate() */
int vf_net_zero_grad(vf_net* net);                                                    /* net:zeroGradParameters() (lazy) */
/* the conv-bias sweep of both closures (train.lua:279-280), one launch; `other` (may be NULL): the second net of the sweep */
int vf_net_zero_conv_biases(vf_net* net, vf_net* other);
int vf_net_forward(vf_net* net, const float* x, const float** y);                     /* net:forward(input) */
int vf_net_backward(vf_net* net, const float* x, const float* gy, const float** gx);  /* net:backward(input, gradOutput) */
int vf_net_update_grad_input(vf_net* net, const float* x, const float* gy, const float** gx); /* train.lua:366 */
/* vf_net_backward leaves out the first layer's gradInput (*gx = NULL): Torch7 always computes it, the drivers never read it for
 * netD's two full backward passes nor for netG's (train.lua:318,348,403) */
int vf_net_set_skip_input_grad(vf_net* net, int on);
/* The next forwards carry G concatenated, independent batches (netD's [real; fake], train.lua:331-349): BatchNorm statistics,
 * running-average updates (in group order) and backward sums per group — what G separate calls compute — while the convolutions
 * see one batch.  vf_net_update_grad_input_group: updateGradInput over group g alone (x / gy hold that group's samples; saved
 * activations and statistics are that group's): fGx's pass over the fake half. */
int vf_net_set_batch_groups(vf_net* net, int G);
int vf_net_update_grad_input_group(vf_net* net, const float* x, const float* gy, int g, int G, const float** gx);
/* Data parallel: the walk cut where a gradient bucket is complete.  The net's execution plan has vf_net_plan_size entries (an
 * absorbed activation shares its producer's); vf_net_bucket_split gives the plan index k and flat offset of the shortest tail
 * plan[k:] owning >= frac of the parameters; vf_net_backward_range(hi, lo) runs backward over entries hi-1 .. lo (hi < 0: from
 * the top; gy = what the entry above hi returned).  After the walk over [k, top) the flat gradient [offset, end) is final. */
int vf_net_plan_size(const vf_net* net);
/* 1 if the last forward left the SIGN BITS of this layer's activated output beside it (a 3-channel-input conv + (Leaky)ReLU): the
 * data-gradient pass of the conv above then reads its derivative mask from them — one broadcast word per pixel and 32 channels
 * instead of the fp32 activation.  Internal to the net object; not under an activation observer. */
int vf_net_layer_has_act_bits(const vf_net* net, int layer);
int vf_net_bucket_split(const vf_net* net, double frac, int* plan_index, int64_t* flat_offset);
int vf_net_backward_range(vf_net* net, const float* x, const float* gy, int hi, int lo, int need_input_grad, const float** gx);
/* the same cut without interrupting the data-gradient chain: vf_net_backward_split walks the WHOLE net, then launches the weight
 * and bias gradients of plan entries >= k only (flat gradient [offset, end) final: start its exchange); vf_net_backward_finish
 * launches the gradients of the entries below k.  Nothing else may record weight gradients on the context in between. */
int vf_net_backward_split(vf_net* net, const float* x, const float* gy, int k, int need_input_grad, const float** gx);
int vf_net_backward_finish(vf_net* net);
/* optim.adam fused into the bottleneck pair's weight gradients (vf_wgrad_adam_outer).  vf_net_set_fused_adam(1): backward walks
 * leave out the weight gradient of every layer the fused kernel takes and remember its operands (the module buffers, untouched
 * until the next forward); *count = the number of such layers (0: nothing to fuse — keep the plain update).  vf_net_fused_adam_range
 * names the i-th layer's weight slice of the flat vectors, which the host's own vf_adam_apply calls must leave out.
 * vf_net_adam_fused, after the backward pass and vf_adam_prep: the fused kernel per marked layer (m, v: the optimiser's flat state,
 * laid out like the parameters; keep_grad: gradParameters receives those slices too).  A marked layer whose gradient was not fresh
 * (a second backward without zeroGradParameters) was accumulated the plain way and gets vf_adam_apply on its slice. */
int vf_net_set_fused_adam(vf_net* net, int on, int* count);
int vf_net_fused_adam_range(const vf_net* net, int i, int64_t* offset, int64_t* length);
int vf_net_adam_fused(vf_net* net, float* m, float* v, double beta1, double beta2, double eps, const int32_t* t_dev, int keep_grad);
/* Data parallel with the fused update: the two weight gradients (92 % of the generator's bytes) are NOT exchanged.  Each rank packs
 * the operands they are the product of — batch x (Nu + Ncols) floats per layer, 6 MB at batchSize 64 against 262 MB of gradient —
 * into its segment of a gather buffer (vf_net_fused_adam_pack, after the backward pass; vf_net_fused_adam_pack_size floats, a
 * multiple of 4), the host all-gathers the segments (vf_comm_allgather_async), and vf_net_adam_fused_gathered forms the gradient
 * of the GLOBAL batch in the fused kernel on every rank (world * batch rows, scaled by 1 / world: the mean over ranks, the same
 * bits on every rank) and applies the update.  The rest of the flat gradient is all-reduced as before. */
int vf_net_fused_adam_pack_size(const vf_net* net, int64_t* floats);
int vf_net_fused_adam_pack(vf_net* net, float* segment);
int vf_net_adam_fused_gathered(vf_net* net, const float* all_segments, int world, int64_t seg_stride, float* m, float* v, double beta1,
                               double beta2, double eps, const int32_t* t_dev, int keep_grad);
/* ... with the update sharded by weight ROWS: this rank (row_rank of row_world) forms the global-batch gradient of its 1 / row_world of
 * every fused tensor's rows and updates them (m, v: those rows only); the host then all-gathers the updated rows of each
 * vf_net_fused_adam_range slice (equal, contiguous row blocks).  vf_net_fused_adam_rows_ok: do the row counts split that way? */
int vf_net_fused_adam_rows_ok(const vf_net* net, int row_world);
/* The row blocks: rank r of row_world owns rows [r * bs, min(Nu, (r + 1) * bs)), bs = 2 * ceil(Nu / (2 * row_world)) — equal blocks
 * where the rows split evenly, a shorter last block where they do not (every block at least 64 rows: vf_net_fused_adam_rows_ok).
 * vf_net_fused_adam_row_range: that block of fused layer i (of vf_net_set_fused_adam's count) as an element range of the flat vectors. */
int vf_net_fused_adam_row_range(const vf_net* net, int i, int row_rank, int row_world, int64_t* offset, int64_t* length);
/* Hiding the exchange of the updated rows: the host issues it on the communicator's stream right after the update (vf_comm_allgather_async
 * / vf_comm_broadcast_async) and hands the tickets to the net; the NEXT vf_net_forward runs the layers in front of the bottleneck conv
 * (train.lua:89-104; they read none of those weights) beside the transfer and waits for the tickets in front of the first layer the fused
 * update takes.  One-shot. */
int vf_net_forward_wait_fused(vf_net* net, vf_comm* comm, int ticket);
int vf_net_adam_fused_gathered_rows(vf_net* net, const float* all_segments, int world, int64_t seg_stride, float* m, float* v, double beta1,
                                    double beta2, double eps, const int32_t* t_dev, int keep_grad, int row_rank, int row_world);
/* SyncBN: BatchNorm sums all-reduced over `comm` (world ranks; statistics of the global batch).  force: take that path at world 1
 * too.  comm NULL / world 1 / force 0: device-local statistics.  world > 1 with comm NULL is refused (it would silently be local). */
int vf_net_set_sync_bn(vf_net* net, vf_comm* comm, int world, int force);
/* Weight planes (bf16 shadows of the conv weights the planes kernels read).  Unmanaged (default): refreshed at the start of every
 * forward / backward call.  Managed: the host calls vf_net_refresh_weight_planes once after each parameter update (optim.adam, a
 * checkpoint load) — one launch per net and update instead of one per call. */
int vf_net_set_weight_planes_managed(vf_net* net, int on);
int vf_net_refresh_weight_planes(vf_net* net);
/* where the planes kernels are used (process-wide; defaults 3 GFLOP per pass and 1024 GEMM rows) */
int vf_net_set_planes_gate(double min_gflop_per_pass, int min_rows);
int vf_net_layer_output(vf_net* net, int layer, const float** y);                     /* net.modules[i].output */
int vf_net_layer_grad_input(vf_net* net, int layer, const float** gx);                /* net.modules[i].gradInput */
int vf_net_layer_shape(const vf_net* net, int layer, int* B, int* C, int* H, int* W, int* Co, int* Ho, int* Wo);
/* module `layer` writes its output into the host's buffer (e.g. netG's last convolution straight into the fake half of netD's
 * [real; fake] input); NULL: the net's own buffer again */
int vf_net_bind_output(vf_net* net, int layer, float* y);
/* Observer of every (Leaky)ReLU output of a forward pass, called on the host right after the producing launch was enqueued:
 * fn(user, layer index of the activation module, device pointer, element count) -> 1 if it edited the tensor (on the context's
 * stream), 0 if not, < 0 on error.  The parity tests pin derivative choices at the kink with it; NULL: none. */
typedef int (*vf_net_act_observer)(void* user, int layer, float* y, int64_t numel);
int vf_net_set_act_observer(vf_net* net, vf_net_act_observer fn, void* user);

/* ---- roctx ranges (the reference's tracing is three torch.Timers; this is the profiler-visible counterpart) ----
 * vf_range_push / vf_range_pop / vf_mark forward to roctx when a roctx library is present (librocprofiler-sdk-roctx.so as
 * injected by rocprofv3, else libroctx64.so; $VF_ROCTX_LIB overrides) and are no-ops otherwise: `rocprofv3 --marker-trace
 * --kernel-trace` shows them beside the kernels.  vf_trace_enable(1) (or VF_ROCTX=1 in the environment) additionally puts one
 * range around every launch site of the library, named like the vf_prof_* kernel table.  vf_trace_available: 1 if bound.
 * vf_range_depth: ranges currently open (pushes minus pops). */
int vf_trace_available(void);
int vf_trace_enable(int on);
int vf_range_push(const char* name);
int vf_range_pop(void);
int vf_mark(const char* message);
int vf_range_depth(void);

#ifdef __cplusplus
}
#endif
#endif /* VF_HIP_H */
