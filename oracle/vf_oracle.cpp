// vf_oracle.cpp — CPU restatement of the reference's hot-path arithmetic.
//
// TEST INFRASTRUCTURE ONLY.  Nothing under video-filler_amd/ may include, link or
// call this file; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
// leg do.  The product path is the HIP library behind include/vf_hip.h.
//
// PARITY UNPINNED: the reference (/root/reference, Lua/Torch7) has no tests, golden
// vectors or fixtures, and Torch7 cannot be run in this pipeline (no Lua, no network).
// The arithmetic the reference executes lives in un-vendored, un-versioned Torch7
// packages (nn/THNN, optim, nngraph; Sept-2016 era).  Each function below restates
// the published THNN / optim algorithm and cites the reference call site that
// reaches it.  The restatement is pinned by (i) PyTorch-CPU functional ops where
// PyTorch agrees with Torch7 (tests/test_oracle_vs_torch.py), (ii) central finite
// differences in double, (iii) the committed fixtures under tests/golden/.
//
// Layout: exactly the reference's — fp32, NCHW, conv weights [Cout][Cin][kH][kW],
// full-conv weights [Cin][Cout][kH][kW], row-major, one contiguous block each.
//
// Build: see oracle/Makefile (g++ -O3 -ffp-contract=off -fopenmp).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

#define VFO_API extern "C" __attribute__((visibility("default")))

static int g_threads = 1;  // reference: torch.setnumthreads(1)  (train.lua:47)

VFO_API void vfo_set_num_threads(int n) { g_threads = n < 1 ? 1 : n; }
VFO_API int vfo_get_num_threads() { return g_threads; }

// ---------------------------------------------------------------------------
// sgemm: C[M,N] = beta*C + A[M,K] * B[K,N]   (row-major, fp32 accumulate like BLAS sgemm)
// THNN SpatialConvolutionMM / SpatialFullConvolution call THBlas_(gemm); summation
// order inside BLAS is unspecified, so any fp32 order is an equally valid restatement.
// ---------------------------------------------------------------------------
static void sgemm_nn(int M, int N, int K, const float* A, int lda, const float* B, int ldb,
                     float* C, int ldc, float beta) {
  constexpr int MR = 4, NR = 32;
  for (int j0 = 0; j0 < N; j0 += NR) {
    const int nr = std::min(NR, N - j0);
    for (int i0 = 0; i0 < M; i0 += MR) {
      const int mr = std::min(MR, M - i0);
      float acc[MR][NR];
      for (int r = 0; r < MR; ++r)
        for (int c = 0; c < NR; ++c) acc[r][c] = 0.f;
      if (mr == MR && nr == NR) {
        for (int k = 0; k < K; ++k) {
          const float* b = B + (size_t)k * ldb + j0;
          for (int r = 0; r < MR; ++r) {
            const float a = A[(size_t)(i0 + r) * lda + k];
#pragma omp simd
            for (int c = 0; c < NR; ++c) acc[r][c] += a * b[c];
          }
        }
      } else {
        for (int k = 0; k < K; ++k) {
          const float* b = B + (size_t)k * ldb + j0;
          for (int r = 0; r < mr; ++r) {
            const float a = A[(size_t)(i0 + r) * lda + k];
            for (int c = 0; c < nr; ++c) acc[r][c] += a * b[c];
          }
        }
      }
      for (int r = 0; r < mr; ++r) {
        float* c = C + (size_t)(i0 + r) * ldc + j0;
        if (beta == 0.f)
          for (int cc = 0; cc < nr; ++cc) c[cc] = acc[r][cc];
        else
          for (int cc = 0; cc < nr; ++cc) c[cc] = beta * c[cc] + acc[r][cc];
      }
    }
  }
}

static void transpose(const float* A, int rows, int cols, float* At) {  // At[cols][rows]
  for (int r = 0; r < rows; ++r)
    for (int c = 0; c < cols; ++c) At[(size_t)c * rows + r] = A[(size_t)r * cols + c];
}

// im2col / col2im as in THNN unfolded_copy / unfolded_acc (SpatialConvolutionMM.c).
// col is [C*kH*kW][Ho*Wo].
static void im2col(const float* x, int C, int H, int W, int kH, int kW, int dH, int dW, int pH,
                   int pW, int Ho, int Wo, float* col) {
  for (int c = 0; c < C; ++c)
    for (int u = 0; u < kH; ++u)
      for (int v = 0; v < kW; ++v) {
        float* dst = col + ((size_t)(c * kH + u) * kW + v) * Ho * Wo;
        for (int i = 0; i < Ho; ++i) {
          const int y = i * dH - pH + u;
          for (int j = 0; j < Wo; ++j) {
            const int xx = j * dW - pW + v;
            dst[i * Wo + j] =
                (y >= 0 && y < H && xx >= 0 && xx < W) ? x[((size_t)c * H + y) * W + xx] : 0.f;
          }
        }
      }
}

static void col2im_acc(const float* col, int C, int H, int W, int kH, int kW, int dH, int dW,
                       int pH, int pW, int Ho, int Wo, float* x) {
  for (int c = 0; c < C; ++c)
    for (int u = 0; u < kH; ++u)
      for (int v = 0; v < kW; ++v) {
        const float* src = col + ((size_t)(c * kH + u) * kW + v) * Ho * Wo;
        for (int i = 0; i < Ho; ++i) {
          const int y = i * dH - pH + u;
          if (y < 0 || y >= H) continue;
          for (int j = 0; j < Wo; ++j) {
            const int xx = j * dW - pW + v;
            if (xx >= 0 && xx < W) x[((size_t)c * H + y) * W + xx] += src[i * Wo + j];
          }
        }
      }
}

// ---------------------------------------------------------------------------
// nn.SpatialConvolution.  Reached from train.lua:89-104,183-196,
// train_vid_weighted.lua:114-129,214-233, train_wholeim_input.lua:137-153,238-257.
// ---------------------------------------------------------------------------
VFO_API void vfo_conv2d_fwd(const float* x, const float* w, const float* bias, float* y, int B,
                            int Cin, int H, int W, int Cout, int kH, int kW, int dH, int dW, int pH,
                            int pW) {
  const int Ho = (H + 2 * pH - kH) / dH + 1, Wo = (W + 2 * pW - kW) / dW + 1;
  const int K = Cin * kH * kW, P = Ho * Wo;
#pragma omp parallel num_threads(g_threads)
  {
    std::vector<float> col((size_t)K * P);
#pragma omp for schedule(static)
    for (int b = 0; b < B; ++b) {
      im2col(x + (size_t)b * Cin * H * W, Cin, H, W, kH, kW, dH, dW, pH, pW, Ho, Wo, col.data());
      float* yb = y + (size_t)b * Cout * P;
      for (int o = 0; o < Cout; ++o) {
        const float bv = bias ? bias[o] : 0.f;
        for (int p = 0; p < P; ++p) yb[(size_t)o * P + p] = bv;
      }
      sgemm_nn(Cout, P, K, w, K, col.data(), P, yb, P, 1.f);
    }
  }
}

// updateGradInput: fgradInput = W^T * gradOutput ; col2im.  Depends on input only for shape.
VFO_API void vfo_conv2d_bwd_input(const float* gy, const float* w, float* gx, int B, int Cin, int H,
                                  int W, int Cout, int kH, int kW, int dH, int dW, int pH, int pW) {
  const int Ho = (H + 2 * pH - kH) / dH + 1, Wo = (W + 2 * pW - kW) / dW + 1;
  const int K = Cin * kH * kW, P = Ho * Wo;
  std::vector<float> wt((size_t)K * Cout);
  transpose(w, Cout, K, wt.data());
#pragma omp parallel num_threads(g_threads)
  {
    std::vector<float> col((size_t)K * P);
#pragma omp for schedule(static)
    for (int b = 0; b < B; ++b) {
      sgemm_nn(K, P, Cout, wt.data(), Cout, gy + (size_t)b * Cout * P, P, col.data(), P, 0.f);
      float* gxb = gx + (size_t)b * Cin * H * W;
      std::memset(gxb, 0, sizeof(float) * (size_t)Cin * H * W);
      col2im_acc(col.data(), Cin, H, W, kH, kW, dH, dW, pH, pW, Ho, Wo, gxb);
    }
  }
}

// accGradParameters: gradWeight += scale * gradOutput * finput^T ; gradBias += scale * sum(gradOutput)
VFO_API void vfo_conv2d_acc_grad(const float* x, const float* gy, float* gw, float* gb, int B,
                                 int Cin, int H, int W, int Cout, int kH, int kW, int dH, int dW,
                                 int pH, int pW, float scale) {
  const int Ho = (H + 2 * pH - kH) / dH + 1, Wo = (W + 2 * pW - kW) / dW + 1;
  const int K = Cin * kH * kW, P = Ho * Wo;
  std::vector<float> col((size_t)K * P), colT((size_t)P * K), tmp((size_t)Cout * K);
  for (int b = 0; b < B; ++b) {  // THNN accumulates sample by sample
    im2col(x + (size_t)b * Cin * H * W, Cin, H, W, kH, kW, dH, dW, pH, pW, Ho, Wo, col.data());
    transpose(col.data(), K, P, colT.data());
    const float* gyb = gy + (size_t)b * Cout * P;
    sgemm_nn(Cout, K, P, gyb, P, colT.data(), K, tmp.data(), K, 0.f);
    for (size_t i = 0; i < (size_t)Cout * K; ++i) gw[i] += scale * tmp[i];
    if (gb)
      for (int o = 0; o < Cout; ++o) {
        float s = 0.f;
        for (int p = 0; p < P; ++p) s += gyb[(size_t)o * P + p];
        gb[o] += scale * s;
      }
  }
}

// ---------------------------------------------------------------------------
// nn.SpatialFullConvolution, adj = 0.  train.lua:134-146,
// train_vid_weighted.lua:159-174.  Weight [Cin][Cout][kH][kW].
// ---------------------------------------------------------------------------
VFO_API void vfo_fullconv2d_fwd(const float* x, const float* w, const float* bias, float* y, int B,
                                int Cin, int H, int W, int Cout, int kH, int kW, int dH, int dW,
                                int pH, int pW) {
  const int Ho = (H - 1) * dH - 2 * pH + kH, Wo = (W - 1) * dW - 2 * pW + kW;
  const int K = Cout * kH * kW, P = H * W;  // columns: [Cout*kH*kW][H*W]
  std::vector<float> wt((size_t)K * Cin);
  transpose(w, Cin, K, wt.data());
#pragma omp parallel num_threads(g_threads)
  {
    std::vector<float> col((size_t)K * P);
#pragma omp for schedule(static)
    for (int b = 0; b < B; ++b) {
      sgemm_nn(K, P, Cin, wt.data(), Cin, x + (size_t)b * Cin * P, P, col.data(), P, 0.f);
      float* yb = y + (size_t)b * Cout * Ho * Wo;
      std::memset(yb, 0, sizeof(float) * (size_t)Cout * Ho * Wo);
      // col2im over the OUTPUT image: output plays the role of the "image", input grid = (H,W)
      col2im_acc(col.data(), Cout, Ho, Wo, kH, kW, dH, dW, pH, pW, H, W, yb);
      if (bias)
        for (int o = 0; o < Cout; ++o)
          for (int p = 0; p < Ho * Wo; ++p) yb[(size_t)o * Ho * Wo + p] += bias[o];
    }
  }
}

VFO_API void vfo_fullconv2d_bwd_input(const float* gy, const float* w, float* gx, int B, int Cin,
                                      int H, int W, int Cout, int kH, int kW, int dH, int dW, int pH,
                                      int pW) {
  const int Ho = (H - 1) * dH - 2 * pH + kH, Wo = (W - 1) * dW - 2 * pW + kW;
  const int K = Cout * kH * kW, P = H * W;
#pragma omp parallel num_threads(g_threads)
  {
    std::vector<float> col((size_t)K * P);
#pragma omp for schedule(static)
    for (int b = 0; b < B; ++b) {
      im2col(gy + (size_t)b * Cout * Ho * Wo, Cout, Ho, Wo, kH, kW, dH, dW, pH, pW, H, W, col.data());
      sgemm_nn(Cin, P, K, w, K, col.data(), P, gx + (size_t)b * Cin * P, P, 0.f);
    }
  }
}

VFO_API void vfo_fullconv2d_acc_grad(const float* x, const float* gy, float* gw, float* gb, int B,
                                     int Cin, int H, int W, int Cout, int kH, int kW, int dH, int dW,
                                     int pH, int pW, float scale) {
  const int Ho = (H - 1) * dH - 2 * pH + kH, Wo = (W - 1) * dW - 2 * pW + kW;
  const int K = Cout * kH * kW, P = H * W;
  std::vector<float> col((size_t)K * P), colT((size_t)P * K), tmp((size_t)Cin * K);
  for (int b = 0; b < B; ++b) {
    const float* gyb = gy + (size_t)b * Cout * Ho * Wo;
    im2col(gyb, Cout, Ho, Wo, kH, kW, dH, dW, pH, pW, H, W, col.data());
    transpose(col.data(), K, P, colT.data());
    sgemm_nn(Cin, K, P, x + (size_t)b * Cin * P, P, colT.data(), K, tmp.data(), K, 0.f);
    for (size_t i = 0; i < (size_t)Cin * K; ++i) gw[i] += scale * tmp[i];
    if (gb)
      for (int o = 0; o < Cout; ++o) {
        float s = 0.f;
        for (int p = 0; p < Ho * Wo; ++p) s += gyb[(size_t)o * Ho * Wo + p];
        gb[o] += scale * s;
      }
  }
}

// ---------------------------------------------------------------------------
// nn.SpatialBatchNormalization (THNN BatchNormalization.c, accreal = double).
// train.lua:92-101,125,135-144,189-193; eval: test_vid.lua:48.
// ---------------------------------------------------------------------------
VFO_API void vfo_bn_train_fwd(const float* x, float* y, const float* gamma, const float* beta,
                              float* running_mean, float* running_var, float* save_mean,
                              float* save_invstd, int B, int C, int HW, float momentum, float eps) {
  const double n = (double)B * HW;
#pragma omp parallel for num_threads(g_threads) schedule(static)
  for (int c = 0; c < C; ++c) {
    double sum = 0;
    for (int b = 0; b < B; ++b) {
      const float* p = x + ((size_t)b * C + c) * HW;
      for (int i = 0; i < HW; ++i) sum += p[i];
    }
    const double mean = sum / n;
    save_mean[c] = (float)mean;
    sum = 0;
    for (int b = 0; b < B; ++b) {
      const float* p = x + ((size_t)b * C + c) * HW;
      for (int i = 0; i < HW; ++i) sum += (p[i] - mean) * (p[i] - mean);
    }
    float invstd;
    if (sum == 0 && eps == 0.0f)
      invstd = 0;
    else
      invstd = (float)(1 / std::sqrt(sum / n + eps));
    save_invstd[c] = invstd;
    running_mean[c] = (float)(momentum * mean + (1 - momentum) * running_mean[c]);
    const double unbiased_var = sum / (n - 1);  // n == 1 -> inf/NaN, as the reference
    running_var[c] = (float)(momentum * unbiased_var + (1 - momentum) * running_var[c]);
    const float w = gamma ? gamma[c] : 1.f, bb = beta ? beta[c] : 0.f;
    for (int b = 0; b < B; ++b) {
      const float* p = x + ((size_t)b * C + c) * HW;
      float* q = y + ((size_t)b * C + c) * HW;
      for (int i = 0; i < HW; ++i) q[i] = (float)(((p[i] - mean) * invstd) * w + bb);
    }
  }
}

VFO_API void vfo_bn_eval_fwd(const float* x, float* y, const float* gamma, const float* beta,
                             const float* running_mean, const float* running_var, int B, int C,
                             int HW, float eps) {
#pragma omp parallel for num_threads(g_threads) schedule(static)
  for (int c = 0; c < C; ++c) {
    const double mean = running_mean[c];
    const float invstd = (float)(1 / std::sqrt((double)running_var[c] + eps));
    const float w = gamma ? gamma[c] : 1.f, bb = beta ? beta[c] : 0.f;
    for (int b = 0; b < B; ++b) {
      const float* p = x + ((size_t)b * C + c) * HW;
      float* q = y + ((size_t)b * C + c) * HW;
      for (int i = 0; i < HW; ++i) q[i] = (float)(((p[i] - mean) * invstd) * w + bb);
    }
  }
}

// gx, ggamma, gbeta each optional (updateGradInput passes only gx; accGradParameters only the
// parameter grads; Module:backward both).  Parameter grads ACCUMULATE (+=, scale).
VFO_API void vfo_bn_bwd(const float* x, const float* gy, float* gx, float* ggamma, float* gbeta,
                        const float* gamma, const float* save_mean, const float* save_invstd, int B,
                        int C, int HW, float scale) {
  const double n = (double)B * HW;
#pragma omp parallel for num_threads(g_threads) schedule(static)
  for (int c = 0; c < C; ++c) {
    const float w = gamma ? gamma[c] : 1.f;
    const double mean = save_mean[c];
    const float invstd = save_invstd[c];
    double sum = 0, dotp = 0;
    for (int b = 0; b < B; ++b) {
      const float* p = x + ((size_t)b * C + c) * HW;
      const float* g = gy + ((size_t)b * C + c) * HW;
      for (int i = 0; i < HW; ++i) {
        sum += g[i];
        dotp += (p[i] - mean) * g[i];
      }
    }
    if (gx) {
      const float k = (float)(dotp * invstd * invstd / n);
      const double gmean = sum / n;
      for (int b = 0; b < B; ++b) {
        const float* p = x + ((size_t)b * C + c) * HW;
        const float* g = gy + ((size_t)b * C + c) * HW;
        float* o = gx + ((size_t)b * C + c) * HW;
        for (int i = 0; i < HW; ++i) {
          const float t = (float)((p[i] - mean) * k);
          o[i] = (float)((g[i] - gmean - t) * invstd * w);
        }
      }
    }
    if (ggamma) ggamma[c] += (float)(scale * dotp * invstd);
    if (gbeta) gbeta[c] += (float)(scale * sum);
  }
}

// ---------------------------------------------------------------------------
// Activations.  In-place forms, as the reference builds them
// (nn.LeakyReLU(0.2,true), nn.ReLU(true)): forward overwrites x; backward reads the
// ACTIVATED values y and overwrites gy.
// ---------------------------------------------------------------------------
VFO_API void vfo_lrelu_fwd(float* x, size_t n, float slope) {
  for (size_t i = 0; i < n; ++i) x[i] = x[i] > 0 ? x[i] : x[i] * slope;
}
VFO_API void vfo_lrelu_bwd(const float* y, float* gy, size_t n, float slope) {
  for (size_t i = 0; i < n; ++i) gy[i] = y[i] > 0 ? gy[i] : gy[i] * slope;
}
VFO_API void vfo_relu_fwd(float* x, size_t n) {
  for (size_t i = 0; i < n; ++i) x[i] = x[i] > 0 ? x[i] : 0.f;
}
VFO_API void vfo_relu_bwd(const float* y, float* gy, size_t n) {
  for (size_t i = 0; i < n; ++i) gy[i] = y[i] > 0 ? gy[i] : 0.f;
}
VFO_API void vfo_tanh_fwd(const float* x, float* y, size_t n) {
  for (size_t i = 0; i < n; ++i) y[i] = std::tanh(x[i]);
}
VFO_API void vfo_tanh_bwd(const float* y, const float* gy, float* gx, size_t n) {
  for (size_t i = 0; i < n; ++i) gx[i] = gy[i] * (1.f - y[i] * y[i]);
}
VFO_API void vfo_sigmoid_fwd(const float* x, float* y, size_t n) {
  for (size_t i = 0; i < n; ++i) y[i] = 1.f / (1.f + std::exp(-x[i]));
}
VFO_API void vfo_sigmoid_bwd(const float* y, const float* gy, float* gx, size_t n) {
  for (size_t i = 0; i < n; ++i) gx[i] = gy[i] * (1.f - y[i]) * y[i];
}

// ---------------------------------------------------------------------------
// Criteria.
// ---------------------------------------------------------------------------
// nn.BCECriterion, eps = 1e-12, sizeAverage.  train.lua:204,312-313,342-343,364-365.
VFO_API double vfo_bce_fwd(const float* x, const float* t, size_t n) {
  const double EPS = 1e-12;
  double sum = 0;
  for (size_t i = 0; i < n; ++i)
    sum -= std::log((double)x[i] + EPS) * t[i] + std::log(1. - x[i] + EPS) * (1. - t[i]);
  return sum / (double)n;
}
VFO_API void vfo_bce_bwd(const float* x, const float* t, float* gx, size_t n) {
  const double EPS = 1e-12;
  const double norm = 1. / (double)n;
  for (size_t i = 0; i < n; ++i)
    gx[i] = (float)(-norm * ((double)t[i] - x[i]) / ((1. - x[i] + EPS) * (x[i] + EPS)));
}
// nn.MSECriterion, sizeAverage.  train.lua:207,377-378; train_vid_weighted.lua:272,489-496.
VFO_API double vfo_mse_fwd(const float* x, const float* t, size_t n) {
  double sum = 0;
  for (size_t i = 0; i < n; ++i) {
    const double z = (double)x[i] - t[i];
    sum += z * z;
  }
  return sum / (double)n;
}
VFO_API void vfo_mse_bwd(const float* x, const float* t, float* gx, size_t n) {
  const float norm = 2.f / (float)n;
  for (size_t i = 0; i < n; ++i) gx[i] = norm * (x[i] - t[i]);
}
// nn.AbsCriterion, sizeAverage (building block of GDL / MaskedMSE).
VFO_API double vfo_abs_fwd(const float* x, const float* t, size_t n) {
  double sum = 0;
  for (size_t i = 0; i < n; ++i) sum += std::fabs((double)x[i] - (t ? t[i] : 0.f));
  return sum / (double)n;
}

// nn.GDLCriterion(1) forward (gdl_criterion.lua:6-45; used train_vid_weighted.lua:523-526).
// input = Yhat, target = Y.  SpatialZeroPadding with negative pads crops (gdl_criterion.lua:12-19):
//   i1 = X[.., 0:H-1, :]   i2 = X[.., :, 0:W-1]   j1 = X[.., 1:H, :]   j2 = X[.., :, 1:W]
// CSubTable{i2, i1} subtracts two contiguous tensors of DIFFERENT shape but equal element count in
// flattened row-major order (requires H == W), plane-aligned.
VFO_API double vfo_gdl_fwd(const float* yhat, const float* y, int B, int C, int H, int W) {
  if (H != W) return NAN;  // the reference errors ("inconsistent tensor size")
  const size_t planes = (size_t)B * C, m = (size_t)(H - 1) * W;  // == H*(W-1)
  double s12 = 0, s34 = 0;
  std::vector<float> a(m), b(m), ah(m), bh(m);
  for (size_t p = 0; p < planes; ++p) {
    const float* Y = y + p * H * W;
    const float* Yh = yhat + p * H * W;
    // flattened crops
    size_t e = 0;
    for (int r = 0; r < H; ++r)
      for (int c = 0; c < W - 1; ++c, ++e) {  // i2: cols 0..W-2 ; j2: cols 1..W-1
        a[e] = Y[r * W + c];
        ah[e] = Yh[r * W + c];
        b[e] = Y[r * W + c + 1];
        bh[e] = Yh[r * W + c + 1];
      }
    // i1 flattened = rows 0..H-2 (all cols) = Y[0 .. m) ; j1 flattened = rows 1..H-1 = Y[W .. W+m)
    for (size_t k = 0; k < m; ++k) {
      const float t1 = std::fabs(a[k] - Y[k]);         // |Yi2 - Yi1|
      const float t2 = std::fabs(ah[k] - Yh[k]);       // |Yhati2 - Yhati1|
      const float t3 = std::fabs(b[k] - Y[W + k]);     // |Yj2 - Yj1|
      const float t4 = std::fabs(bh[k] - Yh[W + k]);   // |Yhatj2 - Yhatj1|
      s12 += std::fabs((double)(t1 - t2));
      s34 += std::fabs((double)(t3 - t4));
    }
  }
  const double cnt = (double)planes * m;
  return s12 / cnt + s34 / cnt;  // ParallelCriterion, weights 1, each AbsCriterion sizeAverage
}

// nn.GDLCriterion(1) updateGradInput (gdl_criterion.lua:47-53): d loss / d Yhat.  Never called by a driver of the reference
// (train_vid_weighted.lua:523-528 uses the forward value only); restated for the nn.Criterion protocol.
//   crit:updateGradInput = nn.AbsCriterion per output, target 0, sizeAverage: g_term = (term >= 0 ? 1 : -1) / count  (THNN
//   AbsCriterion_updateGradInput); term12 = term1 - term2 with term2 = |Yhat_i2 - Yhat_i1| the only part that depends on Yhat:
//   g_d2 = -g_term12 * (d2 >= 0 ? 1 : -1) (THNN Abs_updateGradInput), d2 = Yhat_i2 - Yhat_i1 in the flattened pairing;
//   CSubTable hands +g_d2 to its first input and -g_d2 to its second; the negative SpatialZeroPaddings put zeros back
//   where they cropped.  The same for term34 with the j crops.
VFO_API int vfo_gdl_bwd(const float* yhat, const float* y, float* gyhat, int B, int C, int H, int W) {
  if (H != W) return 1;
  const size_t planes = (size_t)B * C, m = (size_t)(H - 1) * W;
  const float norm = (float)(1.0 / ((double)planes * (double)m));
  for (size_t p = 0; p < planes; ++p) {
    const float* Y = y + p * H * W;
    const float* Yh = yhat + p * H * W;
    float* G = gyhat + p * H * W;
    for (size_t i = 0; i < (size_t)H * W; ++i) G[i] = 0.f;
    for (size_t k = 0; k < m; ++k) {
      const size_t r = k / (W - 1), c = k % (W - 1);
      const size_t i2 = r * W + c, j2 = i2 + 1, i1 = k, j1 = W + k;
      const float d1 = Y[i2] - Y[i1], d2 = Yh[i2] - Yh[i1], d3 = Y[j2] - Y[j1], d4 = Yh[j2] - Yh[j1];
      const float t12 = std::fabs(d1) - std::fabs(d2), t34 = std::fabs(d3) - std::fabs(d4);
      const float g2 = -(t12 >= 0 ? norm : -norm) * (d2 >= 0 ? 1.f : -1.f);
      const float g4 = -(t34 >= 0 ? norm : -norm) * (d4 >= 0 ? 1.f : -1.f);
      G[i2] += g2;
      G[i1] -= g2;
      G[j2] += g4;
      G[j1] -= g4;
    }
  }
  return 0;
}

// nn.MaskedMSECriterion(w) (MaskedMSECriterion.lua:7-42).  wM = (1-w)*M + w ;
// L = mean(|wM * (X - Xhat)^2|) ; dL/dX = (2/N) * wM * (X - Xhat) * sign(wM*(X-Xhat)^2 >= 0 -> +1).
VFO_API double vfo_masked_mse_fwd(const float* x, const float* xhat, const uint8_t* mask, float w,
                                  size_t n) {
  double sum = 0;
  for (size_t i = 0; i < n; ++i) {
    const double wm = (1.0 - w) * mask[i] + w;
    const double d = (double)x[i] - xhat[i];
    sum += std::fabs(wm * d * d);
  }
  return sum / (double)n;
}
VFO_API void vfo_masked_mse_bwd(const float* x, const float* xhat, const uint8_t* mask, float w,
                                float* gx, size_t n) {
  for (size_t i = 0; i < n; ++i) {
    const double wm = (1.0 - w) * mask[i] + w;
    const double d = (double)x[i] - xhat[i];
    const double sgn = (wm * d * d) >= 0 ? 1.0 : -1.0;  // AbsCriterion: (x - 0 >= 0 ? 1 : -1)/N
    gx[i] = (float)(sgn / (double)n * wm * 2.0 * d);
  }
}

// maskedSelect + maskedCopy composite (train_vid_weighted.lua:430-432, inpaint_utils.lua:63-101):
// out = real everywhere, then positions with mask != 0 take fake's value at the same position.
VFO_API void vfo_masked_compose(float* out, const float* real, const float* fake, const float* mask,
                                size_t n) {
  for (size_t i = 0; i < n; ++i) out[i] = mask[i] != 0.f ? fake[i] : real[i];
}

// train_vid_weighted.lua:493-503 — fused restatement of
//   weights = mask*(1-lambda)+lambda ; g_l2 = (2/N)(x - t) .* weights ; df_dg = (1-wtl2)*df_dg + wtl2*g_l2
// kept as separate primitive steps in oracle.py; this helper is only the in-place mask -> weights op.
VFO_API void vfo_mask_to_weights(float* mask, float lambda, size_t n) {
  for (size_t i = 0; i < n; ++i) {
    mask[i] = mask[i] * (1.f - lambda);
    mask[i] = mask[i] + lambda;
  }
}

// ---------------------------------------------------------------------------
// optim.adam (call sites train.lua:421-424, config train.lua:219-226).
// Tensor-op-by-tensor-op in fp32 exactly as optim/adam.lua sequences them.
// ---------------------------------------------------------------------------
VFO_API void vfo_adam_step(float* x, const float* g, float* m, float* v, float* denom, size_t n,
                           double lr, double beta1, double beta2, double eps, int t) {
  const float b1 = (float)beta1, omb1 = (float)(1 - beta1);
  const float b2 = (float)beta2, omb2 = (float)(1 - beta2);
  const float epsf = (float)eps;
  const double bc1 = 1 - std::pow(beta1, t), bc2 = 1 - std::pow(beta2, t);
  const float step = (float)(lr * std::sqrt(bc2) / bc1);
  for (size_t i = 0; i < n; ++i) {
    float mi = m[i] * b1;            // state.m:mul(beta1)
    mi = mi + omb1 * g[i];           //        :add(1-beta1, dfdx)
    float vi = v[i] * b2;            // state.v:mul(beta2)
    vi = vi + omb2 * g[i] * g[i];    //        :addcmul(1-beta2, dfdx, dfdx)
    float d = std::sqrt(vi);         // state.denom:copy(v):sqrt()
    d = d + epsf;                    //            :add(epsilon)
    m[i] = mi;
    v[i] = vi;
    if (denom) denom[i] = d;
    x[i] = x[i] - step * mi / d;     // x:addcdiv(-stepSize, m, denom)
  }
}
