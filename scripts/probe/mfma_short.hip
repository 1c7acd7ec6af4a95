// Probe: pure-register v_mfma_f32_32x32x2_f32 kernels of different lengths, back to back and inside a HIP graph.
// What fraction of the f32 MFMA peak can a SHORT kernel reach at all (launch ramp, drain, clocks)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void k_pure(float* out, int iters, float a0, float b0) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a = a0 + threadIdx.x, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j % NACC], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  if (s == 12345.f) out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
  float* out; hipMalloc(&out, 8192 * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipStream_t st; hipStreamCreate(&st);
  for (int blocks : {512, 1024, 2048}) {
    for (int iters : {8, 16, 32, 64, 128, 512, 4096}) {
      const int wps = blocks / 256;             // waves per SIMD (co-resident or in rounds)
      for (int w = 0; w < 5; ++w) k_pure<2><<<blocks, 256, 0, st>>>(out, iters, 1.f, 2.f);
      const int reps = 40;
      hipEventRecord(e0, st);
      for (int w = 0; w < reps; ++w) k_pure<2><<<blocks, 256, 0, st>>>(out, iters, 1.f, 2.f);
      hipEventRecord(e1, st); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
      const double fl = (double)blocks * 4 * iters * 16 * 4096.0;
      const double ideal_us = fl / 157.3e12 * 1e6;
      printf("blocks=%4d (%d waves/SIMD) mfma/wave=%6d  %8.2f us (ideal %7.2f)  %6.1f TFLOP/s  overhead %5.2f us\n", blocks, wps, iters * 16,
             ms * 1e3, ideal_us, fl / ms / 1e9, ms * 1e3 - ideal_us);
    }
  }
  return 0;
}
