// Probe: what limits v_mfma_f32_32x32x2_f32 throughput?  Variants of a register-only MFMA loop.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC, int MODE>
__global__ __launch_bounds__(256) void k_probe(float* out, int iters, const float* in, const float* gbig, size_t gbig_f4) {
  __shared__ __attribute__((aligned(16))) float lds[8192];
  const int lane = threadIdx.x & 63;
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a = in[threadIdx.x], b = in[threadIdx.x + 256];
  for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = in[i & 1023];
  __syncthreads();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {                       // pure register MFMA, 16 per iteration
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[j % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j % NACC], 0, 0, 0);
    } else if (MODE == 1) {                // operands from LDS b128 reads (like the conv kernel), no barrier
      f32x4 av[2], bv[2];
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        av[s] = *(const f32x4*)(lds + ((lane & 31) * 36 + 8 * s + 4 * (lane >> 5) + (it & 1) * 4096));
        bv[s] = *(const f32x4*)(lds + 2048 + ((lane & 31) * 36 + 8 * s + 4 * (lane >> 5)) + (it & 1) * 4096);
      }
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int t = 0; t < 2; ++t) acc[(2 * j + t) % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s][j], bv[s][j], acc[(2 * j + t) % NACC], 0, 0, 0);
    } else if (MODE == 2) {                // MODE 1 + one __syncthreads per iteration
      f32x4 av[2], bv[2];
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        av[s] = *(const f32x4*)(lds + ((lane & 31) * 36 + 8 * s + 4 * (lane >> 5) + (it & 1) * 4096));
        bv[s] = *(const f32x4*)(lds + 2048 + ((lane & 31) * 36 + 8 * s + 4 * (lane >> 5)) + (it & 1) * 4096);
      }
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int t = 0; t < 2; ++t) acc[(2 * j + t) % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s][j], bv[s][j], acc[(2 * j + t) % NACC], 0, 0, 0);
      __syncthreads();
    } else if (MODE >= 3) {                // full staging: NLD float4 global loads/thread -> LDS double buffer -> MFMA
      constexpr int NLD = MODE >= 3 ? MODE - 2 : 1;       // MODE 3..: 1.. float4 per thread per 16 MFMAs
      f32x4 r[NLD];
      const f32x4* g = (const f32x4*)gbig + ((size_t)blockIdx.x * 65536 + (size_t)it * 256 * NLD) % (gbig_f4 - 256 * NLD);
#pragma unroll
      for (int q = 0; q < NLD; ++q) r[q] = g[q * 256 + threadIdx.x];
      f32x4 av[2], bv[2];
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        av[s] = *(const f32x4*)(lds + ((lane & 31) * 36 + 8 * s + 4 * (lane >> 5) + (it & 1) * 4096));
        bv[s] = *(const f32x4*)(lds + 2048 + ((lane & 31) * 36 + 8 * s + 4 * (lane >> 5)) + (it & 1) * 4096);
      }
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int t = 0; t < 2; ++t) acc[(2 * j + t) % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s][j], bv[s][j], acc[(2 * j + t) % NACC], 0, 0, 0);
#pragma unroll
      for (int q = 0; q < NLD; ++q) *(f32x4*)(lds + ((it + 1) & 1) * 4096 + ((q * 256 + threadIdx.x) * 4) % 4096) = r[q];
      __syncthreads();
    }
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC, int MODE>
void run(const char* name, int blocks, float* out, const float* in, const float* gbig = nullptr, size_t gbig_f4 = 0) {
  const int iters = getenv("ITERS") ? atoi(getenv("ITERS")) : 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 5; ++w) k_probe<NACC, MODE><<<blocks, 256>>>(out, iters, in, gbig, gbig_f4);
  hipEventRecord(e0);
  const int reps = 50;
  for (int w = 0; w < reps; ++w) k_probe<NACC, MODE><<<blocks, 256>>>(out, iters, in, gbig, gbig_f4);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  ms /= reps;
  double fl = (double)blocks * 4 * iters * 16 * 4096.0;
  printf("%-34s blocks=%5d  %8.3f ms  %7.1f TFLOP/s\n", name, blocks, ms, fl / ms / 1e9);
}
int main() {
  float *out, *in; hipMalloc(&out, 8192 * 256 * 4); hipMalloc(&in, 8192 * 4);
  std::vector<float> h(8192); for (int i = 0; i < 8192; ++i) h[i] = (float)((i * 2654435761u) >> 8) / 16777216.f - 0.5f;
  hipMemcpy(in, h.data(), 8192 * 4, hipMemcpyHostToDevice);
  float* gbig; size_t nb = (size_t)64 << 20; hipMalloc(&gbig, nb); hipMemset(gbig, 0, nb);
  for (int blocks : {512, 1024}) {
    run<2, 0>("reg  2 acc", blocks, out, in);
    run<2, 2>("lds+barrier 2 acc", blocks, out, in);
    run<2, 3>("stage 1 f4/16mfma", blocks, out, in, gbig, nb / 16);
    run<2, 4>("stage 2 f4/16mfma", blocks, out, in, gbig, nb / 16);
    run<2, 5>("stage 3 f4/16mfma", blocks, out, in, gbig, nb / 16);
    run<2, 6>("stage 4 f4/16mfma", blocks, out, in, gbig, nb / 16);
    run<2, 8>("stage 6 f4/16mfma", blocks, out, in, gbig, nb / 16);
  }
  return 0;
}
