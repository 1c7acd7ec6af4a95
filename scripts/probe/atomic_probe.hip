// Probe: cost of per-block int64 atomicAdd partials into a small accumulator array (deterministic fixed-point sums).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_at(long long* acc, int ncol, int per_block, int work) {
  // some preceding work so that blocks do not arrive in lockstep
  float v = threadIdx.x;
  for (int i = 0; i < work; ++i) v = v * 1.0001f + 0.5f;
  if ((int)threadIdx.x < per_block) {
    const int col = (blockIdx.x * 7 + threadIdx.x) % ncol;
    atomicAdd((unsigned long long*)&acc[col], (unsigned long long)(long long)(v * 16.f));
  }
}
int main() {
  long long* acc; hipMalloc(&acc, 8192 * 8); hipMemset(acc, 0, 8192 * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int work : {0, 2000}) for (int ncol : {128, 256, 1024}) for (int blocks : {512, 2048}) for (int per : {128, 256}) {
    for (int w = 0; w < 3; ++w) k_at<<<blocks, 256>>>(acc, ncol, per, work);
    hipEventRecord(e0);
    for (int w = 0; w < 20; ++w) k_at<<<blocks, 256>>>(acc, ncol, per, work);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("work=%4d ncol=%4d blocks=%4d atomics/block=%3d -> %.2f us per launch (%d atomics, %d per address)\n", work, ncol, blocks, per,
           ms / 20 * 1e3, blocks * per, blocks * per / ncol);
  }
  return 0;
}
