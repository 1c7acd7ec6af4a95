#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const void* p, unsigned bytes, float* out, unsigned oob) {
  __shared__ __attribute__((aligned(16))) unsigned smem[64 * 4 * 2];
  for (int i = threadIdx.x; i < 512; i += 64) smem[i] = 0xdeadbeefu;
  __syncthreads();
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
  unsigned off = threadIdx.x * 16;
  if (threadIdx.x & 1) off = oob;      // odd lanes out of range
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)smem, 16, off, 0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 256; i += 64) out[i] = __uint_as_float(smem[i]);
}
int main() {
  float* in; float* out;
  hipMalloc(&in, 4096); hipMalloc(&out, 1024);
  std::vector<float> h(1024);
  for (int i = 0; i < 1024; ++i) h[i] = 1.0f + i;
  hipMemcpy(in, h.data(), 4096, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, (const void*)in, 4096u, out, 0x80000000u);
  std::vector<float> o(256);
  hipMemcpy(o.data(), out, 1024, hipMemcpyDeviceToHost);
  for (int l = 0; l < 6; ++l) { unsigned u; std::memcpy(&u, &o[4 * l], 4); printf("lane %d: %g %g %g %g (raw0 %08x)\n", l, o[4*l], o[4*l+1], o[4*l+2], o[4*l+3], u); }
  return 0;
}
