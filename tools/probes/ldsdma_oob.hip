#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k(const float* x, unsigned bytes, float* out) {
  __shared__ __attribute__((aligned(16))) float s[512];
  for (int i = threadIdx.x; i < 512; i += 64) s[i] = -7.f;
  __syncthreads();
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, bytes, 0x00020000);
  unsigned off = (threadIdx.x & 1) ? 0x80000000u : threadIdx.x * 16u;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)s, 16, (int)off, 0, 0, 0);
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  for (int i = threadIdx.x; i < 512; i += 64) out[i] = s[i];
}
int main() {
  float *x, *o; hipMalloc(&x, 4096); hipMalloc(&o, 2048);
  float h[1024]; for (int i = 0; i < 1024; ++i) h[i] = i + 1; hipMemcpy(x, h, 4096, hipMemcpyHostToDevice);
  k<<<1, 64>>>(x, 4096, o); float r[512]; hipMemcpy(r, o, 2048, hipMemcpyDeviceToHost);
  for (int i = 0; i < 32; ++i) printf("%g ", r[i]); printf("\n");
  return 0;
}
