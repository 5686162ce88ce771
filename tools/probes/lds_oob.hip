// Probe (gfx950): what does a ds_read_b128 return at an address beyond the workgroup's LDS allocation?
// Many workgroups per CU fill their own allocation with a pattern; every lane then reads at in-range and out-of-range
// addresses.  Prints the number of out-of-range reads that were not all-zero.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/lds_oob.hip -o /tmp/lds_oob && /tmp/lds_oob
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__global__ void probe(unsigned* out, int alloc_bytes) {
  extern __shared__ __attribute__((aligned(16))) unsigned lds[];
  for (int i = threadIdx.x; i < alloc_bytes / 4; i += blockDim.x) lds[i] = 0xABCD0000u + blockIdx.x;
  __syncthreads();
  const unsigned base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds;
  const unsigned offs[6] = {0u, (unsigned)alloc_bytes, (unsigned)alloc_bytes + 4096u, 163840u, 1u << 18, (1u << 18) + 40000u};
  unsigned bad = 0, inr = 0;
  for (int k = 0; k < 6; ++k) {
    const unsigned addr = base + offs[k] + threadIdx.x * 16;
    u32x4 v;
    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    const unsigned any = v[0] | v[1] | v[2] | v[3];
    if (k == 0) inr = any; else if (any) bad |= 1u << k;
  }
  out[(blockIdx.x * blockDim.x + threadIdx.x) * 2] = inr;
  out[(blockIdx.x * blockDim.x + threadIdx.x) * 2 + 1] = bad;
}
int main() {
  const int blocks = 2048, threads = 256, alloc = 8192;   // 8 KB each: many workgroups share a CU's LDS
  unsigned* d; hipMalloc(&d, blocks * threads * 2 * 4);
  hipLaunchKernelGGL(probe, dim3(blocks), dim3(threads), alloc, 0, d, alloc);
  std::vector<unsigned> h(blocks * threads * 2);
  hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
  long bad[8] = {0}, in_ok = 0;
  for (size_t i = 0; i < h.size(); i += 2) { in_ok += (h[i] >> 16) == 0xABCD; for (int k = 1; k < 6; ++k) bad[k] += (h[i + 1] >> k) & 1; }
  printf("in-range reads with the pattern: %ld of %d\n", in_ok, blocks * threads);
  const char* names[6] = {"", "alloc", "alloc+4K", "160K", "256K", "256K+40000"};
  for (int k = 1; k < 6; ++k) printf("offset %-11s: %ld lanes read non-zero\n", names[k], bad[k]);
  return 0;
}
