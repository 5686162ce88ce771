// v_permlane16_swap_b32 on gfx950: what __builtin_amdgcn_permlane16_swap(a, b, false, false) returns, lane by lane.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/permlane16_swap.hip -o /tmp/pswap && /tmp/pswap
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
  const unsigned l = threadIdx.x;
  unsigned x[4], y[4];
  for (int c = 0; c < 4; ++c) { x[c] = 1000 * (c + 1) + l; y[c] = 5000 + 1000 * (c + 1) + l; }
  for (int c = 0; c < 4; ++c) {
    const auto r = __builtin_amdgcn_permlane16_swap(x[c], y[c], false, false);
    out[(2 * c) * 64 + l] = r[0];
    out[(2 * c + 1) * 64 + l] = r[1];
  }
}
int main() {
  unsigned* d; hipMalloc(&d, 8 * 64 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  unsigned h[8 * 64]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  for (int v = 0; v < 8; ++v) { printf("%s[c=%d] rows:", v & 1 ? "r1" : "r0", v / 2); for (int r = 0; r < 4; ++r) printf(" %u..%u", h[v * 64 + 16 * r], h[v * 64 + 16 * r + 15]); printf("\n"); }
  return 0;
}
