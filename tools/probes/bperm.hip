#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* out) {
  const int lane = threadIdx.x & 63;
  float v = lane * 10.f;
  const float got = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(((lane - 16) & 63) * 4, __builtin_bit_cast(int, v)));
  out[threadIdx.x] = (int)got;
}
int main() {
  int* o; (void)hipMalloc(&o, 512 * 4); k<<<1, 128>>>(o); int h[128]; (void)hipMemcpy(h, o, 512, hipMemcpyDeviceToHost);
  for (int i = 0; i < 128; i += 8) printf("%d:%d ", i, h[i]); printf("\n"); return 0;
}
