// Probe (gfx950): cycles per v_mfma_f32_16x16x32_f16 in the ring kernel's issue pattern -- 3 groups of 16 MFMAs (4 x 4 accumulators,
// A fragment per row block, B fragment per column block), one wave per SIMD (256-thread workgroup, one per CU) or two (512).
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_rate.hip -o /tmp/mfma_rate && /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int THREADS, int MODE>
__global__ __launch_bounds__(THREADS) void probe(float* out, unsigned long long* cyc, int iters) {
  f16x8 a[4], ah[4], b[4], bl[4];
  for (int i = 0; i < 4; ++i)
    for (int e = 0; e < 8; ++e) {
      a[i][e] = (_Float16)(0.001f * (threadIdx.x + i + e)); ah[i][e] = (_Float16)(0.002f * (threadIdx.x + 2 * i + e));
      b[i][e] = (_Float16)(0.003f * (threadIdx.x + 3 * i + e)); bl[i][e] = (_Float16)(0.004f * (threadIdx.x + i + 2 * e));
    }
  f32x4 acc[4][4];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 4 || MODE == 5) {          // accumulator-major: the three products of one accumulator back to back (4: per (i, j); 5: pairs of rows)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[j], a[i], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[j], ah[i], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl[j], ah[i], acc[i][j], 0, 0, 0);
        }
      __builtin_amdgcn_sched_barrier(0);
      continue;
    }
#pragma unroll
    for (int g = 0; g < 3; ++g) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (MODE == 0) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(g == 2 ? bl[j] : b[j], g == 0 ? a[i] : ah[i], acc[i][j], 0, 0, 0);
          else if (MODE == 1) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[0], a[0], acc[i][j], 0, 0, 0);            // same operands, 16 accumulators
          else if (MODE == 2) acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[j], a[i], acc[0][0], 0, 0, 0);            // one accumulator chain
          else acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(g == 0 ? a[i] : ah[i], g == 2 ? bl[j] : b[j], acc[i][j], 0, 0, 0);   // operands swapped
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][3];
  out[blockIdx.x * THREADS + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (THREADS / 64) + threadIdx.x / 64] = t1 - t0;
}
template <int THREADS, int MODE>
void run(const char* what, int blocks) {
  float* d; unsigned long long* c; const int iters = 2000;
  hipMalloc(&d, blocks * THREADS * 4); hipMalloc(&c, blocks * (THREADS / 64) * 8);
  for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((probe<THREADS, MODE>), dim3(blocks), dim3(THREADS), 0, 0, d, c, iters);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(blocks * (THREADS / 64));
  hipMemcpy(h.data(), c, h.size() * 8, hipMemcpyDeviceToHost);
  std::sort(h.begin(), h.end());
  printf("%-44s %d workgroups: %.2f cycles per MFMA per wave (median wave), %.2f per SIMD\n", what, blocks, (double)h[h.size() / 2] / (iters * 48.0),
         (double)h[h.size() / 2] / (iters * 48.0) / (THREADS / 256));
}
int main() {
  run<256, 0>("kernel pattern, one wave per SIMD", 256);
  run<512, 0>("kernel pattern, two waves per SIMD", 256);
  run<256, 1>("same operands, 16 accumulators, one wave", 256);
  run<256, 2>("one accumulator chain, one wave", 256);
  run<256, 3>("operands swapped, one wave", 256);
  run<512, 3>("operands swapped, two waves", 256);
  run<256, 4>("accumulator-major (3 products chained), one wave", 256);
  run<512, 4>("accumulator-major, two waves", 256);
  return 0;
}
