// Shared device/host helpers for the gfx950 kernels (internal; the public ABI is include/deeplip_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "deeplip_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#define DLIP_OOB_OFFSET 0x80000000u  // byte offset beyond any 2 GiB buffer: buffer loads return 0
#define DLIP_MAX_BUFFER_BYTES 0x7FFFFFFFll

#define DLIP_CHECK_ARG(cond) \
  do {                       \
    if (!(cond)) return DLIP_EINVAL; \
  } while (0)

static inline int dlip_launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? DLIP_OK : (int)e;
}

// Raw buffer resource over [ptr, ptr+bytes): out-of-range lanes read 0 (used for conv halos, M/K tails).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t dlip_make_rsrc(const void* ptr, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(ptr), (short)0, (int)bytes, 0x00020000);
}

__device__ __forceinline__ f32x4 dlip_buffer_load_f4(__amdgpu_buffer_rsrc_t r, uint32_t byte_off) {
  u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 0);
  return __builtin_bit_cast(f32x4, v);
}

__device__ __forceinline__ float dlip_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__device__ __forceinline__ double dlip_wave_sum_f64(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
