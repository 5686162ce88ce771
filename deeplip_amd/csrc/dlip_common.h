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

// Diagnostic overrides (dlip_debug_set, include/deeplip_hip.h): -1 = the built-in choice.  Plain ints read on
// the launch path -- the library itself reads no environment variable there.
enum { DLIP_DBG_CONV_TILE = 0, DLIP_DBG_DMA_TILE = 1, DLIP_DBG_DMA_ENABLE = 2, DLIP_DBG_STREAMK = 3, DLIP_DBG_WIN = 4, DLIP_DBG_NINNER = 5, DLIP_DBG_COUNT = 6 };
extern "C" __attribute__((visibility("hidden"))) int dlip_dbg_value[DLIP_DBG_COUNT];

// Range-status words (dlip_set_status_words): where the f16x3 kernels report an activation the split format
// cannot hold.  NULL = not registered (nothing is reported).  One word per kernel family.
enum { DLIP_ST_CONV = 0, DLIP_ST_STEM = 1, DLIP_ST_PACK = 2, DLIP_ST_POOL = 3, DLIP_ST_COUNT = 4 };
extern "C" __attribute__((visibility("hidden"))) int32_t* dlip_status_words(void);
#define DLIP_F16_OVERFLOW 65520.0f   // the smallest magnitude that rounds to infinity in fp16

// Called once per wave at the end of a producer of split-format values: amax = the largest |v| the lane converted.
__device__ __forceinline__ void dlip_report_range(float amax, int32_t* status_word) {
  if (status_word != nullptr && __builtin_amdgcn_ballot_w64(!(amax < DLIP_F16_OVERFLOW)) != 0ull) {
    if ((threadIdx.x & 63) == 0) __hip_atomic_store(status_word, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

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
