// Shared device/host helpers for the gfx950 kernels (internal; the public ABI is include/deeplip_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <mutex>

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
enum { DLIP_DBG_CONV_TILE = 0, DLIP_DBG_DMA_TILE = 1, DLIP_DBG_DMA_ENABLE = 2, DLIP_DBG_STREAMK = 3, DLIP_DBG_WIN = 4, DLIP_DBG_NINNER = 5, DLIP_DBG_ROWS = 6, DLIP_DBG_ROWS2D = 7, DLIP_DBG_BN_FUSED = 8, DLIP_DBG_ROWS_TAIL = 9, DLIP_DBG_COUNT = 10 };
extern "C" __attribute__((visibility("hidden"))) int dlip_dbg_value[DLIP_DBG_COUNT];

// Range-status words (dlip_set_status_words): where the f16x3 kernels report an activation the split format
// cannot hold.  NULL = not registered (nothing is reported).  One word per kernel family.
enum { DLIP_ST_CONV = 0, DLIP_ST_STEM = 1, DLIP_ST_PACK = 2, DLIP_ST_POOL = 3, DLIP_ST_LOW = 4, DLIP_ST_COUNT = 8 };
extern "C" __attribute__((visibility("hidden"))) int32_t* dlip_status_words(void);
#define DLIP_F16_OVERFLOW 65520.0f   // the smallest magnitude that rounds to infinity in fp16
// Low side of the split format.  hi keeps 11 significant bits down to 6.1e-5, but lo = v - hi is ~2^-11 |v|: it is a NORMAL
// fp16 number only while |v| >= 2^-3, below that a subnormal on the fixed grid 2^-24 -- an absolute error of up to 2^-25 = 3e-8
// per element, whatever the element's size.  What that costs a contraction depends on the SCALE of the tensor: measured on a
// 3-layer 3x3 chain of Gaussian activations (tests/test_range_gpu.py) the result is 5.9e-7 off at sigma = 1, 2.8e-5 at sigma =
// 1e-3 (largest element 4.5e-3) and 3.1e-2 at sigma = 1e-6, i.e. 1e-5 is crossed at a largest element of ~1.2e-2.  A produced
// tensor whose largest magnitude lies in (0, DLIP_SPLIT_LOW) is therefore REPORTED like an overflow (an all-zero tensor is
// exact and is not).  The line: 2^-2 -- 3e-8 / 0.25 = 1.2e-7 of the tensor's scale per layer, fp32's own grade.  It was 2^-6 until
// round 6 (1.9e-6 of the scale per layer): a seven-layer speech encoder whose tensors all sat at 2^-3 (a model calibrated on a hot
// batch, then fed an ordinary one: tests/test_arith_gpu.py, the soak test) came out 2.1e-6 of the row's scale off in one element
// of 2048 -- inside the old line, outside the repo's bar (1e-4 |b| + 1e-6 max|b|).  The shipped models' tensors sit at 2^0 .. 2^4.
#define DLIP_SPLIT_LOW 0.25f       // 2^-2

// What a producer of split-format values is handed per launch: the host-pinned overflow word of its kernel family and the
// pair of device words that collects the launch's low-side evidence (dlip_range_scope_*; NULL outside a scope: low side unguarded).
struct DlipRange {
  int32_t* status = nullptr;
  int32_t* lo = nullptr;
  int32_t code = 0;        // family + 1: the value of a raised evidence flag (names the kernel in the error)
};
__attribute__((visibility("hidden"))) DlipRange dlip_range_for(int family);

// Called by every wave at the end of a producer of split-format values: amax = the largest |v| the lane converted.
//   high side: any lane >= 65520 -> 1 into the family's host-pinned word (system scope);
//   low side : the launch owns DLIP_EVID_LINES cache lines of evidence; in each, word 0 = "a wave saw |v| >= 2^-2", word 1 = "a wave
//              saw 0 < |v| < 2^-2 and nothing larger"; the scope's verdict kernel (capi.hip) ORs the lines and reports the launch
//              iff flag 1 is set and flag 0 is not, i.e. iff the whole tensor's largest magnitude lies in (0, 2^-2).
//              FLAGS, not counters, and SPREAD: a wave picks its line by workgroup and wave index, looks at the flag with a plain
//              (L1-cacheable: a stale 0 only costs a redundant store) load and stores the family code only while it reads 0.
//              The first version OR-ed bits atomically into ONE word per launch: the 16 k waves of an element-wise pass (or
//              the first round of a GEMM) finishing together queued at that word's L2 channel -- 12 ns per atomic, and
//              plain loads of one hot line were no better: the 22 us stem pre-pass became a 100 us one.
#define DLIP_EVID_LINES 32
#define DLIP_EVID_WORDS (DLIP_EVID_LINES * 32)   // int32 words of evidence per launch (32 lines of 128 B)
__device__ __forceinline__ void dlip_report_range(float amax, const DlipRange r) {
  if (r.status != nullptr && __builtin_amdgcn_ballot_w64(!(amax < DLIP_F16_OVERFLOW)) != 0ull) {
    if ((threadIdx.x & 63) == 0) __hip_atomic_store(r.status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  if (r.lo != nullptr) {
    const bool big = __builtin_amdgcn_ballot_w64(amax >= DLIP_SPLIT_LOW) != 0ull;
    const bool some = __builtin_amdgcn_ballot_w64(amax > 0.f) != 0ull;
    if ((threadIdx.x & 63) == 0 && some) {
      const unsigned line = (blockIdx.x * 5u + (threadIdx.x >> 6)) & (DLIP_EVID_LINES - 1);
      volatile int32_t* w = r.lo + line * 32 + (big ? 0 : 1);
      if (*w == 0) *w = r.code;
    }
  }
}

// The same report for kernels whose every thread reaches the call (element-wise passes with thousands of small workgroups): the
// waves of a workgroup meet in LDS first, so ONE lane per workgroup touches the evidence lines instead of one per wave.
__device__ __forceinline__ void dlip_report_range_block(float amax, const DlipRange r) {
  if (r.status != nullptr && __builtin_amdgcn_ballot_w64(!(amax < DLIP_F16_OVERFLOW)) != 0ull) {
    if ((threadIdx.x & 63) == 0) __hip_atomic_store(r.status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  if (r.lo == nullptr) return;      // (kernel-uniform)
  __shared__ int wg_big, wg_some;
  if (threadIdx.x == 0) { wg_big = 0; wg_some = 0; }
  __syncthreads();
  const bool big = __builtin_amdgcn_ballot_w64(amax >= DLIP_SPLIT_LOW) != 0ull;
  const bool some = __builtin_amdgcn_ballot_w64(amax > 0.f) != 0ull;
  if ((threadIdx.x & 63) == 0) {
    if (big) wg_big = 1;
    else if (some) wg_some = 1;
  }
  __syncthreads();
  if (threadIdx.x == 0 && (wg_big | wg_some)) {
    volatile int32_t* w = r.lo + (blockIdx.x & (DLIP_EVID_LINES - 1)) * 32 + (wg_big ? 0 : 1);
    if (*w == 0) *w = r.code;
  }
}

// Pixel arithmetic of the clip ingest (models/video_models/dataloaders.py:11-22 "val" pipeline: Normalize(0,255) ->
// CenterCrop -> Normalize(0.421, 0.165); BT.601 gray of preprocess.py:44 kept in float).  ONE definition for every kernel
// that turns uint8 frames into normalised pixels (ingest_rgb_kernel, crop_norm_kernel, the stem's uint8 pre-pass), with
// contraction off: mul, add, IEEE divide in exactly the oracle's order (numpy float32, oracle/deeplip_oracle.py
// ingest_rgb_u8 / video_preprocess_u8), so the kernels agree with each other AND with the oracle bit for bit.
__device__ __forceinline__ float dlip_gray601(float r, float g, float b) {
#pragma clang fp contract(off)
  return (0.299f * r + 0.587f * g) + 0.114f * b;
}
__device__ __forceinline__ float dlip_pixel_norm(float gray_0_255) {
#pragma clang fp contract(off)
  return (gray_0_255 / 255.0f - 0.421f) / 0.165f;
}

// Ragged batches (zero-padded [B, Tmax, ...] + a length vector, as models/video_models/dataset.py:123-139 collates them): how many of
// group g's `cap` rows are valid = clamp(len[g] * mul + add, 0, cap); len == NULL: all of them.  `mul` / `add` turn the caller's unit
// (frames of the INPUT clip / utterance) into rows of the pooled tensor: lip clips mul = Ho Wo of the last convolution, utterances
// add = -(frames the valid convolutions consume).
struct DlipLen {
  const int32_t* len = nullptr;
  int32_t mul = 1, add = 0;
};
__device__ __forceinline__ int dlip_valid_rows(const DlipLen l, long long g, int cap) {
  if (l.len == nullptr) return cap;
  const long long v = (long long)l.len[g] * l.mul + l.add;
  return v < 0 ? 0 : (v > cap ? cap : (int)v);
}

// Launch-side state that HIP keeps PER DEVICE: the MaxDynamicSharedMemorySize attribute of a kernel and how many of its
// workgroups the device holds at once.  One instance per kernel instantiation (a function-local static); a process that
// drives several GPUs (one thread per device, or hipSetDevice in a loop) gets the attribute set and the grid sized on each.
#define DLIP_MAX_DEVICES 32
struct DlipKernelState {
  std::mutex mu;
  size_t lds_set[DLIP_MAX_DEVICES] = {};
  int slots[DLIP_MAX_DEVICES] = {};
  // raise the dynamic-LDS limit of `kern` on the current device to `bytes` (once per device and size)
  int ensure_lds(const void* kern, size_t bytes) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= DLIP_MAX_DEVICES) return DLIP_EINVAL;
    std::lock_guard<std::mutex> lock(mu);
    if (bytes > lds_set[dev]) {
      hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
      if (e != hipSuccess) return (int)e;
      lds_set[dev] = bytes;
    }
    return DLIP_OK;
  }
  // CUs x resident workgroups per CU of `kern` on the current device (persistent grids)
  int resident(const void* kern, int threads, size_t lds, int* out) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= DLIP_MAX_DEVICES) return DLIP_EINVAL;
    std::lock_guard<std::mutex> lock(mu);
    if (slots[dev] == 0) {
      int cus = 0, per_cu = 0;
      if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
          hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, threads, lds) != hipSuccess || per_cu <= 0)
        return DLIP_EINVAL;
      slots[dev] = cus * per_cu;
    }
    *out = slots[dev];
    return DLIP_OK;
  }
};
// Next record of the open span scope (capi.hip: dlip_span_scope_*), or NULL: in-kernel timing of replayed launches.  A record is
// DLIP_SPAN_WORDS uint64 (one 128-byte line): word 0 = start (min), words 8..15 = end (max) per workgroup shard.
#define DLIP_SPAN_WORDS 16
__attribute__((visibility("hidden"))) unsigned long long* dlip_span_next(void);
// Measurement only; one scalar test when no scope is open.  `wg` = the workgroup's index in whatever order the kernel walks.
// Entry: every 16th workgroup folds the clock into the start (they start within half a microsecond of each other; 500 workgroups
// at one word would add microseconds to the launch they time).  Exit: EVERY workgroup folds it into the end word of its shard --
// the last one out is what ends a launch, and under a second stream's load it is not a predictable one (a sampled end read up to
// 10 % short of the kernel trace) -- behind a drain of its own stores.
__device__ __forceinline__ void dlip_span_enter(unsigned long long* span, int wg) {
  if (span != nullptr && threadIdx.x == 0 && (wg & 15) == 0) atomicMin(span, (unsigned long long)__builtin_amdgcn_s_memrealtime());
}
__device__ __forceinline__ void dlip_span_exit(unsigned long long* span) {
  if (span != nullptr && threadIdx.x == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's stores have left
    atomicMax(span + 8 + (blockIdx.x & 7), (unsigned long long)__builtin_amdgcn_s_memrealtime());
  }
}
// 1 when the current device is gfx950 (cached per device; capi.hip).  Kernels that lean on probed gfx950 behaviour (conv_win's
// out-of-allocation ds_read returning zeros) are only selected there.
extern "C" __attribute__((visibility("hidden"))) int dlip_device_is_gfx950(void);

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
