// ABI bookkeeping entry points of libdeeplip_hip.so (see include/deeplip_hip.h).
#include "dlip_common.h"

extern "C" int dlip_abi_version(void) { return DLIP_ABI_VERSION; }

extern "C" const char* dlip_error_string(int code) {
  if (code == DLIP_OK) return "ok";
  if (code == DLIP_EINVAL) return "deeplip_hip: invalid argument (shape / pointer / alignment)";
  if (code == DLIP_ERANGE) return "deeplip_hip: tensor exceeds the 2 GiB single-launch addressing window";
  if (code > 0) return hipGetErrorString(static_cast<hipError_t>(code));
  return "deeplip_hip: unknown error code";
}

// ---- diagnostic overrides and range status (include/deeplip_hip.h) ----
int dlip_dbg_value[DLIP_DBG_COUNT] = {-1, -1, -1, -1, -1, -1, -1, -1, -1, -1};

extern "C" int dlip_debug_set(int32_t key, int32_t value) {
  DLIP_CHECK_ARG(key >= 0 && key < DLIP_DBG_COUNT);
#ifndef DLIP_LAB
  // key 7 (the rows kernel's general mode: built, measured slower than the ring kernel on the trunk, round 4) exists in the lab
  // library only; resetting it (-1 / 0) is always fine
  DLIP_CHECK_ARG(!(key == DLIP_DBG_ROWS2D && value > 0));
#endif
  dlip_dbg_value[key] = value;
  return DLIP_OK;
}

namespace {
int32_t* g_status_words = nullptr;
// dlip_status_scope: the block the launches of THIS host thread report to instead of the process-wide one (thread-local like the
// range scope and stream capture: a step plan is recorded by one thread, and the words its launches were handed stay baked into it)
thread_local int32_t* g_status_scope = nullptr;
inline int32_t* status_now() { return g_status_scope != nullptr ? g_status_scope : g_status_words; }
}

extern "C" int32_t* dlip_status_words(void) { return status_now(); }

// ---- low-side range scope (include/deeplip_hip.h: dlip_range_scope_begin / _end) ----
// Thread-local like stream capture: a scope belongs to the host thread that issues the launches of one forward pass.
namespace {
struct RangeScope {
  int32_t* slots = nullptr;
  int n = 0, cur = 0, depth = 0;
};
thread_local RangeScope g_scope;

// One thread per launch of the scope, over its DLIP_EVID_LINES lines: flag 1 ("something, all below 2^-2") anywhere without flag 0
// ("something >= 2^-2") anywhere = the launch's whole tensor lay in (0, 2^-2) -> its kernel family + 1 into the host-pinned
// word DLIP_ST_LOW (sticky until the host clears it); raised flags are re-zeroed for the scope's next use.
__global__ __launch_bounds__(64) void range_verdict_kernel(int32_t* slots, int n, int32_t* status) {
  // one WAVE per launch: lane 2 l + f looks at flag f of line l (a single thread walking the 64 words took 20 us)
  const int i = blockIdx.x;
  if (i >= n) return;
  static_assert(DLIP_EVID_LINES == 32, "one lane per (line, flag)");
  int32_t* w = slots + (size_t)i * DLIP_EVID_WORDS + (threadIdx.x >> 1) * 32 + (threadIdx.x & 1);
  const int32_t v = *w;
  if (v != 0) *w = 0;
  const unsigned long long up = __builtin_amdgcn_ballot_w64(v != 0);
  const bool big = (up & 0x5555555555555555ull) != 0ull, small = (up & 0xAAAAAAAAAAAAAAAAull) != 0ull;
  if (small && !big && status != nullptr) {
    const int lane = __builtin_ctzll(up & 0xAAAAAAAAAAAAAAAAull);       // a lane that holds the family code
    const int32_t code = __shfl(v, lane, 64);
    if (threadIdx.x == 0) __hip_atomic_store(status + DLIP_ST_LOW, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
}  // namespace

DlipRange dlip_range_for(int family) {
  DlipRange r;
  if (status_now() != nullptr) r.status = status_now() + family;
  RangeScope& sc = g_scope;
  if (sc.slots != nullptr) {
    if (sc.cur < sc.n) r.lo = sc.slots + (size_t)DLIP_EVID_WORDS * sc.cur;
    ++sc.cur;            // (counted beyond n as well: a scope that ran out of slots must not end quietly, dlip_range_scope_end)
  }
  r.code = family + 1;
  return r;
}

extern "C" int dlip_range_scope_begin(int32_t* slots, int32_t n) {
  DLIP_CHECK_ARG(slots && n > 0);
  RangeScope& sc = g_scope;
  if (sc.depth++ == 0) {
    sc.slots = slots;
    sc.n = n;
    sc.cur = 0;
  }
  return DLIP_OK;
}

extern "C" int dlip_range_scope_end(dlip_stream_t stream) {
  RangeScope& sc = g_scope;
  DLIP_CHECK_ARG(sc.depth > 0);
  if (--sc.depth > 0) return DLIP_OK;          // an inner scope: the outermost one owns the verdict
  int32_t* slots = sc.slots;
  const int used = sc.cur < sc.n ? sc.cur : sc.n;
  const bool over = sc.cur > sc.n;     // more split-producing launches than the scope has slots: the later ones went unguarded
  sc = RangeScope{};
  if (used > 0) {
    hipLaunchKernelGGL(range_verdict_kernel, dim3((unsigned)used), dim3(64), 0, static_cast<hipStream_t>(stream), slots, used, status_now());
    const int e = dlip_launch_status();
    if (e != DLIP_OK) return e;
  }
  return over ? DLIP_ERANGE : DLIP_OK;
}

extern "C" int dlip_set_status_words(int32_t* words) {
  g_status_words = words;
  return DLIP_OK;
}

extern "C" int dlip_status_scope(int32_t* words) {
  g_status_scope = words;
  return DLIP_OK;
}

// ---- span scope: in-kernel timing of the LDS-DMA convolution launches (include/deeplip_hip.h) ----
namespace {
struct SpanScope {
  unsigned long long* pairs = nullptr;
  unsigned long long* acc = nullptr;
  int n = 0, cur = 0;
};
thread_local SpanScope g_span;

// record i = {min start, -, .., 8 x max end by workgroup shard} of one launch in 100 MHz ticks -> acc[2 i] += max end - start,
// acc[2 i + 1] += 1; the record is re-armed
__global__ __launch_bounds__(256) void span_collect_kernel(unsigned long long* pairs, unsigned long long* acc, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  unsigned long long* r = pairs + (size_t)DLIP_SPAN_WORDS * i;
  const unsigned long long t0 = r[0];
  unsigned long long t1 = 0ull;
#pragma unroll
  for (int k = 0; k < 8; ++k) { t1 = r[8 + k] > t1 ? r[8 + k] : t1; r[8 + k] = 0ull; }
  if (t1 != 0ull && t1 >= t0) { acc[2 * i] += t1 - t0; acc[2 * i + 1] += 1ull; }
  r[0] = ~0ull;
}
}  // namespace

unsigned long long* dlip_span_next(void) {
  SpanScope& sc = g_span;
  if (sc.pairs == nullptr || sc.cur >= sc.n) return nullptr;
  return sc.pairs + (size_t)DLIP_SPAN_WORDS * (sc.cur++);
}

extern "C" int dlip_span_scope_begin(uint64_t* pairs, uint64_t* acc, int32_t n) {
  DLIP_CHECK_ARG(pairs && acc && n > 0 && g_span.pairs == nullptr);
  g_span.pairs = reinterpret_cast<unsigned long long*>(pairs);
  g_span.acc = reinterpret_cast<unsigned long long*>(acc);
  g_span.n = n;
  g_span.cur = 0;
  return DLIP_OK;
}

extern "C" int dlip_span_scope_end(dlip_stream_t stream, int32_t* used) {
  SpanScope sc = g_span;
  g_span = SpanScope{};
  DLIP_CHECK_ARG(sc.pairs != nullptr);
  if (used) *used = sc.cur;
  if (sc.cur > 0) {
    hipLaunchKernelGGL(span_collect_kernel, dim3((unsigned)((sc.cur + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), sc.pairs, sc.acc,
                       sc.cur);
    return dlip_launch_status();
  }
  return DLIP_OK;
}

// ---- device identity (cached per device) ----
extern "C" int dlip_device_is_gfx950(void) {
  static std::mutex mu;
  static signed char known[DLIP_MAX_DEVICES] = {};   // 0 unknown, 1 yes, -1 no
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= DLIP_MAX_DEVICES) return 0;
  std::lock_guard<std::mutex> lock(mu);
  if (known[dev] == 0) {
    hipDeviceProp_t prop;
    known[dev] = -1;
    if (hipGetDeviceProperties(&prop, dev) == hipSuccess) {
      const char* n = prop.gcnArchName;     // "gfx950:sramecc+:xnack-"
      if (n[0] == 'g' && n[1] == 'f' && n[2] == 'x' && n[3] == '9' && n[4] == '5' && n[5] == '0' && (n[6] == 0 || n[6] == ':')) known[dev] = 1;
    }
  }
  return known[dev] > 0;
}
