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
int dlip_dbg_value[DLIP_DBG_COUNT] = {-1, -1, -1, -1, -1, -1};

extern "C" int dlip_debug_set(int32_t key, int32_t value) {
  DLIP_CHECK_ARG(key >= 0 && key < DLIP_DBG_COUNT);
  dlip_dbg_value[key] = value;
  return DLIP_OK;
}

namespace {
int32_t* g_status_words = nullptr;
}

extern "C" int32_t* dlip_status_words(void) { return g_status_words; }

extern "C" int dlip_set_status_words(int32_t* words) {
  g_status_words = words;
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
