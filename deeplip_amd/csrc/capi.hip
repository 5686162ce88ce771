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
