// ABI bookkeeping of libscae_hip.so (see include/scae_hip.h).
#include "common.h"

extern "C" int scae_abi_version(void) { return SCAE_ABI_VERSION; }

extern "C" const char *scae_error_string(int code) {
  if (code == SCAE_OK) return "ok";
  if (code == SCAE_ERR_BAD_ARG) return "scae: null pointer or non-positive size";
  if (code == SCAE_ERR_UNSUPPORTED) return "scae: shape outside this build's kernel limits";
  if (code > 0) return hipGetErrorString((hipError_t)code);
  return "scae: unknown error";
}
