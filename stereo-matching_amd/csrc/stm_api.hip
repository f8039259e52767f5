// Error reporting and version entry points of the C ABI (include/stereomatch_hip.h).
#include <cstdarg>
#include <cstdio>

#include "stm_common.h"

namespace stm {

static thread_local char g_last_error[512] = "";

int fail(int code, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_last_error, sizeof(g_last_error), fmt, ap);
  va_end(ap);
  return code;
}

// Launch errors only: hipGetLastError does not synchronise, so an asynchronous
// fault surfaces on the caller's next synchronising call, as with any
// stream-ordered API.
int check_launch(const char *what) {
  hipError_t err = hipGetLastError();
  if (err != hipSuccess)
    return fail(STM_EHIP, "%s: HIP launch error: %s", what, hipGetErrorString(err));
  return STM_OK;
}

}  // namespace stm

extern "C" int stm_abi_version(void) { return STM_ABI_VERSION; }
extern "C" const char *stm_last_error(void) { return stm::g_last_error; }
