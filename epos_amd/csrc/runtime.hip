// Error reporting and device queries of libepos_hip.so.
#include <stdarg.h>
#include <string.h>

#include "common.h"

namespace epos {
namespace {
thread_local char g_error[512] = "";
}

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_error, sizeof(g_error), fmt, ap);
  va_end(ap);
}
}  // namespace epos

extern "C" int epos_abi_version(void) { return EPOS_ABI_VERSION; }

extern "C" const char* epos_last_error(void) { return epos::g_error; }

extern "C" int epos_device_count(void) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    epos::set_error("hipGetDeviceCount: %s", hipGetErrorString(e));
    (void)hipGetLastError();
    return 0;
  }
  return n;
}
