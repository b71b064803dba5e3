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

namespace {
// Spins for `ticks` ticks of the 100 MHz constant counter and reports how many
// shader-clock cycles passed: out = {core cycles, 100 MHz ticks}.
__global__ void clock_probe_kernel(int64_t* out, int ticks) {
  const unsigned long long w0 = wall_clock64(), c0 = clock64();
  unsigned long long w = w0;
  while (w - w0 < static_cast<unsigned long long>(ticks)) {
    __builtin_amdgcn_s_sleep(8);
    w = wall_clock64();
  }
  const unsigned long long c1 = clock64();
  out[0] = static_cast<int64_t>(c1 - c0);
  out[1] = static_cast<int64_t>(w - w0);
}
}  // namespace
}  // namespace epos

extern "C" int epos_clock_probe(int64_t* out2, int microseconds, void* stream) {
  EPOS_REQUIRE(out2 && microseconds > 0 && microseconds <= 100000, "bad arguments");
  hipLaunchKernelGGL(epos::clock_probe_kernel, dim3(1), dim3(64), 0,
                     static_cast<hipStream_t>(stream), out2, microseconds * 100);
  return epos::launch_status("clock_probe_kernel");
}

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
