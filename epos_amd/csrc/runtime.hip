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

// A HIP stream whose kernels run on a subset of the CUs (hipExtStreamCreateWithCUMask). Bit
// layout on gfx950 (probed: tools/cu_mask/cu_mask_probe.hip, profiles/r05/cu_mask_probe.txt):
// bit i = CU number i / 8 of XCD i % 8 (consecutive CU numbers go round the XCD's shader
// engines); workgroup b still goes to XCD b % 8, so a partition must own CUs in EVERY XCD --
// an XCD whose share of the mask is empty runs unmasked. hipGraph launches into such a stream
// honour the mask.
extern "C" int epos_stream_create_cu_mask(const uint32_t* mask, int words, void** stream) {
  EPOS_REQUIRE(mask && stream && words > 0 && words <= 32, "bad arguments");
  for (int x = 0; x < 8; ++x) {           // every XCD needs at least one CU
    bool any = false;
    for (int i = x; i < words * 32; i += 8) any = any || ((mask[i >> 5] >> (i & 31)) & 1u);
    EPOS_REQUIRE(any, "the mask leaves an XCD (bits = x mod 8) without a CU: it would run unmasked");
  }
  hipStream_t s = nullptr;
  const int rc = epos::check_hip(hipExtStreamCreateWithCUMask(&s, static_cast<uint32_t>(words), mask),
                                 "hipExtStreamCreateWithCUMask");
  if (rc) return rc;
  *stream = s;
  return EPOS_OK;
}

extern "C" int epos_stream_destroy(void* stream) {
  return epos::check_hip(hipStreamDestroy(static_cast<hipStream_t>(stream)), "hipStreamDestroy");
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
