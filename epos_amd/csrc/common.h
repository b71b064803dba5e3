// Shared host-side helpers for libepos_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/epos_hip.h"

namespace epos {

void set_error(const char* fmt, ...);

inline int check_hip(hipError_t e, const char* what) {
  if (e == hipSuccess) return EPOS_OK;
  set_error("%s: %s", what, hipGetErrorString(e));
  return EPOS_E_HIP_BASE - static_cast<int>(e);
}

inline int launch_status(const char* kernel) {
  return check_hip(hipGetLastError(), kernel);
}

inline int64_t round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }
inline int64_t ceil_div(int64_t x, int64_t m) { return (x + m - 1) / m; }

#define EPOS_REQUIRE(cond, msg)                  \
  do {                                           \
    if (!(cond)) {                               \
      ::epos::set_error("%s: %s", __func__, msg); \
      return EPOS_E_INVALID;                     \
    }                                            \
  } while (0)

}  // namespace epos
