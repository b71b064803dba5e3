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

// Issue priority of the depthwise kernels' waves: when such a wave shares a SIMD with a GEMM
// wave of another image in flight it gets the issue slots first -- it is short and mostly
// waits for memory, the GEMM wave loses nothing it could use. Same-box A/Bs of round 5
// (profiles/r05/ab_dw_setprio.txt, ab_wave_priorities.txt): priority 1-3 vs 0 = +0.6 % end to
// end; the same for the other helper kernels (softmax, resize, correspondences) and for the
// fitting kernels: no effect, not kept. Wave priority only: nothing about results changes.
#ifndef EPOS_DW_PRIO
#define EPOS_DW_PRIO 3
#endif
#define EPOS_SET_PRIO(p) do { if ((p) > 0) __builtin_amdgcn_s_setprio(p); } while (0)

#define EPOS_REQUIRE(cond, msg)                  \
  do {                                           \
    if (!(cond)) {                               \
      ::epos::set_error("%s: %s", __func__, msg); \
      return EPOS_E_INVALID;                     \
    }                                            \
  } while (0)

}  // namespace epos
