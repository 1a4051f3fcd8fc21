// Internal helpers shared by the HIP translation units of libp2phd_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include "p2phd.h"

namespace p2phd {

void set_error(const char* fmt, ...);

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return P2PHD_ELAUNCH;
  }
  return P2PHD_OK;
}

#define P2PHD_REQUIRE(cond, ...)            \
  do {                                      \
    if (!(cond)) {                          \
      p2phd::set_error(__VA_ARGS__);        \
      return P2PHD_EINVAL;                  \
    }                                       \
  } while (0)

inline bool is_pow2(int64_t v) { return v > 0 && (v & (v - 1)) == 0; }
inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

}  // namespace p2phd
