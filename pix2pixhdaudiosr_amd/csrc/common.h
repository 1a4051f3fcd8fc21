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

// tuning override (p2phd_set_option "mdct_generic"): 1 = always the generic LDS kernels of mdct.hip
extern int g_opt_mdct_generic;
// mdct_fast.hip: register-resident kernels for hop = n_fft/2, win = n_fft, n_fft in {1024, 2048}
bool mdct4_fast_ok(int n_fft, int hop, int win, int64_t row_len, int64_t start_pad, const void* a, const void* b);
int mdct4_fast_fwd(const float* x, int64_t B, int64_t T, int n_fft, const float* window, const float* tables,
                   int64_t start_pad, int64_t n_frames, float scale, float* out, hipStream_t st);
int imdct4_fast(const float* spec, int64_t B, int64_t n_frames, int n_fft, const float* window, const float* tables,
                int64_t crop, int64_t out_len, float scale, float* out, hipStream_t st);

inline bool is_pow2(int64_t v) { return v > 0 && (v & (v - 1)) == 0; }
inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

}  // namespace p2phd
