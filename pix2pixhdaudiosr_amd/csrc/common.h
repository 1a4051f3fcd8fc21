// Internal helpers shared by the HIP translation units of libp2phd_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include "p2phd.h"

namespace p2phd {

void set_error(const char* fmt, ...);

void fold_launched();   // core.hip: marks the launch that just went out as the latest user of the reduction-scratch region it took
extern thread_local int g_fold_pending;   // region handed out by fold_scratch() and not yet launched on (-1: none)

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (g_fold_pending >= 0) fold_launched();
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
extern int g_opt_mdct_iters;     // tiles per workgroup of the fast forward transform (0 = heuristic)
// mdct_fast.hip: register-resident kernels for hop = n_fft/2, win = n_fft, n_fft in {1024, 2048}
bool mdct4_fast_ok(int n_fft, int hop, int win, int64_t row_len, int64_t start_pad, const void* a, const void* b);
int mdct4_fast_fwd(const float* x, int64_t B, int64_t T, int n_fft, const float* window, const float* tables,
                   int64_t start_pad, int64_t n_frames, float scale, float* out, hipStream_t st);
int imdct4_fast(const float* spec, int64_t B, int64_t n_frames, int n_fft, const float* window, const float* tables,
                int64_t crop, int64_t out_len, float scale, float* out, hipStream_t st);

// ---- fixed-order reductions across workgroups (no float atomics) -------------------------------------------------
// "Last workgroup folds": every workgroup of a group stores its partial row into a slice of a library-owned scratch
// (`sc1` write-through stores: the per-XCD L2s are not coherent with each other), takes a ticket of the group with an
// agent-scope integer atomic, and the workgroup that drew the last ticket sums the rows in index order -- the result
// does not depend on which workgroup finishes when.  The scratch is a `__device__` array of the code object (no
// allocation in any entry point); one region per kernel family, so the families may run on different streams, but two
// launches of the SAME family must be ordered: fold_scratch() remembers the region's last stream, the launch that follows
// records the region's event behind itself (check_launch -> fold_launched), and a launch arriving on another stream first
// makes that stream wait for the event -- the two launches are ordered on the device instead of mixing partials.
enum FoldRegion { FOLD_IN_BWD = 0, FOLD_COLSUM = 1, FOLD_ACT_DB = 2, FOLD_LOSS = 3, FOLD_GCONV = 4 };
struct FoldScratch { float* part; unsigned* ticket; size_t floats; int tickets; };
FoldScratch fold_scratch(int region, hipStream_t stream);   // part == nullptr: refused (error text set), see core.hip

#ifdef __HIPCC__
__device__ __forceinline__ void fold_store(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float fold_load(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// True in exactly one workgroup of the `expected` that call this on `ticket`: the one that arrived last.  All its threads
// may then fold_load() what the others fold_store()d before arriving.  Re-arms the ticket for the next launch.
__device__ __forceinline__ bool fold_arrive_last(unsigned* ticket, unsigned expected) {
  __shared__ unsigned s_fold_last;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // this wave's partial stores have left
  __syncthreads();                                              // ... and every other wave's of the workgroup
  if (threadIdx.x == 0 && threadIdx.y == 0) {
    const unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const bool last = t + 1u == expected;
    if (last) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_fold_last = last ? 1u : 0u;
  }
  __syncthreads();
  const bool last = s_fold_last != 0u;
  if (last) {                                                   // uniform over the workgroup
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");          // belt and braces beside the sc1 loads
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  return last;
}
#endif

}  // namespace p2phd

// ---- the library's 16-bit storage type ------------------------------------------------------------------------------------------
// The same sources build two libraries: libp2phd_hip.so stores activations as bf16 (the benchmarked mode, BASELINE configs[1]),
// libp2phd_hip_f16.so (-DP2PHD_F16) as IEEE fp16 -- the reference's actual AMP type (train.py:62-67: autocast + GradScaler; 11
// significand bits against 8).  Both MFMA forms issue at the same rate (MI355X_MICROARCH.md, matrix cores), accumulate in f32, and
// every kernel names the type `bf16_t` / the ABI's dtype code P2PHD_BF16 = "this library's 16-bit type" (p2phd_half_type() tells which).
#ifdef P2PHD_F16
typedef _Float16 p2phd_h16;
#else
typedef __bf16 p2phd_h16;
#endif
#ifdef __HIPCC__
typedef __attribute__((ext_vector_type(8))) p2phd_h16 p2phd_h16x8;
typedef __attribute__((ext_vector_type(16))) float p2phd_f32x16;
typedef __attribute__((ext_vector_type(4))) float p2phd_f32x4;
__device__ __forceinline__ p2phd_f32x16 p2phd_mfma_32x32x16(p2phd_h16x8 a, p2phd_h16x8 b, p2phd_f32x16 c) {
#ifdef P2PHD_F16
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
#else
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
#endif
}
__device__ __forceinline__ p2phd_f32x4 p2phd_mfma_16x16x32(p2phd_h16x8 a, p2phd_h16x8 b, p2phd_f32x4 c) {
#ifdef P2PHD_F16
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
#else
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
#endif
}
// the two 16-bit values packed in one dword, widened to f32
__device__ __forceinline__ void p2phd_unpack2(unsigned w, float& lo, float& hi) {
#ifdef P2PHD_F16
  lo = (float)__builtin_bit_cast(_Float16, (unsigned short)(w & 0xFFFFu));
  hi = (float)__builtin_bit_cast(_Float16, (unsigned short)(w >> 16));
#else
  lo = __uint_as_float(w << 16);
  hi = __uint_as_float(w & 0xFFFF0000u);
#endif
}
#endif

namespace p2phd {

inline bool is_pow2(int64_t v) { return v > 0 && (v & (v - 1)) == 0; }
inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

}  // namespace p2phd
