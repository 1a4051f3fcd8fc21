// Wavefront-cooperative Stockham FFT in LDS shared by the MDCT4 (mdct.hip) and DCT-II/III (dct.hip) kernels.
#pragma once
#include <hip/hip_runtime.h>

namespace p2phd_fft {

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
  return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }

// Stockham autosort FFT of Q points (Q a power of two >= 4) owned by ONE wavefront; every wave of
// the workgroup calls it in lockstep (the barriers are workgroup barriers).  tw[j] = exp(-2 pi i j / Q).
// Returns the buffer holding the result.
__device__ float2* fft_wave(float2* a, float2* b, const float2* __restrict__ tw, int Q, int lane, bool active) {
  int ns = 1;
  while (ns < Q) {
    if (ns * 4 <= Q) {
      const int t = Q >> 2;
      const int step = Q / (4 * ns);
      if (active) {
        for (int j = lane; j < t; j += 64) {
          const int k = j & (ns - 1);
          float2 v0 = a[j], v1 = a[j + t], v2 = a[j + 2 * t], v3 = a[j + 3 * t];
          if (ns > 1) {
            v1 = cmul(v1, tw[k * step]);
            v2 = cmul(v2, tw[2 * k * step]);
            v3 = cmul(v3, tw[3 * k * step]);
          }
          const float2 A = cadd(v0, v2), B = csub(v0, v2), C = cadd(v1, v3);
          const float2 d = csub(v1, v3);
          const float2 D = make_float2(d.y, -d.x);  // -i * (v1 - v3)
          const int j0 = ((j - k) << 2) + k;
          b[j0] = cadd(A, C);
          b[j0 + ns] = cadd(B, D);
          b[j0 + 2 * ns] = csub(A, C);
          b[j0 + 3 * ns] = csub(B, D);
        }
      }
      ns <<= 2;
    } else {
      const int t = Q >> 1;
      const int step = Q / (2 * ns);
      if (active) {
        for (int j = lane; j < t; j += 64) {
          const int k = j & (ns - 1);
          const float2 v0 = a[j];
          const float2 v1 = cmul(a[j + t], tw[k * step]);
          const int j0 = ((j - k) << 1) + k;
          b[j0] = cadd(v0, v1);
          b[j0 + ns] = csub(v0, v1);
        }
      }
      ns <<= 1;
    }
    __syncthreads();
    float2* s = a; a = b; b = s;
  }
  return a;
}

// The same Stockham FFT run by `nthr` threads of a workgroup on one Q-point transform (metrics.hip's 2048/4096-point
// STFT frames, too large for one buffer pair per wave).  Every thread of the workgroup must call it.
__device__ float2* fft_coop(float2* a, float2* b, const float2* __restrict__ tw, int Q, int tid, int nthr) {
  int ns = 1;
  while (ns < Q) {
    if (ns * 4 <= Q) {
      const int t = Q >> 2;
      const int step = Q / (4 * ns);
      for (int j = tid; j < t; j += nthr) {
        const int k = j & (ns - 1);
        float2 v0 = a[j], v1 = a[j + t], v2 = a[j + 2 * t], v3 = a[j + 3 * t];
        if (ns > 1) {
          v1 = cmul(v1, tw[k * step]);
          v2 = cmul(v2, tw[2 * k * step]);
          v3 = cmul(v3, tw[3 * k * step]);
        }
        const float2 A = cadd(v0, v2), B = csub(v0, v2), C = cadd(v1, v3);
        const float2 d = csub(v1, v3);
        const float2 D = make_float2(d.y, -d.x);
        const int j0 = ((j - k) << 2) + k;
        b[j0] = cadd(A, C);
        b[j0 + ns] = cadd(B, D);
        b[j0 + 2 * ns] = csub(A, C);
        b[j0 + 3 * ns] = csub(B, D);
      }
      ns <<= 2;
    } else {
      const int t = Q >> 1;
      const int step = Q / (2 * ns);
      for (int j = tid; j < t; j += nthr) {
        const int k = j & (ns - 1);
        const float2 v0 = a[j];
        const float2 v1 = cmul(a[j + t], tw[k * step]);
        const int j0 = ((j - k) << 1) + k;
        b[j0] = cadd(v0, v1);
        b[j0 + ns] = csub(v0, v1);
      }
      ns <<= 1;
    }
    __syncthreads();
    float2* s = a; a = b; b = s;
  }
  return a;
}

}  // namespace p2phd_fft
