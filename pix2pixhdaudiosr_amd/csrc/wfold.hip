// W-fold helpers for the tiny-channel convolutions (the 2-channel 7x7 input / output layers of the generator,
// networks.py:190,207, and the 1-channel 4x4 head of the discriminator, :361).
//
// An implicit GEMM whose channel dimension is 1-4 wastes most of every MFMA tile (channel pitch 8, N tile >= 32).
// For those layers the horizontal kernel taps are folded into the channel axis, turning an R x S convolution into
// an R x 1 one over S*C (input fold) or S*K (output fold) channels:
//   input fold  : Xe[n,h,w,(tw,c)]   = x[n, h, w + tw - pad, c]          conv' = R x 1 over S*C channels
//   output fold : Y [n,h,w',(tw,k)]  = sum_{th,c} x[n,h+th-pad,w'-pad,c] w[k,c,th,tw]   (one GEMM with N = S*K)
//                 y [n,h,w,k]        = act(bias_k + sum_tw Y[n,h,w+tw,(tw,k)])            (hsum_kernel)
//   and for the backward of an output-folded layer dyE[n,h,w',(tw,k)] = dy[n,h,w'-tw,k].
// These three kernels are the cheap HBM-bound data movers around the GEMMs; tensors are NHWC with pitch 8.
#include "convplan.h"

namespace {

typedef p2phd_h16 bf16_t;                 // the library's 16-bit storage type: bf16, or fp16 in the -DP2PHD_F16 build (common.h)
__device__ __forceinline__ float to_f(float v) { return v; }
__device__ __forceinline__ float to_f(bf16_t v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f(float v);
template <> __device__ __forceinline__ float from_f<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f<bf16_t>(float v) { return (bf16_t)v; }

__device__ __forceinline__ int reflect_idx(int i, int n) {
  if (i < 0) i = -i;
  if (i >= n) i = 2 * (n - 1) - i;
  return i;
}

// Xe[n,h,wo,(tw*C + c)] = x[n,h,map(wo + tw - pad),c]   (map = reflect or zero), wo in [0,Wo)
// One thread per 16-byte output piece; its elements are gathered into registers with compile-time indices (a per-pixel
// row[] array filled at run-time positions lives in scratch memory and made this kernel 3-4x slower than its traffic).
template <typename T>
__global__ __launch_bounds__(256) void expand_in_kernel(const T* __restrict__ x, T* __restrict__ xe, long NH, int W, int Wo, int C, int Cp,
                                                        int S, int pad, int pad_mode, int Cep) {
  constexpr int EPP = 16 / (int)sizeof(T);
  const int ppr = Cep / EPP;
  const long total = NH * Wo * ppr;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int pc = (int)(e % ppr);
    const long pix = e / ppr;
    const int wo = (int)(pix % Wo);
    const long nh = pix / Wo;
    const T* srow = x + nh * W * Cp;
    alignas(16) T vals[EPP];
#pragma unroll
    for (int j = 0; j < EPP; ++j) {
      const int q = pc * EPP + j;
      const int tw = q / C, c = q - tw * C;
      int wi = wo + tw - pad;
      if (pad_mode == 1) wi = reflect_idx(wi, W);
      const bool ok = tw < S && wi >= 0 && wi < W;
      const T v = srow[(size_t)(ok ? wi : 0) * Cp + c];          // unconditional load, masked below
      vals[j] = ok ? v : from_f<T>(0.f);
    }
    *reinterpret_cast<uint4*>(xe + e * EPP) = *reinterpret_cast<const uint4*>(vals);
  }
}

// dyE[n,h,w',(tw*K + k)] = dy[n,h,w'-tw,k] (zero outside [0,Wo)), w' in [0,Wy); one thread per 16-byte output piece
template <typename T>
__global__ __launch_bounds__(256) void expand_dy_kernel(const T* __restrict__ dy, T* __restrict__ dye, long NH, int Wo, int Wy, int K, int Kp,
                                                        int S, int Cep) {
  constexpr int EPP = 16 / (int)sizeof(T);
  const int ppr = Cep / EPP;
  const long total = NH * Wy * ppr;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int pc = (int)(e % ppr);
    const long pix = e / ppr;
    const int wy = (int)(pix % Wy);
    const long nh = pix / Wy;
    const T* srow = dy + nh * Wo * Kp;
    alignas(16) T vals[EPP];
#pragma unroll
    for (int j = 0; j < EPP; ++j) {
      const int q = pc * EPP + j;
      const int tw = q / K, k = q - tw * K;
      const int w = wy - tw;
      const bool ok = tw < S && w >= 0 && w < Wo;
      const T v = srow[(size_t)(ok ? w : 0) * Kp + k];
      vals[j] = ok ? v : from_f<T>(0.f);
    }
    *reinterpret_cast<uint4*>(dye + e * EPP) = *reinterpret_cast<const uint4*>(vals);
  }
}

// y[n,h,w,k] = act(bias_k + sum_tw Y[n,h,w+tw,(tw*K+k)])
template <typename T>
__global__ __launch_bounds__(256) void hsum_kernel(const T* __restrict__ Y, const float* __restrict__ bias, T* __restrict__ y,
                                                   int H, int Wo, int Wy, int K, int Kp, int S, int Cep, int act) {
  // grid: (blocks over H*Wo pixels, N); thread -> one output pixel, all K (<= 4) channels
  const int n = blockIdx.y;
  const long HW = (long)H * Wo;
  for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < HW; p += (long)gridDim.x * 256) {
    const int w = (int)(p % Wo);
    const long h = p / Wo;
    const T* row = Y + (((size_t)n * H + h) * Wy + w) * Cep;
    alignas(16) T outv[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) outv[k] = from_f<T>(0.f);
#pragma unroll
    for (int k = 0; k < 4; ++k) {                                // K <= 4; literal k keeps outv in registers
      if (k < K) {
        float a = bias != nullptr ? bias[k] : 0.f;
        for (int tw = 0; tw < S; ++tw) a += to_f(row[(size_t)tw * Cep + tw * K + k]);
        if (act == P2PHD_ACT_TANH) a = tanhf(a);
        else if (act == P2PHD_ACT_LRELU) a = a > 0.f ? a : 0.2f * a;
        else if (act == P2PHD_ACT_RELU) a = a > 0.f ? a : 0.f;
        outv[k] = from_f<T>(a);
      }
    }
    T* o = y + ((size_t)n * HW + p) * Kp;
    if (sizeof(T) == 2) *reinterpret_cast<uint4*>(o) = *reinterpret_cast<uint4*>(outv);
    else { *reinterpret_cast<uint4*>(o) = *reinterpret_cast<uint4*>(outv); *reinterpret_cast<uint4*>(o + 4) = *reinterpret_cast<uint4*>(outv + 4); }
  }
}

inline int grid_for(long work, int cap = 8192) { return (int)std::max<long>(1, std::min<long>((work + 255) / 256, cap)); }

}  // namespace

namespace p2phd {

int launch_expand_in(int dtype, const void* x, void* xe, int N, int H, int W, int Wo, int C, int S, int pad, int pad_mode,
                     hipStream_t st) {
  const int Cp = cpitch(C), Cep = cpitch(S * C);
  const long NH = (long)N * H;
  const long total = NH * Wo * (Cep / (dtype == P2PHD_BF16 ? 8 : 4));
  P2PHD_REQUIRE(Cep <= 32 && C >= 1, "expand_in: folded channel count must be <= 32");
  if (dtype == P2PHD_BF16)
    hipLaunchKernelGGL(expand_in_kernel<bf16_t>, dim3(grid_for(total)), dim3(256), 0, st, (const bf16_t*)x, (bf16_t*)xe, NH, W, Wo, C, Cp, S, pad, pad_mode, Cep);
  else
    hipLaunchKernelGGL(expand_in_kernel<float>, dim3(grid_for(total)), dim3(256), 0, st, (const float*)x, (float*)xe, NH, W, Wo, C, Cp, S, pad, pad_mode, Cep);
  return check_launch("expand_in");
}

int launch_expand_dy(int dtype, const void* dy, void* dye, int N, int Ho, int Wo, int Wy, int K, int S, hipStream_t st) {
  const int Kp = cpitch(K), Cep = cpitch(S * K);
  const long NH = (long)N * Ho;
  const long total = NH * Wy * (Cep / (dtype == P2PHD_BF16 ? 8 : 4));
  P2PHD_REQUIRE(Cep <= 32 && K >= 1, "expand_dy: folded channel count must be <= 32");
  if (dtype == P2PHD_BF16)
    hipLaunchKernelGGL(expand_dy_kernel<bf16_t>, dim3(grid_for(total)), dim3(256), 0, st, (const bf16_t*)dy, (bf16_t*)dye, NH, Wo, Wy, K, Kp, S, Cep);
  else
    hipLaunchKernelGGL(expand_dy_kernel<float>, dim3(grid_for(total)), dim3(256), 0, st, (const float*)dy, (float*)dye, NH, Wo, Wy, K, Kp, S, Cep);
  return check_launch("expand_dy");
}

int launch_hsum(int dtype, const void* Y, const float* bias, void* y, int N, int H, int Wo, int Wy, int K, int S,
                int act, hipStream_t st) {
  P2PHD_REQUIRE(K <= 4, "hsum: at most 4 output channels");
  const int Kp = cpitch(K), Cep = cpitch(S * K);
  dim3 grid(grid_for((long)H * Wo, 1024), N);
  if (dtype == P2PHD_BF16)
    hipLaunchKernelGGL(hsum_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)Y, bias, (bf16_t*)y, H, Wo, Wy, K, Kp, S, Cep, act);
  else
    hipLaunchKernelGGL(hsum_kernel<float>, grid, dim3(256), 0, st, (const float*)Y, bias, (float*)y, H, Wo, Wy, K, Kp, S, Cep, act);
  return check_launch("hsum");
}

}  // namespace p2phd
