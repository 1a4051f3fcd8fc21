// Evaluation metrics of the generation tail: compute_matrics (util/util.py:133-184, caller generate_audio.py:47-49).
//
//   sr' = (sr - mean(sr)) / std(sr) * std(hr) + mean(hr)     per row, unbiased std            (:139-140)
//   mse = mean((sr' - hr)^2);  snr_x = mean_rows 10 log10(sum hr^2 / sum (x - hr)^2)            (:148-152)
//   lsd = mean_{rows,frames} sqrt(mean_bins (log10(P_hr + 1e-6) - log10(P_sr' + 1e-6))^2)       (:178-182)
// with P = |STFT|^2 of n_fft2 = 2 n_fft, hop2 = 2 hop, a KBD window of 2 win, reflect-centred frames.
//
// Five launches, no host synchronisation, deterministic (per-block partials in fp64, summed by one block):
//   moments -> row constants -> match (+ error partials) -> STFT/LSD -> finalize.
// The STFT transforms hr and sr' together: z = hr + i sr' through ONE n_fft2-point complex Stockham FFT in LDS run by
// the whole workgroup, separated by H[k] = (Z[k] + conj Z[-k]) / 2, S[k] = (Z[k] - conj Z[-k]) / 2i.
#include "common.h"
#include "fft_wave.h"
#include <cmath>

namespace {
using namespace p2phd_fft;

constexpr int kThreads = 256;
constexpr int kMaxChunks = 64;

__device__ __forceinline__ double block_sum(double v, double* s_red) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) s_red[wave] = v;
  __syncthreads();
  return s_red[0] + s_red[1] + s_red[2] + s_red[3];
}

// P1[b][chunk][5] = sum sr, sum sr^2, sum hr, sum hr^2, sum (lr - hr)^2
__global__ __launch_bounds__(kThreads) void moments_kernel(const float* __restrict__ hr, const float* __restrict__ lr,
                                                           const float* __restrict__ sr, long T, int chunks, double* __restrict__ P1) {
  __shared__ double s_red[4];
  const long b = blockIdx.y;
  const long per = (T + chunks - 1) / chunks;
  const long lo = blockIdx.x * per, hi = min(T, lo + per);
  double a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0;
  for (long i = lo + threadIdx.x; i < hi; i += kThreads) {
    const float s = sr[b * T + i], h = hr[b * T + i], d = lr[b * T + i] - h;
    a0 += s; a1 += (double)s * s; a2 += h; a3 += (double)h * h; a4 += (double)d * d;
  }
  double* o = P1 + (b * chunks + blockIdx.x) * 5;
  a0 = block_sum(a0, s_red); a1 = block_sum(a1, s_red); a2 = block_sum(a2, s_red);
  a3 = block_sum(a3, s_red); a4 = block_sum(a4, s_red);
  if (threadIdx.x == 0) { o[0] = a0; o[1] = a1; o[2] = a2; o[3] = a3; o[4] = a4; }
}

// ROW[b][6] = mean_sr, std_sr, mean_hr, std_hr, sum hr^2, sum (lr - hr)^2
__global__ __launch_bounds__(64) void row_consts_kernel(const double* __restrict__ P1, long B, long T, int chunks, double* __restrict__ ROW) {
  const long b = (long)blockIdx.x * 64 + threadIdx.x;
  if (b >= B) return;
  double a[5] = {0, 0, 0, 0, 0};
  for (int c = 0; c < chunks; ++c)
    for (int j = 0; j < 5; ++j) a[j] += P1[(b * chunks + c) * 5 + j];
  const double n = (double)T;
  const double ms = a[0] / n, mh = a[2] / n;
  const double vs = (a[1] - n * ms * ms) / (n - 1.0), vh = (a[3] - n * mh * mh) / (n - 1.0);
  double* o = ROW + b * 6;
  o[0] = ms; o[1] = sqrt(fmax(vs, 0.0)); o[2] = mh; o[3] = sqrt(fmax(vh, 0.0)); o[4] = a[3]; o[5] = a[4];
}

__global__ __launch_bounds__(kThreads) void match_kernel(const float* __restrict__ hr, const float* __restrict__ sr,
                                                         const double* __restrict__ ROW, long T, int chunks,
                                                         float* __restrict__ out, double* __restrict__ P3) {
  __shared__ double s_red[4];
  const long b = blockIdx.y;
  const float ms = (float)ROW[b * 6 + 0], ss = (float)ROW[b * 6 + 1], mh = (float)ROW[b * 6 + 2], sh = (float)ROW[b * 6 + 3];
  const long per = (T + chunks - 1) / chunks;
  const long lo = blockIdx.x * per, hi = min(T, lo + per);
  double acc = 0;
  for (long i = lo + threadIdx.x; i < hi; i += kThreads) {
    const float v = (sr[b * T + i] - ms) / ss * sh + mh;
    out[b * T + i] = v;
    const float d = v - hr[b * T + i];
    acc += (double)d * d;
  }
  acc = block_sum(acc, s_red);
  if (threadIdx.x == 0) P3[b * chunks + blockIdx.x] = acc;
}

__global__ __launch_bounds__(kThreads) void lsd_kernel(const float* __restrict__ hr, const float* __restrict__ sr, long T, int n2,
                                                       int hop2, int win2, const float* __restrict__ window2,
                                                       const float* __restrict__ tables, int pad, long frames, int fchunks,
                                                       double* __restrict__ P4) {
  extern __shared__ float4 smem_raw[];
  __shared__ double s_red[4];
  float2* buf0 = reinterpret_cast<float2*>(smem_raw);
  float2* buf1 = buf0 + n2;
  float2* s_tw = buf1 + n2;
  float* s_win = reinterpret_cast<float*>(s_tw + n2);
  const int tid = threadIdx.x;
  const long b = blockIdx.y;
  const int left = (n2 - win2) >> 1;
  for (int i = tid; i < n2; i += kThreads) {
    s_tw[i] = reinterpret_cast<const float2*>(tables)[i];
    s_win[i] = (i >= left && i < left + win2) ? window2[i - left] : 0.f;
  }
  const long per = (frames + fchunks - 1) / fchunks;
  const long f_lo = blockIdx.x * per, f_hi = min(frames, f_lo + per);
  const float* h = hr + b * T;
  const float* s = sr + b * T;
  const int nbins = (n2 >> 1) + 1;
  double total = 0;
  __syncthreads();
  for (long f = f_lo; f < f_hi; ++f) {
    for (int n = tid; n < n2; n += kThreads) {
      long i = f * hop2 - pad + n;
      if (i < 0) i = -i;
      if (i >= T) i = 2 * (T - 1) - i;
      const float w = s_win[n];
      buf0[n] = make_float2(w * h[i], w * s[i]);
    }
    __syncthreads();
    const float2* Z = fft_coop(buf0, buf1, s_tw, n2, tid, kThreads);
    double acc = 0;
    for (int k = tid; k < nbins; k += kThreads) {
      const float2 a = Z[k], m = Z[(n2 - k) & (n2 - 1)];
      const float hx = 0.5f * (a.x + m.x), hy = 0.5f * (a.y - m.y);
      const float sx = 0.5f * (a.y + m.y), sy = 0.5f * (m.x - a.x);
      const float d = log10f(hx * hx + hy * hy + 1e-6f) - log10f(sx * sx + sy * sy + 1e-6f);
      acc += (double)d * d;
    }
    acc = block_sum(acc, s_red);           // also fences Z before the next frame overwrites buf0
    total += sqrt(acc / nbins);
    __syncthreads();
  }
  if (tid == 0) P4[b * fchunks + blockIdx.x] = total;
}

__global__ __launch_bounds__(kThreads) void finalize_kernel(const double* __restrict__ ROW, const double* __restrict__ P3,
                                                            const double* __restrict__ P4, long B, long T, int chunks, long frames,
                                                            int fchunks, float* __restrict__ result) {
  __shared__ double s_red[4];
  double err_all = 0, snr_s = 0, snr_l = 0, lsd = 0;
  for (long b = threadIdx.x; b < B; b += kThreads) {
    double e = 0;
    for (int c = 0; c < chunks; ++c) e += P3[b * chunks + c];
    err_all += e;
    snr_s += 10.0 * log10(ROW[b * 6 + 4] / e);
    snr_l += 10.0 * log10(ROW[b * 6 + 4] / ROW[b * 6 + 5]);
    for (int c = 0; c < fchunks; ++c) lsd += P4[b * fchunks + c];
  }
  err_all = block_sum(err_all, s_red); snr_s = block_sum(snr_s, s_red);
  snr_l = block_sum(snr_l, s_red); lsd = block_sum(lsd, s_red);
  if (threadIdx.x == 0) {
    result[0] = (float)(err_all / ((double)B * (double)T));
    result[1] = (float)(snr_s / (double)B);
    result[2] = (float)(snr_l / (double)B);
    result[3] = (float)(lsd / ((double)B * (double)frames));
  }
}

struct Plan { int chunks, fchunks; long frames; int pad; };

int make_plan(int64_t B, int64_t T, int n2, int hop2, int win2, int center, Plan* p) {
  P2PHD_REQUIRE(p2phd::is_pow2(n2) && n2 >= 16 && n2 <= 4096, "metrics: STFT length must be a power of two in [16, 4096], got %d", n2);
  P2PHD_REQUIRE(win2 >= 1 && win2 <= n2 && hop2 >= 1, "metrics: bad STFT window %d / hop %d for n_fft %d", win2, hop2, n2);
  P2PHD_REQUIRE(B >= 1 && T >= 2, "metrics: need at least one row of two samples");
  p->pad = center ? n2 / 2 : 0;
  P2PHD_REQUIRE(!center || T > p->pad, "metrics: reflect padding %d needs a longer signal than %lld", p->pad, (long long)T);
  P2PHD_REQUIRE(T + 2 * p->pad >= n2, "metrics: signal of %lld samples shorter than one STFT frame (%d)", (long long)T, n2);
  p->frames = 1 + (T + 2 * p->pad - n2) / hop2;
  p->chunks = (int)std::min<int64_t>(kMaxChunks, p2phd::cdiv(T, 4096));
  int64_t fc = std::max<int64_t>(1, std::min<int64_t>(p->frames, 2048 / std::max<int64_t>(B, 1)));
  p->fchunks = (int)std::min<int64_t>(fc, 1024);
  P2PHD_REQUIRE(B < 65536, "metrics: too many rows");
  return P2PHD_OK;
}

}  // namespace

extern "C" size_t p2phd_stft_tables_floats(int n_fft2) { return 2 * (size_t)n_fft2; }

extern "C" int p2phd_stft_tables_fill(int n_fft2, float* host_out) {
  P2PHD_REQUIRE(p2phd::is_pow2(n_fft2) && n_fft2 >= 16 && n_fft2 <= 4096, "stft tables: bad n_fft %d", n_fft2);
  P2PHD_REQUIRE(host_out != nullptr, "stft tables: null output");
  const double pi = 3.14159265358979323846264338327950288;
  for (int j = 0; j < n_fft2; ++j) {
    const double a = -2.0 * pi * j / n_fft2;
    host_out[2 * j] = (float)std::cos(a);
    host_out[2 * j + 1] = (float)std::sin(a);
  }
  return P2PHD_OK;
}

extern "C" size_t p2phd_metrics_workspace_bytes(int64_t B, int64_t T, int n_fft2, int hop2, int win2, int center) {
  Plan p;
  if (make_plan(B, T, n_fft2, hop2, win2, center, &p) != P2PHD_OK) return 0;
  return sizeof(double) * (size_t)B * ((size_t)p.chunks * 6 + 6 + (size_t)p.fchunks);
}

extern "C" int p2phd_audio_metrics(const float* hr, const float* lr, const float* sr, int64_t B, int64_t T, int n_fft2, int hop2,
                                   int win2, const float* window2, const float* tables, int center, float* sr_matched,
                                   float* result4, void* workspace, void* stream) {
  Plan p;
  if (int rc = make_plan(B, T, n_fft2, hop2, win2, center, &p)) return rc;
  P2PHD_REQUIRE(hr && lr && sr && window2 && tables && sr_matched && result4 && workspace, "audio_metrics: null pointer");
  hipStream_t st = (hipStream_t)stream;
  double* P1 = reinterpret_cast<double*>(workspace);
  double* ROW = P1 + (size_t)B * p.chunks * 5;
  double* P3 = ROW + (size_t)B * 6;
  double* P4 = P3 + (size_t)B * p.chunks;
  hipLaunchKernelGGL(moments_kernel, dim3(p.chunks, (unsigned)B), dim3(kThreads), 0, st, hr, lr, sr, (long)T, p.chunks, P1);
  hipLaunchKernelGGL(row_consts_kernel, dim3((unsigned)p2phd::cdiv(B, 64)), dim3(64), 0, st, P1, (long)B, (long)T, p.chunks, ROW);
  hipLaunchKernelGGL(match_kernel, dim3(p.chunks, (unsigned)B), dim3(kThreads), 0, st, hr, sr, ROW, (long)T, p.chunks, sr_matched, P3);
  const size_t lds = (size_t)n_fft2 * (3 * sizeof(float2) + sizeof(float));
  if (lds > 48 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(lsd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(lsd_kernel, dim3(p.fchunks, (unsigned)B), dim3(kThreads), lds, st, hr, sr_matched, (long)T, n_fft2, hop2, win2,
                     window2, tables, p.pad, p.frames, p.fchunks, P4);
  hipLaunchKernelGGL(finalize_kernel, dim3(1), dim3(kThreads), 0, st, ROW, P3, P4, (long)B, (long)T, p.chunks, p.frames, p.fchunks, result4);
  return p2phd::check_launch("audio_metrics");
}
