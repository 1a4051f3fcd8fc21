// MDCT2 / IMDCT2 = framing + window + DCT-II / DCT-III as the reference's DCT_2N_native / IDCT_2N_native define them
// (models/mdct.py:352-454, dct/dct_native.py:7-68; the vendored CUDA counterparts are dct/src/dct_2N_cuda.cpp and
// dct_cuda_kernel.cu:267-406: pad -> rFFT(2N) -> twiddle -> truncate, four launches plus cuFFT).
//
//   forward  X[k] = scale * c_k * (2/N) * sum_i u[i] cos(pi (2i+1) k / 2N),  c_0 = k0_scale, c_k = 1
//   inverse  y[i] = scale * (k0_scale * X[0] + 2 * sum_{k>=1} X[k] cos(pi (2i+1) k / 2N))
// (k0_scale = 1 reproduces DCT_2N_native / IDCT_2N_native, for which idct(dct(a)) = 2a -- test/DCT_test.ipynb cell 34;
//  other values give the exact autograd adjoints.)
//
// One workgroup owns a tile of frames of one batch row, exactly like mdct.hip: the signal segment is staged once in
// LDS, each wavefront runs Makhoul's even/odd reordering + one N-point Stockham FFT in its own LDS buffers + one
// rotation by exp(-i pi k / 2N); no zero-padded 2N buffer, no separate reorder / twiddle / truncate passes.  The
// inverse overlap-adds by gather from an LDS ring of time-domain frames.  n_fft a power of two in [16, 2048] (at 2048
// two of the four wavefronts run the FFTs: their buffers are what fits beside the frame ring in 160 KiB).
#include "common.h"
#include "fft_wave.h"
#include <cmath>

namespace {
using namespace p2phd_fft;

constexpr int kThreads = 256;
constexpr int kWaves = 4;

__global__ __launch_bounds__(kThreads) void mdct2_fwd_kernel(
    const float* __restrict__ x, long T, int N, int hop, int win, const float* __restrict__ window,
    const float* __restrict__ tables, long start_pad, long F, float scale, float k0_scale, float* __restrict__ out,
    int f_tile, int n_tiles, int seg_cap, int win_cap, int fft_waves) {
  extern __shared__ float4 smem_raw[];
  float* smem = reinterpret_cast<float*>(smem_raw);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long b = blockIdx.x / n_tiles;
  const long t0 = (long)(blockIdx.x % n_tiles) * f_tile;
  const int nf = (int)min((long)f_tile, F - t0);

  float* s_sig = smem;
  float* s_win = s_sig + seg_cap;
  float2* s_ftw = reinterpret_cast<float2*>(s_win + win_cap);   // exp(-2 pi i j / N)
  float2* s_rot = s_ftw + N;                                    // exp(-i pi k / 2N)
  float2* buf0 = s_rot + N + (size_t)wave * 2 * N;
  float2* buf1 = buf0 + N;

  const int seg = (nf - 1) * hop + win;
  const long p0 = t0 * hop - start_pad;
  const float* xb = x + b * T;
  for (int i = tid; i < seg; i += kThreads) {
    const long idx = p0 + i;
    s_sig[i] = (idx >= 0 && idx < T) ? xb[idx] : 0.f;
  }
  for (int i = tid; i < win; i += kThreads) s_win[i] = window[i];
  const float2* tb = reinterpret_cast<const float2*>(tables);
  for (int i = tid; i < 2 * N; i += kThreads) s_ftw[i] = tb[i];
  __syncthreads();

  const float nrm = scale * 2.f / (float)N;
  for (int f0 = 0; f0 < nf; f0 += fft_waves) {                  // fft_waves < 4 at n_fft 2048: idle waves only keep the barriers
    const int f = f0 + wave;
    const bool active = wave < fft_waves && f < nf;
    if (active) {
      const float* u = s_sig + f * hop;
      auto U = [&](int n) -> float { return n < win ? u[n] * s_win[n] : 0.f; };
      for (int n = lane; n < (N >> 1); n += 64) {               // Makhoul: v[n] = u[2n], v[N-1-n] = u[2n+1]
        buf0[n] = make_float2(U(2 * n), 0.f);
        buf0[N - 1 - n] = make_float2(U(2 * n + 1), 0.f);
      }
    }
    __syncthreads();
    float2* res = fft_wave(buf0, buf1, s_ftw, N, lane, active);
    float* stage = reinterpret_cast<float*>(res == buf0 ? buf1 : buf0);
    if (active) {
      for (int k = lane; k < N; k += 64) {
        const float2 c = cmul(res[k], s_rot[k]);
        stage[k] = c.x * nrm * (k == 0 ? k0_scale : 1.f);
      }
    }
    __syncthreads();
    if (active) {
      float* o = out + ((b * F + t0 + f) * (long)N);
      const float4* s4 = reinterpret_cast<const float4*>(stage);
      float4* o4 = reinterpret_cast<float4*>(o);
      for (int i = lane; i < (N >> 2); i += 64) o4[i] = s4[i];
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(kThreads) void imdct2_fwd_kernel(
    const float* __restrict__ spec, long F, int N, int hop, int win, const float* __restrict__ window,
    const float* __restrict__ tables, long crop, long out_len, float scale, float k0_scale, float* __restrict__ out,
    int ts, int n_tiles, int fr_cap, int win_cap, int fft_waves) {
  extern __shared__ float4 smem_raw[];
  float* smem = reinterpret_cast<float*>(smem_raw);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long b = blockIdx.x / n_tiles;
  const long m0 = (long)(blockIdx.x % n_tiles) * ts;
  const int tile_len = (int)min((long)ts, out_len - m0);

  float* s_d = smem;                                         // [fr_cap][win] time-domain frames
  float* s_win = s_d + (size_t)fr_cap * win;
  float2* s_ftw = reinterpret_cast<float2*>(s_win + win_cap);
  float2* s_rot = s_ftw + N;
  float2* buf0 = s_rot + N + (size_t)wave * 2 * N;
  float2* buf1 = buf0 + N;

  for (int i = tid; i < win; i += kThreads) s_win[i] = window[i];
  const float2* tb = reinterpret_cast<const float2*>(tables);
  for (int i = tid; i < 2 * N; i += kThreads) s_ftw[i] = tb[i];

  const long a_lo = m0 + crop - win + 1;
  const long t_lo = a_lo > 0 ? (a_lo + hop - 1) / hop : 0;
  long t_hi = (m0 + tile_len - 1 + crop) / hop;
  if (t_hi > F - 1) t_hi = F - 1;
  const int nfr = (int)(t_hi - t_lo + 1);
  __syncthreads();

  for (int f0 = 0; f0 < nfr; f0 += fft_waves) {
    const int f = f0 + wave;
    const bool active = wave < fft_waves && f < nfr;
    float* X = reinterpret_cast<float*>(buf1);
    if (active) {
      const float* row = spec + ((b * F + t_lo + f) * (long)N);
      for (int i = lane; i < N; i += 64) X[i] = row[i];
    }
    __syncthreads();
    if (active) {
      // conj(V_k) = (X'_k + i X'_{N-k}) * exp(-i pi k / 2N), X'_0 = k0 * X_0, X'_N = 0;  v = Re(FFT(conj V)) = N * ifft(V)
      for (int k = lane; k < N; k += 64) {
        const float a = k == 0 ? k0_scale * X[0] : X[k];
        const float c = k == 0 ? 0.f : X[N - k];
        buf0[k] = cmul(make_float2(a, c), s_rot[k]);
      }
    }
    __syncthreads();
    float2* res = fft_wave(buf0, buf1, s_ftw, N, lane, active);
    if (active) {
      float* d = s_d + (size_t)f * win;
      for (int n = lane; n < (N >> 1); n += 64) {
        const int i0 = 2 * n, i1 = 2 * n + 1;
        if (i0 < win) d[i0] = res[n].x;
        if (i1 < win) d[i1] = res[N - 1 - n].x;
      }
    }
    __syncthreads();
  }

  float* ob = out + b * out_len + m0;
  for (int m = tid; m < tile_len; m += kThreads) {
    const long pm = m0 + m + crop;
    const long a = pm - win + 1;
    long ts0 = a > 0 ? (a + hop - 1) / hop : 0;
    if (ts0 < t_lo) ts0 = t_lo;
    long ts1 = pm / hop;
    if (ts1 > t_hi) ts1 = t_hi;
    float acc = 0.f;
    for (long t = ts0; t <= ts1; ++t) {
      const int q = (int)(pm - t * hop);
      acc += s_win[q] * s_d[(size_t)(t - t_lo) * win + q];
    }
    ob[m] = scale * acc;
  }
}

int frames_per_tile(int N) { return N <= 256 ? 8 : (N <= 512 ? 4 : 2); }
int fft_waves_for(int N) { return N <= 1024 ? kWaves : 2; }      // 2 x (2 buffers of N float2) = 64 KiB at N = 2048

int check_common(int n_fft, int hop, int win) {
  P2PHD_REQUIRE(p2phd::is_pow2(n_fft) && n_fft >= 16 && n_fft <= 2048,
                "mdct2: n_fft must be a power of two in [16, 2048], got %d", n_fft);
  P2PHD_REQUIRE(win >= 1 && win <= n_fft, "mdct2: window length %d should be no more than fft length %d", win, n_fft);
  P2PHD_REQUIRE(hop >= 1 && hop <= win, "mdct2: hop %d exceeds the window (%d): you hopped more than one frame", hop, win);
  return P2PHD_OK;
}

}  // namespace

extern "C" size_t p2phd_dct_tables_floats(int n_fft) { return 4 * (size_t)n_fft; }

extern "C" int p2phd_dct_tables_fill(int n_fft, float* host_out) {
  P2PHD_REQUIRE(p2phd::is_pow2(n_fft) && n_fft >= 16 && n_fft <= 2048, "dct tables: bad n_fft %d", n_fft);
  P2PHD_REQUIRE(host_out != nullptr, "dct tables: null output");
  const double pi = 3.14159265358979323846264338327950288;
  for (int j = 0; j < n_fft; ++j) {
    const double a = -2.0 * pi * j / n_fft;
    host_out[2 * j] = (float)std::cos(a);
    host_out[2 * j + 1] = (float)std::sin(a);
  }
  for (int k = 0; k < n_fft; ++k) {
    const double a = -pi * k / (2.0 * n_fft);
    host_out[2 * n_fft + 2 * k] = (float)std::cos(a);
    host_out[2 * n_fft + 2 * k + 1] = (float)std::sin(a);
  }
  return P2PHD_OK;
}

extern "C" int p2phd_mdct2_fwd(const float* x, int64_t B, int64_t T, int n_fft, int hop, int win, const float* window,
                               const float* tables, int64_t start_pad, int64_t n_frames, float scale, float k0_scale,
                               float* out, void* stream) {
  if (int rc = check_common(n_fft, hop, win)) return rc;
  P2PHD_REQUIRE(B >= 0 && T >= 0 && n_frames >= 0 && start_pad >= 0, "mdct2_fwd: negative size");
  if (B == 0 || n_frames == 0) return P2PHD_OK;
  P2PHD_REQUIRE(x && window && tables && out, "mdct2_fwd: null pointer");
  const int f_tile = frames_per_tile(n_fft);
  const int64_t n_tiles = p2phd::cdiv(n_frames, f_tile);
  P2PHD_REQUIRE(B * n_tiles < (1ll << 31), "mdct2_fwd: grid too large");
  const int seg_cap = (((f_tile - 1) * hop + win) + 3) & ~3;
  const int win_cap = (win + 3) & ~3;
  const int fw = fft_waves_for(n_fft);
  const size_t lds = sizeof(float) * ((size_t)seg_cap + win_cap + 4 * n_fft + (size_t)fw * 4 * n_fft);
  P2PHD_REQUIRE(lds <= 160 * 1024, "mdct2_fwd: hop %d / window %d need %zu B of LDS", hop, win, lds);
  if (lds > 48 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(mdct2_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(mdct2_fwd_kernel, dim3((unsigned)(B * n_tiles)), dim3(kThreads), lds, (hipStream_t)stream, x, (long)T, n_fft, hop,
                     win, window, tables, (long)start_pad, (long)n_frames, scale, k0_scale, out, f_tile, (int)n_tiles, seg_cap, win_cap, fw);
  return p2phd::check_launch("mdct2_fwd");
}

extern "C" int p2phd_imdct2_fwd(const float* spec, int64_t B, int64_t n_frames, int n_fft, int hop, int win, const float* window,
                                const float* tables, int64_t crop_start, int64_t out_len, float scale, float k0_scale,
                                float* out, void* stream) {
  if (int rc = check_common(n_fft, hop, win)) return rc;
  P2PHD_REQUIRE(B >= 0 && n_frames >= 0 && out_len >= 0 && crop_start >= 0, "imdct2_fwd: negative size");
  if (B == 0 || out_len == 0) return P2PHD_OK;
  P2PHD_REQUIRE(window && tables && out && (spec || n_frames == 0), "imdct2_fwd: null pointer");
  const int f_tile = frames_per_tile(n_fft);
  const int ts = f_tile * hop;
  const int64_t n_tiles = p2phd::cdiv(out_len, ts);
  P2PHD_REQUIRE(B * n_tiles < (1ll << 31), "imdct2_fwd: grid too large");
  const int fr_cap = f_tile + (win - 1) / hop + 1;
  const int win_cap = (win + 3) & ~3;
  const int fw = fft_waves_for(n_fft);
  const size_t lds = sizeof(float) * ((size_t)fr_cap * win + win_cap + 4 * n_fft + (size_t)fw * 4 * n_fft);
  P2PHD_REQUIRE(lds <= 160 * 1024, "imdct2_fwd: hop %d too small for n_fft %d (LDS frame ring %zu B)", hop, n_fft, lds);
  if (lds > 48 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(imdct2_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(imdct2_fwd_kernel, dim3((unsigned)(B * n_tiles)), dim3(kThreads), lds, (hipStream_t)stream, spec, (long)n_frames,
                     n_fft, hop, win, window, tables, (long)crop_start, (long)out_len, scale, k0_scale, out, ts, (int)n_tiles, fr_cap, win_cap, fw);
  return p2phd::check_launch("imdct2_fwd");
}
