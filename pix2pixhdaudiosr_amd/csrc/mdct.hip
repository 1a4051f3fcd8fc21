// MDCT4 / IMDCT4 for gfx950: framed DCT-IV through an N/4-point complex FFT held in LDS.
//
// Replaces models/mdct.py:461-566 of the reference, which runs pad -> unfold -> window -> exp1 ->
// N-point complex128 FFT -> truncate -> exp2 -> real as seven full-size tensor ops.  Here one
// workgroup owns a tile of consecutive frames of one batch row:
//   * the signal segment of the tile is read from HBM once (coalesced) into LDS, so the 50 %
//     overlap between frames never re-reads HBM;
//   * each wavefront folds + windows one frame into N/4 complex points (TDAC folding), rotates
//     them (pre-twiddle), runs a radix-4/2 Stockham FFT in its own LDS ping-pong buffers,
//     rotates again (post-twiddle) and emits N/2 real bins;
//   * bins are staged in LDS and stored as whole rows (16 B per lane).
// The inverse runs the same DCT-IV core per frame into an LDS frame ring and then overlap-adds by
// GATHER (each output sample sums the <= ceil(win/hop) frames that cover it), so the reference's
// `fold` buffer [B, win, frames] is never materialised and the sum order is fixed.
//
// Algorithmic HBM bytes per frame at hop = N/2: read N/2 + write N/2 floats = 2N bytes... x4 B
// (4 KiB at N = 1024): the kernel is HBM-bound; the FFT work lives in LDS.
#include "common.h"
#include "fft_wave.h"
#include <cmath>
#include <vector>

namespace {

constexpr int kThreads = 256;
constexpr int kWaves = 4;

using namespace p2phd_fft;

struct LdsPlan {
  int seg_cap;   // floats reserved for the staged signal (fwd) / frame ring (inverse)
  int win_cap;
};

// ------------------------------------------------------------------------------------------
// forward: x[B,T] -> out[B,F,M]
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void mdct4_fwd_kernel(
    const float* __restrict__ x, long T, int N, int hop, int win, const float* __restrict__ window,
    const float* __restrict__ tables, long start_pad, long F, float scale, float* __restrict__ out,
    int f_tile, int n_tiles, int seg_cap, int win_cap) {
  extern __shared__ float4 smem_raw[];
  float* smem = reinterpret_cast<float*>(smem_raw);
  const int M = N >> 1, Q = N >> 2;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long b = blockIdx.x / n_tiles;
  const long t0 = (long)(blockIdx.x % n_tiles) * f_tile;
  const int nf = (int)min((long)f_tile, F - t0);

  float* s_sig = smem;
  float* s_win = s_sig + seg_cap;
  float2* s_ftw = reinterpret_cast<float2*>(s_win + win_cap);
  float2* s_rtw = s_ftw + Q;
  float2* buf0 = s_rtw + Q + (size_t)wave * 2 * Q;
  float2* buf1 = buf0 + Q;

  const int seg = (nf - 1) * hop + win;
  const long p0 = t0 * hop - start_pad;
  const float* xb = x + b * T;
  for (int i = tid; i < seg; i += kThreads) {
    const long idx = p0 + i;
    s_sig[i] = (idx >= 0 && idx < T) ? xb[idx] : 0.f;
  }
  for (int i = tid; i < win; i += kThreads) s_win[i] = window[i];
  const float2* tb = reinterpret_cast<const float2*>(tables);
  for (int i = tid; i < 2 * Q; i += kThreads) s_ftw[i] = tb[i];   // fft twiddles then rotation twiddles
  __syncthreads();

  for (int f0 = 0; f0 < nf; f0 += kWaves) {
    const int f = f0 + wave;
    const bool active = f < nf;
    if (active) {
      const float* u = s_sig + f * hop;
      // u[n] (n < N): windowed frame, zero beyond win
      auto U = [&](int n) -> float { return n < win ? u[n] * s_win[n] : 0.f; };
      auto V = [&](int p) -> float {   // TDAC fold N -> M
        return p < (M >> 1) ? -U(3 * (M >> 1) - 1 - p) - U(3 * (M >> 1) + p)
                            : U(p - (M >> 1)) - U(3 * (M >> 1) - 1 - p);
      };
      for (int i = lane; i < Q; i += 64) {
        const float2 z = make_float2(V(2 * i), V(M - 1 - 2 * i));
        buf0[i] = cmul(z, s_rtw[i]);
      }
    }
    __syncthreads();
    float2* res = fft_wave(buf0, buf1, s_ftw, Q, lane, active);
    float* stage = reinterpret_cast<float*>(res == buf0 ? buf1 : buf0);
    if (active) {
      for (int i = lane; i < Q; i += 64) {
        const float2 c = cmul(res[i], s_rtw[i]);
        stage[2 * i] = c.x * scale;
        stage[M - 1 - 2 * i] = -c.y * scale;
      }
    }
    __syncthreads();
    if (active) {
      float* o = out + ((b * F + t0 + f) * (long)M);
      if ((M & 3) == 0) {
        const float4* s4 = reinterpret_cast<const float4*>(stage);
        float4* o4 = reinterpret_cast<float4*>(o);
        for (int i = lane; i < (M >> 2); i += 64) o4[i] = s4[i];
      } else {
        for (int i = lane; i < M; i += 64) o[i] = stage[i];
      }
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------
// inverse with gather overlap-add: spec[B,F,M] -> out[B,out_len]
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void imdct4_fwd_kernel(
    const float* __restrict__ spec, long F, int N, int hop, int win, const float* __restrict__ window,
    const float* __restrict__ tables, long crop, long out_len, float scale, float* __restrict__ out,
    int ts, int n_tiles, int fr_cap, int win_cap) {
  extern __shared__ float4 smem_raw[];
  float* smem = reinterpret_cast<float*>(smem_raw);
  const int M = N >> 1, Q = N >> 2;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long b = blockIdx.x / n_tiles;
  const long m0 = (long)(blockIdx.x % n_tiles) * ts;
  const int tile_len = (int)min((long)ts, out_len - m0);

  float* s_d = smem;                                       // [fr_cap][M]
  float* s_win = s_d + (size_t)fr_cap * M;
  float2* s_ftw = reinterpret_cast<float2*>(s_win + win_cap);
  float2* s_rtw = s_ftw + Q;
  float2* buf0 = s_rtw + Q + (size_t)wave * 2 * Q;
  float2* buf1 = buf0 + Q;

  for (int i = tid; i < win; i += kThreads) s_win[i] = window[i];
  const float2* tb = reinterpret_cast<const float2*>(tables);
  for (int i = tid; i < 2 * Q; i += kThreads) s_ftw[i] = tb[i];

  const long a_lo = m0 + crop - win + 1;
  const long t_lo = a_lo > 0 ? (a_lo + hop - 1) / hop : 0;
  long t_hi = (m0 + tile_len - 1 + crop) / hop;
  if (t_hi > F - 1) t_hi = F - 1;
  const int nfr = (int)(t_hi - t_lo + 1);                  // may be <= 0
  __syncthreads();

  for (int f0 = 0; f0 < nfr; f0 += kWaves) {
    const int f = f0 + wave;
    const bool active = f < nfr;
    float* X = reinterpret_cast<float*>(buf1);
    if (active) {
      const float* row = spec + ((b * F + t_lo + f) * (long)M);
      for (int i = lane; i < M; i += 64) X[i] = row[i];
    }
    __syncthreads();
    if (active) {
      for (int i = lane; i < Q; i += 64) {
        const float2 z = make_float2(X[2 * i], X[M - 1 - 2 * i]);
        buf0[i] = cmul(z, s_rtw[i]);
      }
    }
    __syncthreads();
    float2* res = fft_wave(buf0, buf1, s_ftw, Q, lane, active);
    if (active) {
      float* d = s_d + (size_t)f * M;
      for (int i = lane; i < Q; i += 64) {
        const float2 c = cmul(res[i], s_rtw[i]);
        d[2 * i] = c.x;
        d[M - 1 - 2 * i] = -c.y;
      }
    }
    __syncthreads();
  }

  const int h = M >> 1;
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
      const float* d = s_d + (size_t)(t - t_lo) * M;
      const float y = q < h ? d[q + h] : (q < 3 * h ? -d[3 * h - 1 - q] : -d[q - 3 * h]);
      acc += s_win[q] * y;
    }
    ob[m] = scale * acc;
  }
}

int frames_per_tile(int N) { return N <= 1024 ? 8 : (N == 2048 ? 4 : 2); }

int check_common(int n_fft, int hop, int win) {
  P2PHD_REQUIRE(p2phd::is_pow2(n_fft) && n_fft >= 16 && n_fft <= 4096,
                "mdct4: n_fft must be a power of two in [16, 4096], got %d", n_fft);
  P2PHD_REQUIRE(win >= 1 && win <= n_fft, "mdct4: window length %d should be no more than fft length %d", win, n_fft);
  P2PHD_REQUIRE(hop >= 1 && hop <= win, "mdct4: hop %d exceeds the window (%d): you hopped more than one frame", hop, win);
  return P2PHD_OK;
}

}  // namespace

extern "C" size_t p2phd_mdct4_tables_floats(int n_fft) { return (size_t)n_fft; }

extern "C" int p2phd_mdct4_tables_fill(int n_fft, float* host_out) {
  P2PHD_REQUIRE(p2phd::is_pow2(n_fft) && n_fft >= 16 && n_fft <= 4096, "mdct4 tables: bad n_fft %d", n_fft);
  P2PHD_REQUIRE(host_out != nullptr, "mdct4 tables: null output");
  const int Q = n_fft / 4, M = n_fft / 2;
  const double pi = 3.14159265358979323846264338327950288;
  for (int j = 0; j < Q; ++j) {   // FFT twiddles exp(-2 pi i j / Q)
    const double a = -2.0 * pi * j / Q;
    host_out[2 * j] = (float)std::cos(a);
    host_out[2 * j + 1] = (float)std::sin(a);
  }
  for (int n = 0; n < Q; ++n) {   // DCT-IV rotation exp(-i pi (8n+1) / (8M)), used before AND after the FFT
    const double a = -pi * (8.0 * n + 1.0) / (8.0 * M);
    host_out[2 * Q + 2 * n] = (float)std::cos(a);
    host_out[2 * Q + 2 * n + 1] = (float)std::sin(a);
  }
  return P2PHD_OK;
}

extern "C" int p2phd_mdct4_frame_layout(int64_t dim0, int64_t T, int hop, int win, int center,
                                        int64_t* start_pad, int64_t* end_pad, int64_t* n_frames) {
  P2PHD_REQUIRE(hop >= 1 && win >= 1 && T >= 0 && dim0 >= 0, "frame_layout: bad geometry");
  // models/mdct.py:488-496 -- signal_len is len(signal), i.e. the size of dim 0
  const int64_t sp = center ? hop : 0;
  const int64_t add = dim0 % hop;
  int64_t ep = sp;
  if (add) ep = sp + hop - add;
  const int64_t padded = T + sp + ep;
  if (start_pad) *start_pad = sp;
  if (end_pad) *end_pad = ep;
  if (n_frames) *n_frames = padded >= win ? (padded - win) / hop + 1 : 0;
  return P2PHD_OK;
}

extern "C" int p2phd_mdct4_fwd(const float* x, int64_t B, int64_t T, int n_fft, int hop, int win,
                               const float* window, const float* tables, int64_t start_pad,
                               int64_t n_frames, float scale, float* out, void* stream) {
  if (int rc = check_common(n_fft, hop, win)) return rc;
  P2PHD_REQUIRE(B >= 0 && T >= 0 && n_frames >= 0 && start_pad >= 0, "mdct4_fwd: negative size");
  if (B == 0 || n_frames == 0) return P2PHD_OK;
  P2PHD_REQUIRE(x && window && tables && out, "mdct4_fwd: null pointer");
  if (p2phd::mdct4_fast_ok(n_fft, hop, win, T, start_pad, x, out))
    return p2phd::mdct4_fast_fwd(x, B, T, n_fft, window, tables, start_pad, n_frames, scale, out, (hipStream_t)stream);
  const int f_tile = frames_per_tile(n_fft);
  const int64_t n_tiles = p2phd::cdiv(n_frames, f_tile);
  P2PHD_REQUIRE(B * n_tiles < (1ll << 31), "mdct4_fwd: grid too large");
  const int Q = n_fft / 4;
  const int seg_cap = (((f_tile - 1) * hop + win) + 3) & ~3;
  const int win_cap = (win + 3) & ~3;
  const size_t lds = sizeof(float) * ((size_t)seg_cap + win_cap + 4 * Q + (size_t)kWaves * 4 * Q);
  if (lds > 48 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(mdct4_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(mdct4_fwd_kernel, dim3((unsigned)(B * n_tiles)), dim3(kThreads), lds, (hipStream_t)stream,
                     x, (long)T, n_fft, hop, win, window, tables, (long)start_pad, (long)n_frames, scale, out,
                     f_tile, (int)n_tiles, seg_cap, win_cap);
  return p2phd::check_launch("mdct4_fwd");
}

extern "C" int p2phd_imdct4_fwd(const float* spec, int64_t B, int64_t n_frames, int n_fft, int hop, int win,
                                const float* window, const float* tables, int64_t crop_start,
                                int64_t out_len, float scale, float* out, void* stream) {
  if (int rc = check_common(n_fft, hop, win)) return rc;
  P2PHD_REQUIRE(B >= 0 && n_frames >= 0 && out_len >= 0 && crop_start >= 0, "imdct4_fwd: negative size");
  if (B == 0 || out_len == 0) return P2PHD_OK;
  P2PHD_REQUIRE(window && tables && out && (spec || n_frames == 0), "imdct4_fwd: null pointer");
  if (n_frames > 0 && p2phd::mdct4_fast_ok(n_fft, hop, win, out_len, crop_start, spec, out))
    return p2phd::imdct4_fast(spec, B, n_frames, n_fft, window, tables, crop_start, out_len, scale, out, (hipStream_t)stream);
  const int f_tile = frames_per_tile(n_fft);
  const int ts = f_tile * hop;
  const int64_t n_tiles = p2phd::cdiv(out_len, ts);
  P2PHD_REQUIRE(B * n_tiles < (1ll << 31), "imdct4_fwd: grid too large");
  const int Q = n_fft / 4, M = n_fft / 2;
  const int fr_cap = f_tile + (win - 1) / hop + 1;
  const int win_cap = (win + 3) & ~3;
  const size_t lds = sizeof(float) * ((size_t)fr_cap * M + win_cap + 4 * Q + (size_t)kWaves * 4 * Q);
  P2PHD_REQUIRE(lds <= 160 * 1024, "imdct4_fwd: hop %d too small for n_fft %d (LDS frame ring %zu B)", hop, n_fft, lds);
  if (lds > 48 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(imdct4_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(imdct4_fwd_kernel, dim3((unsigned)(B * n_tiles)), dim3(kThreads), lds, (hipStream_t)stream,
                     spec, (long)n_frames, n_fft, hop, win, window, tables, (long)crop_start, (long)out_len, scale, out,
                     ts, (int)n_tiles, fr_cap, win_cap);
  return p2phd::check_launch("imdct4_fwd");
}
