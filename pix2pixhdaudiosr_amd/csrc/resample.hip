// Band-limited sample-rate conversion for the input feeder (data/audio_dataset.py:55-57,109-113: HR -> LR -> HR through
// torchaudio.functional.resample, whose source is not part of the reference; the definition used here -- Hann-windowed
// sinc, lowpass_filter_width 6, rolloff 0.99 -- is stated in oracle/feeder.py and DESIGN.md, parity unpinned).
//
// Polyphase gather: with o = orig/gcd, n = new/gcd, y[j*n + p] = sum_k h[p][k] * x[j*o + k - width].  One workgroup stages
// the input span of a run of j-blocks in LDS once (each input sample is read from HBM once per workgroup, zero outside
// the row) and every thread produces outputs from it; the phase table h [n][klen] stays L1/L2 resident.  HBM-bound:
// 4 B read + 4*n/o B written per input sample.
#include "common.h"
#include <cmath>
#include <numeric>

namespace {

constexpr int kThreads = 256;

__global__ __launch_bounds__(kThreads) void resample_kernel(const float* __restrict__ x, long T, int o, int n, int klen, int width,
                                                            const float* __restrict__ h, float* __restrict__ out, long T_out,
                                                            int JT, long J_total) {
  extern __shared__ float s_x[];
  const long b = blockIdx.y;
  const long j0 = (long)blockIdx.x * JT;
  const int nj = (int)min((long)JT, J_total - j0);
  const int span = (nj - 1) * o + klen;
  const long xi0 = j0 * o - width;
  const float* xb = x + b * T;
  for (int i = threadIdx.x; i < span; i += kThreads) {
    const long idx = xi0 + i;
    s_x[i] = (idx >= 0 && idx < T) ? xb[idx] : 0.f;
  }
  __syncthreads();
  float* ob = out + b * T_out;
  const int nout = nj * n;
  for (int e = threadIdx.x; e < nout; e += kThreads) {
    const int j = e / n, p = e - j * n;
    const long m = (j0 + j) * n + p;
    if (m >= T_out) continue;
    const float* hp = h + (size_t)p * klen;
    const float* xs = s_x + j * o;
    float acc = 0.f;
    for (int k = 0; k < klen; ++k) acc = fmaf(hp[k], xs[k], acc);
    ob[m] = acc;
  }
}

struct Geo { int o, n, width, klen; double base; };

int geometry(int orig, int nw, int lpw, double rolloff, Geo* g) {
  P2PHD_REQUIRE(orig >= 1 && nw >= 1, "resample: sample rates must be positive (got %d -> %d)", orig, nw);
  P2PHD_REQUIRE(lpw >= 1 && rolloff > 0.0 && rolloff <= 1.0, "resample: bad lowpass_filter_width %d / rolloff %g", lpw, rolloff);
  const int gd = std::gcd(orig, nw);
  g->o = orig / gd;
  g->n = nw / gd;
  g->base = std::min(g->o, g->n) * rolloff;
  g->width = (int)std::ceil(lpw * g->o / g->base);
  g->klen = 2 * g->width + g->o;
  P2PHD_REQUIRE((int64_t)g->n * g->klen <= (1 << 24), "resample: %d -> %d needs a %lld-entry phase table; reduce the rates",
                orig, nw, (long long)g->n * g->klen);
  return P2PHD_OK;
}

}  // namespace

extern "C" int p2phd_resample_geometry(int orig_freq, int new_freq, int lowpass_filter_width, double rolloff, int* o, int* n,
                                       int* width, int* klen) {
  Geo g;
  if (int rc = geometry(orig_freq, new_freq, lowpass_filter_width, rolloff, &g)) return rc;
  if (o) *o = g.o;
  if (n) *n = g.n;
  if (width) *width = g.width;
  if (klen) *klen = g.klen;
  return P2PHD_OK;
}

extern "C" size_t p2phd_resample_kernel_floats(int orig_freq, int new_freq, int lowpass_filter_width, double rolloff) {
  Geo g;
  if (geometry(orig_freq, new_freq, lowpass_filter_width, rolloff, &g) != P2PHD_OK) return 0;
  return (size_t)g.n * g.klen;
}

extern "C" int p2phd_resample_kernel_fill(int orig_freq, int new_freq, int lowpass_filter_width, double rolloff, float* host_out) {
  Geo g;
  if (int rc = geometry(orig_freq, new_freq, lowpass_filter_width, rolloff, &g)) return rc;
  P2PHD_REQUIRE(host_out != nullptr, "resample_kernel_fill: null output");
  const double pi = 3.14159265358979323846264338327950288;
  const double lpw = lowpass_filter_width;
  for (int p = 0; p < g.n; ++p)
    for (int k = 0; k < g.klen; ++k) {
      double t = (-(double)p / g.n + (double)(k - g.width) / g.o) * g.base;
      t = std::min(std::max(t, -lpw), lpw);
      const double w = std::cos(t * pi / lpw / 2);
      const double tp = t * pi;
      const double s = tp == 0.0 ? 1.0 : std::sin(tp) / tp;
      host_out[(size_t)p * g.klen + k] = (float)(s * w * w * (g.base / g.o));
    }
  return P2PHD_OK;
}

extern "C" int64_t p2phd_resample_out_len(int64_t T, int orig_freq, int new_freq) {
  if (T < 0 || orig_freq < 1 || new_freq < 1) return -1;
  const int gd = std::gcd(orig_freq, new_freq);
  const int64_t o = orig_freq / gd, n = new_freq / gd;
  return (n * T + o - 1) / o;
}

extern "C" int p2phd_resample_fwd(const float* x, int64_t B, int64_t T, int orig_freq, int new_freq, int lowpass_filter_width,
                                  double rolloff, const float* kernel, float* out, int64_t T_out, void* stream) {
  Geo g;
  if (int rc = geometry(orig_freq, new_freq, lowpass_filter_width, rolloff, &g)) return rc;
  P2PHD_REQUIRE(B >= 0 && T >= 0, "resample_fwd: negative size");
  P2PHD_REQUIRE(T_out == p2phd_resample_out_len(T, orig_freq, new_freq), "resample_fwd: output length %lld, expected %lld",
                (long long)T_out, (long long)p2phd_resample_out_len(T, orig_freq, new_freq));
  if (B == 0 || T_out == 0) return P2PHD_OK;
  P2PHD_REQUIRE(x && kernel && out, "resample_fwd: null pointer");
  P2PHD_REQUIRE(B < 65536, "resample_fwd: too many rows");
  const int64_t J_total = p2phd::cdiv(T_out, g.n);
  int JT = std::max(1, std::min(4096 / g.n, 8192 / g.o));
  const int64_t tiles = p2phd::cdiv(J_total, JT);
  P2PHD_REQUIRE(tiles < (1ll << 31), "resample_fwd: grid too large");
  const size_t lds = sizeof(float) * ((size_t)(JT - 1) * g.o + g.klen);
  P2PHD_REQUIRE(lds <= 160 * 1024, "resample_fwd: %d -> %d needs %zu B of LDS", orig_freq, new_freq, lds);
  if (lds > 48 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(resample_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(resample_kernel, dim3((unsigned)tiles, (unsigned)B), dim3(kThreads), lds, (hipStream_t)stream, x, (long)T, g.o,
                     g.n, g.klen, g.width, kernel, out, (long)T_out, JT, (long)J_total);
  return p2phd::check_launch("resample_fwd");
}
