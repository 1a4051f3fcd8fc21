// MDCT4 / IMDCT4 fast path for the 50 %-overlap geometry every caller of the reference uses (hop = n_fft / 2,
// win = n_fft; models/mdct.py:461-566, pix2pixHD_model.py:37-40): n_fft = 1024 or 2048.
//
// One WAVEFRONT owns a frame: the N/4-point complex FFT of the DCT-IV lives in its registers (R = N/256 points per
// lane, point index i = lane + 64 r) and changes hands between radix-4 passes through a wave-private LDS buffer.
// Nothing in the transform needs a workgroup barrier: LDS operations of one wave execute in issue order, so a
// compiler-level wave fence between the writes and the reads of an exchange is all the synchronisation there is
// (the generic kernel of mdct.hip puts a __syncthreads() behind every radix pass, twice per frame round).
//   forward : the wave stages the 3 hops its two consecutive frames span (coalesced float4, zero padding by
//             bounds), folds + windows + pre-rotates them into registers, runs the FFT, post-rotates and writes
//             whole bin rows (float4 per lane).  HBM: read hop + write N/2 floats per frame.
//   inverse : a workgroup computes 8 frames (two per wave) into a shared LDS ring and -- after the ONE barrier of
//             the kernel -- overlap-adds them into 7 hops of output by gather (fixed summation order, no `fold`
//             buffer, no atomics); the frame that two neighbouring workgroups both need is recomputed (1/7 extra
//             FFT work, its row comes from L2).
// LDS images are bank-conflict free by construction: float buffers swap the two elements of a pair when bit 5 of the
// index is set (stride-2 accesses of the fold / unfold then hit 32 distinct banks), the complex exchange buffer XORs
// index bits [5:4] into bits [3:2] and [1:0] (every radix pass writes 16 distinct 8-byte slots per 16-lane group).
#include "common.h"
#include "fft_wave.h"

namespace {

using p2phd_fft::cmul;
using p2phd_fft::cadd;
using p2phd_fft::csub;

constexpr int kWaves = 4;
constexpr int kFramesPerWave = 2;
constexpr int kFramesPerWG = kWaves * kFramesPerWave;

// LDS executes one wave's operations in order; this keeps the compiler from moving them across each other and waits for
// the wave's earlier LDS operations only (a wavefront-scope fence would also wait for every global load / store in flight)
__device__ __forceinline__ void wave_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

__device__ __forceinline__ int sf(int n) { return n ^ ((n >> 5) & 1); }                 // float image
__device__ __forceinline__ int sc(int i) { return i ^ (((i >> 4) & 3) * 5); }           // complex exchange image

// float4 at an aligned index n of a swizzled float image (the four elements share bit 5)
__device__ __forceinline__ void store4(float* img, int n, float4 v) {
  if ((n >> 5) & 1) v = make_float4(v.y, v.x, v.w, v.z);
  *reinterpret_cast<float4*>(img + n) = v;
}
__device__ __forceinline__ float4 load4(const float* img, int n) {
  float4 v = *reinterpret_cast<const float4*>(img + n);
  if ((n >> 5) & 1) v = make_float4(v.y, v.x, v.w, v.z);
  return v;
}

// Per-lane constants of the transform: FFT twiddles of the passes with stride 4, 16, 64 (k = lane & (ns - 1): the same
// for every butterfly of the lane), of the final radix-2 pass (n_fft 2048) and the DCT-IV rotation of the lane's points.
template <int R>
struct Consts {
  float2 t4[3], t16[3], t64[3];
  float2 t2[R == 8 ? 4 : 1];
  float2 rot[R];
};

template <int R>
__device__ __forceinline__ void load_consts(Consts<R>& c, const float* __restrict__ tables, int lane) {
  constexpr int Q = 64 * R;
  const float2* tw = reinterpret_cast<const float2*>(tables);
  const float2* rt = tw + Q;
#pragma unroll
  for (int m = 1; m <= 3; ++m) {
    c.t4[m - 1] = tw[m * (lane & 3) * (Q / 16)];
    c.t16[m - 1] = tw[m * (lane & 15) * (Q / 64)];
    c.t64[m - 1] = tw[m * (lane & 63) * (Q / 256)];
  }
  if constexpr (R == 8) {
#pragma unroll
    for (int q = 0; q < 4; ++q) c.t2[q] = tw[lane + 64 * q];
  }
#pragma unroll
  for (int r = 0; r < R; ++r) c.rot[r] = rt[lane + 64 * r];
}

// One radix-4 Stockham pass with stride NS on the register-resident points (v[r] = point lane + 64 r); the result is
// redistributed to the same ownership through `xb` unless the pass already leaves it there (NS = 64).
template <int R, int NS>
__device__ __forceinline__ void pass4(float2 (&v)[R], const float2 (&tw)[3], float2* xb, int lane) {
  constexpr int NB = R / 4;
  float2 o[R];
#pragma unroll
  for (int q = 0; q < NB; ++q) {
    float2 v0 = v[q], v1 = v[q + NB], v2 = v[q + 2 * NB], v3 = v[q + 3 * NB];
    if constexpr (NS > 1) {
      v1 = cmul(v1, tw[0]);
      v2 = cmul(v2, tw[1]);
      v3 = cmul(v3, tw[2]);
    }
    const float2 A = cadd(v0, v2), B = csub(v0, v2), C = cadd(v1, v3);
    const float2 d = csub(v1, v3);
    const float2 D = make_float2(d.y, -d.x);                 // -i (v1 - v3)
    o[4 * q + 0] = cadd(A, C);
    o[4 * q + 1] = cadd(B, D);
    o[4 * q + 2] = csub(A, C);
    o[4 * q + 3] = csub(B, D);
  }
  if constexpr (NS == 64) {
    // j0 = 256 q + lane, outputs j0 + 64 m: already point lane + 64 (4 q + m)
#pragma unroll
    for (int r = 0; r < R; ++r) v[r] = o[r];
  } else {
#pragma unroll
    for (int q = 0; q < NB; ++q) {
      const int j = lane + 64 * q;
      const int k = j & (NS - 1);
      const int j0 = ((j - k) << 2) + k;
#pragma unroll
      for (int m = 0; m < 4; ++m) xb[sc(j0 + m * NS)] = o[4 * q + m];
    }
    wave_sync();
#pragma unroll
    for (int r = 0; r < R; ++r) v[r] = xb[sc(lane + 64 * r)];
    wave_sync();
  }
}

template <int R>
__device__ __forceinline__ void fft_regs(float2 (&v)[R], const Consts<R>& c, float2* xb, int lane) {
  const float2 none[3] = {};
  pass4<R, 1>(v, none, xb, lane);
  pass4<R, 4>(v, c.t4, xb, lane);
  pass4<R, 16>(v, c.t16, xb, lane);
  pass4<R, 64>(v, c.t64, xb, lane);
  if constexpr (R == 8) {
    // radix-2, stride 256: butterflies j = lane + 64 q use points j and j + 256 and leave them in place
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float2 a = v[q], b = cmul(v[q + 4], c.t2[q]);
      v[q] = cadd(a, b);
      v[q + 4] = csub(a, b);
    }
  }
}

// ------------------------------------------------------------------------------------------
// forward: x[B,T] -> out[B,F,M], hop = M = N/2, win = N
// ------------------------------------------------------------------------------------------
template <int R>
__global__ __launch_bounds__(64 * kWaves) void mdct4_fast_fwd_kernel(const float* __restrict__ x, long T,
                                                                      const float* __restrict__ window,
                                                                      const float* __restrict__ tables, long start_pad, long F,
                                                                      float scale, float* __restrict__ out, int n_tiles,
                                                                      int wg_tiles, int iters) {
  constexpr int N = 256 * R, M = N / 2;
  constexpr int SEG = (kFramesPerWave + 1) * M;                // floats staged per wave
  extern __shared__ float4 smem_raw[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float* seg = reinterpret_cast<float*>(smem_raw) + (size_t)wave * SEG;

  // A workgroup walks `iters` consecutive 8-frame tiles of one row (round 3): the per-lane constants below (twiddles,
  // rotations, window values: ~30 dependent loads) are fetched once per wave instead of once per tile, and a grid of more
  // than one resident round (4 workgroups per CU: 1024) collapses into one.
  const long b = blockIdx.x / wg_tiles;
  const long tile0 = (long)(blockIdx.x % wg_tiles) * iters;
  const float* xb = x + b * T;
  Consts<R> cs;
  load_consts<R>(cs, tables, lane);
  // window values of the lane's fold: points r < R/2 use positions (3M/2-1-2i, 3M/2+2i, M/2-1-2i, M/2+2i),
  // points r >= R/2 use (2i-M/2, 3M/2-1-2i, M/2+2i, 5M/2-1-2i).  Positions are one add from the lane index and are
  // recomputed where they are used (kept in registers they cost a wave per SIMD); the window values stay resident.
  auto fold_pos = [&](int r, int e) -> int {
    const int i2 = 2 * (lane + 64 * r);
    if (r < R / 2) return e == 0 ? 3 * (M / 2) - 1 - i2 : (e == 1 ? 3 * (M / 2) + i2 : (e == 2 ? M / 2 - 1 - i2 : M / 2 + i2));
    return e == 0 ? i2 - M / 2 : (e == 1 ? 3 * (M / 2) - 1 - i2 : (e == 2 ? M / 2 + i2 : 5 * (M / 2) - 1 - i2));
  };
  float wv[R][4];
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int e = 0; e < 4; ++e) wv[r][e] = window[fold_pos(r, e)];
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
  const long tile = tile0 + it;
  const long t0 = tile * kFramesPerWG + wave * kFramesPerWave;
  if (tile >= n_tiles || t0 >= F) break;                       // no workgroup barrier anywhere: a wave may leave
  const int nf = (int)min((long)kFramesPerWave, F - t0);

  // stage the wave's signal span: coalesced float4, zeros outside [0, T) (T, start_pad, hop are multiples of 4)
  const long p0 = t0 * M - start_pad;
  float4 ld[SEG / 256];
#pragma unroll
  for (int c = 0; c < SEG / 256; ++c) {
    const long idx = p0 + 4 * (lane + 64 * c);
    const bool ok = idx >= 0 && idx + 3 < T;
    ld[c] = *reinterpret_cast<const float4*>(xb + (ok ? idx : 0));
    if (!ok) ld[c] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
#pragma unroll
  for (int c = 0; c < SEG / 256; ++c) store4(seg, 4 * (lane + 64 * c), ld[c]);
  wave_sync();

  // fold + window + pre-rotation of both frames, before the staging area is reused
  float2 v[kFramesPerWave][R];
#pragma unroll
  for (int f = 0; f < kFramesPerWave; ++f) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      float u[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) u[e] = seg[sf(f * M + fold_pos(r, e))] * wv[r][e];
      const float2 z = r < R / 2 ? make_float2(-u[0] - u[1], u[2] - u[3]) : make_float2(u[0] - u[1], -u[2] - u[3]);
      v[f][r] = cmul(z, cs.rot[r]);
    }
  }
  wave_sync();

  float2* xbuf = reinterpret_cast<float2*>(seg);               // Q complex = M floats
  float* stage = seg + M;                                      // M floats (SEG = 3 M)
#pragma unroll
  for (int f = 0; f < kFramesPerWave; ++f) {
    if (f < nf) {
      fft_regs<R>(v[f], cs, xbuf, lane);
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int i2 = 2 * (lane + 64 * r);
        const float2 c = cmul(v[f][r], cs.rot[r]);
        stage[sf(i2)] = c.x * scale;
        stage[sf(M - 1 - i2)] = -c.y * scale;
      }
      wave_sync();
      float4* o4 = reinterpret_cast<float4*>(out + ((b * F + t0 + f) * (long)M));
#pragma unroll
      for (int c = 0; c < M / 256; ++c) o4[lane + 64 * c] = load4(stage, 4 * (lane + 64 * c));
      wave_sync();
    }
  }
  }                                                            // tiles of this workgroup
}

// ------------------------------------------------------------------------------------------
// inverse with gather overlap-add: spec[B,F,M] -> out[B,out_len]; out[m] = scale * sum_t w[q] y_t(q), q = m + crop - t M
// ------------------------------------------------------------------------------------------
template <int R>
__global__ __launch_bounds__(64 * kWaves) void imdct4_fast_kernel(const float* __restrict__ spec, long F,
                                                                   const float* __restrict__ window,
                                                                   const float* __restrict__ tables, long crop, long out_len,
                                                                   float scale, float* __restrict__ out, int n_tiles) {
  constexpr int N = 256 * R, M = N / 2, H = M / 2;
  constexpr int BLK = kFramesPerWG - 1;                        // hops of output per workgroup
  extern __shared__ float4 smem_raw[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float* ring = reinterpret_cast<float*>(smem_raw);            // [kFramesPerWG][M] DCT-IV outputs d_t (swizzled)
  float* priv = ring + kFramesPerWG * M + (size_t)wave * M;    // wave-private: input row / FFT exchange
  float2* xbuf = reinterpret_cast<float2*>(priv);

  const long b = blockIdx.x / n_tiles;
  // padded-domain hop s (samples [s M, (s+1) M)) = second half of frame s-1 + first half of frame s;
  // this workgroup produces hops s0 .. s0 + BLK - 1 from frames s0 - 1 .. s0 + BLK - 1 (ring slot = t - (s0 - 1))
  const long s_first = crop / M;                               // first hop that holds an output sample
  const long s0 = s_first + (long)(blockIdx.x % n_tiles) * BLK;
  const long tf0 = s0 - 1;

  Consts<R> cs;
  load_consts<R>(cs, tables, lane);
  float4 ld[kFramesPerWave][M / 256];
  bool live[kFramesPerWave];
#pragma unroll
  for (int f = 0; f < kFramesPerWave; ++f) {
    const long t = tf0 + wave * kFramesPerWave + f;
    live[f] = t >= 0 && t < F;
    const float4* row = reinterpret_cast<const float4*>(spec + ((b * F + (live[f] ? t : 0)) * (long)M));
#pragma unroll
    for (int c = 0; c < M / 256; ++c) ld[f][c] = row[lane + 64 * c];
  }
#pragma unroll
  for (int f = 0; f < kFramesPerWave; ++f) {
    float* d = ring + (size_t)(wave * kFramesPerWave + f) * M;
    if (live[f]) {
#pragma unroll
      for (int c = 0; c < M / 256; ++c) store4(priv, 4 * (lane + 64 * c), ld[f][c]);
      wave_sync();
      float2 v[R];
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int i2 = 2 * (lane + 64 * r);
        v[r] = cmul(make_float2(priv[sf(i2)], priv[sf(M - 1 - i2)]), cs.rot[r]);
      }
      wave_sync();
      fft_regs<R>(v, cs, xbuf, lane);
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int i2 = 2 * (lane + 64 * r);
        const float2 c = cmul(v[r], cs.rot[r]);
        d[sf(i2)] = c.x;
        d[sf(M - 1 - i2)] = -c.y;
      }
    }
  }
  __syncthreads();

  // overlap-add: 4 consecutive samples per thread and pass; hop s, offset q (multiple of 4) in [0, M):
  //   frame s   (first half,  window w[q]):      q <  H:  d_s[q + H]            q >= H: -d_s[3H - 1 - q]
  //   frame s-1 (second half, window w[q + M]):  q <  H: -d_{s-1}[H - 1 - q]    q >= H: -d_{s-1}[q - H]
  for (int e = tid; e < BLK * (M / 4); e += 64 * kWaves) {
    const int blk = e / (M / 4);
    const int q = 4 * (e - blk * (M / 4));
    const long s = s0 + blk;
    const long m = s * M + q - crop;                           // first of the four output samples
    if (m + 3 < 0 || m >= out_len) continue;
    const float* dprev = ring + (size_t)blk * M;               // frame s - 1
    const float* dcur = dprev + M;                             // frame s
    const bool has_prev = s - 1 >= 0 && s - 1 < F, has_cur = s < F;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (has_cur) {
      const float4 w = *reinterpret_cast<const float4*>(window + q);
      if (q < H) {
        const float4 y = load4(dcur, q + H);
        acc = make_float4(w.x * y.x, w.y * y.y, w.z * y.z, w.w * y.w);
      } else {
        const float4 y = load4(dcur, 3 * H - 4 - q);           // elements 3H-4-q .. 3H-1-q, used in reverse
        acc = make_float4(-w.x * y.w, -w.y * y.z, -w.z * y.y, -w.w * y.x);
      }
    }
    if (has_prev) {
      const float4 w = *reinterpret_cast<const float4*>(window + M + q);
      if (q < H) {
        const float4 y = load4(dprev, H - 4 - q);              // elements H-4-q .. H-1-q, reversed
        acc.x -= w.x * y.w; acc.y -= w.y * y.z; acc.z -= w.z * y.y; acc.w -= w.w * y.x;
      } else {
        const float4 y = load4(dprev, q - H);
        acc.x -= w.x * y.x; acc.y -= w.y * y.y; acc.z -= w.z * y.z; acc.w -= w.w * y.w;
      }
    }
    acc = make_float4(acc.x * scale, acc.y * scale, acc.z * scale, acc.w * scale);
    float* o = out + b * out_len + m;
    if (m >= 0 && m + 3 < out_len) {
      *reinterpret_cast<float4*>(o) = acc;
    } else {
      const float a4[4] = {acc.x, acc.y, acc.z, acc.w};
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (m + k >= 0 && m + k < out_len) o[k] = a4[k];
    }
  }
}

template <int R>
int launch_fwd(const float* x, int64_t B, int64_t T, const float* window, const float* tables, int64_t start_pad,
               int64_t n_frames, float scale, float* out, hipStream_t st) {
  constexpr int M = 128 * R;
  const int64_t n_tiles = p2phd::cdiv(n_frames, kFramesPerWG);
  P2PHD_REQUIRE(B * n_tiles < (1ll << 31), "mdct4_fwd: grid too large");
  const size_t lds = sizeof(float) * (size_t)kWaves * (kFramesPerWave + 1) * M;
  auto kern = mdct4_fast_fwd_kernel<R>;
  if (lds > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  // tiles per workgroup: as many as keep the grid within one resident round (4 workgroups per CU by registers), at most 4
  // (resident workgroups: 3 per CU for n_fft 1024 -- 148 VGPRs --, 2 for n_fft 2048; p2phd_set_option("mdct_iters", n) forces n)
  const int64_t resident = (R == 4 ? 3 : 2) * 256;
  int iters = 1;
  while (iters < 4 && B * p2phd::cdiv(n_tiles, iters) > resident) ++iters;
  if (p2phd::g_opt_mdct_iters > 0) iters = p2phd::g_opt_mdct_iters;
  const int64_t wg_tiles = p2phd::cdiv(n_tiles, iters);
  hipLaunchKernelGGL(kern, dim3((unsigned)(B * wg_tiles)), dim3(64 * kWaves), lds, st, x, (long)T, window, tables, (long)start_pad,
                     (long)n_frames, scale, out, (int)n_tiles, (int)wg_tiles, iters);
  return p2phd::check_launch("mdct4_fwd(fast)");
}

template <int R>
int launch_inv(const float* spec, int64_t B, int64_t n_frames, const float* window, const float* tables, int64_t crop,
               int64_t out_len, float scale, float* out, hipStream_t st) {
  constexpr int M = 128 * R;
  const int64_t s_first = crop / M, s_last = (out_len - 1 + crop) / M;
  const int64_t n_tiles = p2phd::cdiv(s_last - s_first + 1, kFramesPerWG - 1);
  P2PHD_REQUIRE(B * n_tiles < (1ll << 31), "imdct4_fwd: grid too large");
  const size_t lds = sizeof(float) * (size_t)(kFramesPerWG + kWaves) * M;
  auto kern = imdct4_fast_kernel<R>;
  if (lds > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(kern, dim3((unsigned)(B * n_tiles)), dim3(64 * kWaves), lds, st, spec, (long)n_frames, window, tables, (long)crop,
                     (long)out_len, scale, out, (int)n_tiles);
  return p2phd::check_launch("imdct4_fwd(fast)");
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

namespace p2phd {

// The 50 %-overlap geometry with 16-byte aligned rows takes the register-resident kernels; everything else (other
// hops / windows / sizes, odd T) the generic LDS kernels of mdct.hip.  Same results either way (tests run both).
bool mdct4_fast_ok(int n_fft, int hop, int win, int64_t row_len, int64_t start_pad, const void* a, const void* b) {
  if (g_opt_mdct_generic) return false;
  return (n_fft == 1024 || n_fft == 2048) && hop * 2 == n_fft && win == n_fft && row_len % 4 == 0 && start_pad % 4 == 0 &&
         aligned16(a) && aligned16(b);
}

int mdct4_fast_fwd(const float* x, int64_t B, int64_t T, int n_fft, const float* window, const float* tables,
                   int64_t start_pad, int64_t n_frames, float scale, float* out, hipStream_t st) {
  if (n_fft == 1024) return launch_fwd<4>(x, B, T, window, tables, start_pad, n_frames, scale, out, st);
  return launch_fwd<8>(x, B, T, window, tables, start_pad, n_frames, scale, out, st);
}

int imdct4_fast(const float* spec, int64_t B, int64_t n_frames, int n_fft, const float* window, const float* tables,
                int64_t crop, int64_t out_len, float scale, float* out, hipStream_t st) {
  if (n_fft == 1024) return launch_inv<4>(spec, B, n_frames, window, tables, crop, out_len, scale, out, st);
  return launch_inv<8>(spec, B, n_frames, window, tables, crop, out_len, scale, out, st);
}

}  // namespace p2phd
