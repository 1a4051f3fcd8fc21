// Spectrogram codec around the MDCT: Pix2PixHDModel.to_spectro / denormalize / to_audio
// (models/pix2pixHD_model.py:142-249): the explicit two-channel encoding of the published runs and the single-channel
// magnitude encoding, every mask_mode.
//
// encode: spec[B,F,M] (frames x bins, MDCT4 output) -> log_spectro[B,2,M,F] in [0,1], pha[B,1,M,F]
//   pass 1  transposes through a 32x32 LDS tile (reads coalesced along bins, writes coalesced along frames),
//           applies the neg/pos split (:150-151) and amplitude_to_DB(., 20, min_value, 1) (:154-155), and leaves
//           per-block (min, max, sum, sumsq) partials -- the global min/max of :167-168 without a host sync;
//   finalize reduces the partials to (min, max, mean, std) on the device;
//   pass 2  normalises in place (:193) and overwrites the top mask rows with min-max-scaled noise (:196-226).
// decode: log_spectro[B,2,M,F] -> spec[B,F,M] = (A0 - A1)/(2 alpha - 1), A = DB_to_amplitude(|x|(max-min)+min, 10, .5) - min_value
#include "common.h"

namespace {

__device__ __forceinline__ void block_reduce4(float mn, float mx, float s1, float s2, float* out4) {
  __shared__ float red[4][4];
  for (int o = 32; o > 0; o >>= 1) {
    mn = fminf(mn, __shfl_xor(mn, o)); mx = fmaxf(mx, __shfl_xor(mx, o));
    s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o);
  }
  const int tid = threadIdx.y * blockDim.x + threadIdx.x;
  const int w = tid >> 6;
  if ((tid & 63) == 0) { red[w][0] = mn; red[w][1] = mx; red[w][2] = s1; red[w][3] = s2; }
  __syncthreads();
  if (tid == 0) {
    const int nw = (blockDim.x * blockDim.y + 63) / 64;
    for (int i = 1; i < nw; ++i) { mn = fminf(mn, red[i][0]); mx = fmaxf(mx, red[i][1]); s1 += red[i][2]; s2 += red[i][3]; }
    out4[0] = mn; out4[1] = mx; out4[2] = s1; out4[3] = s2;
  }
}

__global__ __launch_bounds__(256) void encode_pass1_kernel(const float* __restrict__ spec, float* __restrict__ db,
                                                           float* __restrict__ pha, float* __restrict__ partials, int F,
                                                           int M, float alpha, float min_value, int channels) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z, m0 = blockIdx.x * 32, f0 = blockIdx.y * 32;
  const int tx = threadIdx.x, ty = threadIdx.y;
  for (int i = 0; i < 4; ++i) {
    const int f = f0 + ty + 8 * i, m = m0 + tx;
    tile[ty + 8 * i][tx] = (f < F && m < M) ? spec[((size_t)b * F + f) * M + m] : 0.f;
  }
  __syncthreads();
  float mn = INFINITY, mx = -INFINITY, s1 = 0.f, s2 = 0.f;
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + ty + 8 * i, f = f0 + tx;
    if (m < M && f < F) {
      const float s = tile[tx][ty + 8 * i];
      pha[((size_t)b * M + m) * F + f] = s > 0.f ? 1.f : (s < 0.f ? -1.f : 0.f);
      if (channels == 2) {                                       // explicit encoding (:149-158)
        const float neg = 0.5f * (fabsf(s) - s);
        const float pos = s + neg;
        const float d0 = 20.f * log10f(fmaxf(alpha * pos + (1.f - alpha) * neg, min_value)) - 20.f;
        const float d1 = 20.f * log10f(fmaxf((1.f - alpha) * pos + alpha * neg, min_value)) - 20.f;
        const size_t o = ((size_t)b * 2 * M + m) * F + f;
        db[o] = d0;
        db[o + (size_t)M * F] = d1;
        mn = fminf(mn, fminf(d0, d1)); mx = fmaxf(mx, fmaxf(d0, d1));
        s1 += d0 + d1; s2 += d0 * d0 + d1 * d1;
      } else {                                                   // magnitude only (:159-162); the sign travels in pha
        const float d0 = 20.f * log10f(fmaxf(fabsf(s) + min_value, min_value)) - 20.f;
        db[((size_t)b * M + m) * F + f] = d0;
        mn = fminf(mn, d0); mx = fmaxf(mx, d0);
        s1 += d0; s2 += d0 * d0;
      }
    }
  }
  const size_t blk = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
  block_reduce4(mn, mx, s1, s2, partials + 4 * blk);
}

__global__ __launch_bounds__(256) void stats4_kernel(const float* __restrict__ x, long n, float* __restrict__ partials) {
  float mn = INFINITY, mx = -INFINITY, s1 = 0.f, s2 = 0.f;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) {
    const float v = x[e];
    mn = fminf(mn, v); mx = fmaxf(mx, v); s1 += v; s2 += v * v;
  }
  block_reduce4(mn, mx, s1, s2, partials + 4 * blockIdx.x);
}

// out4 = (min, max, mean, unbiased std) over `count` values described by nblk partials
__global__ __launch_bounds__(256) void finalize4_kernel(const float* __restrict__ partials, int nblk, double count, float* __restrict__ out4) {
  __shared__ double rs[256], rq[256];
  __shared__ float rmn[256], rmx[256];
  float mn = INFINITY, mx = -INFINITY;
  double s1 = 0.0, s2 = 0.0;
  for (int i = threadIdx.x; i < nblk; i += 256) {
    mn = fminf(mn, partials[4 * i]); mx = fmaxf(mx, partials[4 * i + 1]);
    s1 += partials[4 * i + 2]; s2 += partials[4 * i + 3];
  }
  rmn[threadIdx.x] = mn; rmx[threadIdx.x] = mx; rs[threadIdx.x] = s1; rq[threadIdx.x] = s2;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int i = 1; i < 256; ++i) { mn = fminf(mn, rmn[i]); mx = fmaxf(mx, rmx[i]); s1 += rs[i]; s2 += rq[i]; }
    const double mean = s1 / count;
    const double var = count > 1 ? (s2 - count * mean * mean) / (count - 1) : 0.0;
    out4[0] = mn; out4[1] = mx; out4[2] = (float)mean; out4[3] = (float)sqrt(var > 0 ? var : 0.0);
  }
}

// mask_mode (:207-221): 0 = noise / (max - min) (single peak at 0), 1 = min-max scaled noise times a random sign,
// 2 = min-max scaled noise; no noise pointer = zeros (mask_mode None)
__global__ __launch_bounds__(256) void encode_pass2_kernel(float* __restrict__ db, const float* __restrict__ norm4,
                                                           const float* __restrict__ noise, const float* __restrict__ nnorm4,
                                                           int M, int F, int mask_rows, long total, int mask_mode,
                                                           const float* __restrict__ noise_sign) {
  const float mn = norm4[0], scale = 1.f / (norm4[1] - norm4[0]);
  float nmn = 0.f, nscale = 0.f;
  if (noise != nullptr) { nmn = mask_mode == 0 ? 0.f : nnorm4[0]; nscale = 1.f / (nnorm4[1] - nnorm4[0]); }
  const int keep = M - mask_rows;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int f = (int)(e % F);
    const long r = e / F;
    const int m = (int)(r % M);
    const long bc = r / M;                                     // b*2 + channel
    float v;
    if (m < keep) v = (db[e] - mn) * scale;
    else if (noise != nullptr) {
      const long q = (bc * mask_rows + (m - keep)) * F + f;
      v = (noise[q] - nmn) * nscale;
      if (mask_mode == 1) v *= noise_sign[q];
    } else v = 0.f;
    db[e] = v;
  }
}

__global__ __launch_bounds__(256) void decode_kernel(const float* __restrict__ x, const float* __restrict__ norm2,
                                                     float* __restrict__ spec, int F, int M, float alpha, float min_value) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z, m0 = blockIdx.x * 32, f0 = blockIdx.y * 32;
  const int tx = threadIdx.x, ty = threadIdx.y;
  const float mn = norm2[0], range = norm2[1] - norm2[0];
  const float inv = 1.f / (2.f * alpha - 1.f);
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + ty + 8 * i, f = f0 + tx;
    float v = 0.f;
    if (m < M && f < F) {
      const size_t o = ((size_t)b * 2 * M + m) * F + f;
      const float a0 = 10.f * exp10f((fabsf(x[o]) * range + mn) * 0.05f) - min_value;
      const float a1 = 10.f * exp10f((fabsf(x[o + (size_t)M * F]) * range + mn) * 0.05f) - min_value;
      v = (a0 - a1) * inv;
    }
    tile[ty + 8 * i][tx] = v;
  }
  __syncthreads();
  for (int i = 0; i < 4; ++i) {
    const int f = f0 + ty + 8 * i, m = m0 + tx;
    if (f < F && m < M) spec[((size_t)b * F + f) * M + m] = tile[tx][ty + 8 * i];
  }
}

// util.imdct's decode (util/util.py:104-126): amplitude = sum of the channels, sign = pha below `keep` rows and
// sign(ch0 - ch1) above (explicit encoding); single channel: sign = pha everywhere (the caller draws the random signs).
__global__ __launch_bounds__(256) void decode_signed_kernel(const float* __restrict__ x, const float* __restrict__ pha,
                                                            const float* __restrict__ norm2, float* __restrict__ spec, int F, int M,
                                                            int channels, int keep, float min_value, float scale) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z, m0 = blockIdx.x * 32, f0 = blockIdx.y * 32;
  const int tx = threadIdx.x, ty = threadIdx.y;
  const float mn = norm2[0], range = norm2[1] - norm2[0];
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + ty + 8 * i, f = f0 + tx;
    float v = 0.f;
    if (m < M && f < F) {
      const size_t o = ((size_t)b * channels * M + m) * F + f;
      const float a0 = 10.f * exp10f((fabsf(x[o]) * range + mn) * 0.05f) - min_value;
      float sgn = pha[((size_t)b * M + m) * F + f];
      float amp = a0;
      if (channels == 2) {
        const float a1 = 10.f * exp10f((fabsf(x[o + (size_t)M * F]) * range + mn) * 0.05f) - min_value;
        amp = a0 + a1;
        if (m >= keep) sgn = a0 > a1 ? 1.f : (a0 < a1 ? -1.f : 0.f);
      }
      v = amp * sgn * scale;
    }
    tile[ty + 8 * i][tx] = v;
  }
  __syncthreads();
  for (int i = 0; i < 4; ++i) {
    const int f = f0 + ty + 8 * i, m = m0 + tx;
    if (f < F && m < M) spec[((size_t)b * F + f) * M + m] = tile[tx][ty + 8 * i];
  }
}

}  // namespace

extern "C" int64_t p2phd_spectro_partials_floats(int64_t B, int64_t F, int64_t M) {
  return 4 * B * ((F + 31) / 32) * ((M + 31) / 32) + 4 * 1024;   // encode blocks + noise-statistics blocks
}

extern "C" int p2phd_spectro_encode_ex(const float* spec, int64_t B, int64_t F, int64_t M, int channels, float alpha,
                                       float min_value, int mask_rows, int mask_mode, const float* noise,
                                       const float* noise_sign, float* log_spectro, float* pha, float* norm4, float* partials,
                                       void* stream) {
  P2PHD_REQUIRE(B >= 0 && F >= 1 && M >= 1 && mask_rows >= 0 && mask_rows <= M, "spectro_encode: bad geometry");
  P2PHD_REQUIRE(channels == 1 || channels == 2, "spectro_encode: channels must be 2 (explicit encoding) or 1");
  P2PHD_REQUIRE(mask_mode >= 0 && mask_mode <= 2, "spectro_encode: mask_mode must be 0, 1 or 2");
  if (B == 0) return P2PHD_OK;
  P2PHD_REQUIRE(spec && log_spectro && pha && norm4 && partials, "spectro_encode: null pointer");
  P2PHD_REQUIRE(!(mask_mode == 1 && noise != nullptr && mask_rows > 0 && noise_sign == nullptr), "spectro_encode: mask mode1 needs the sign tensor");
  P2PHD_REQUIRE(B < 65536 && (F + 31) / 32 < 65536, "spectro_encode: grid too large");
  hipStream_t st = (hipStream_t)stream;
  dim3 grid((unsigned)((M + 31) / 32), (unsigned)((F + 31) / 32), (unsigned)B);
  const int nblk = (int)(grid.x * grid.y * grid.z);
  hipLaunchKernelGGL(encode_pass1_kernel, grid, dim3(32, 8), 0, st, spec, log_spectro, pha, partials, (int)F, (int)M, alpha, min_value, channels);
  hipLaunchKernelGGL(finalize4_kernel, dim3(1), dim3(256), 0, st, partials, nblk, (double)(channels * B * F * M), norm4);
  float* npart = partials + 4 * (size_t)nblk;
  float* nnorm = norm4 + 4;
  if (noise != nullptr && mask_rows > 0) {
    const long n = channels * B * mask_rows * F;
    const int nb = (int)std::min<long>((n + 255) / 256, 1024);
    hipLaunchKernelGGL(stats4_kernel, dim3(nb), dim3(256), 0, st, noise, n, npart);
    hipLaunchKernelGGL(finalize4_kernel, dim3(1), dim3(256), 0, st, npart, nb, (double)n, nnorm);
  }
  const long total = channels * B * M * F;
  const int blocks = (int)std::min<long>((total + 255) / 256, 8192);
  hipLaunchKernelGGL(encode_pass2_kernel, dim3(blocks), dim3(256), 0, st, log_spectro, norm4, mask_rows > 0 ? noise : nullptr,
                     nnorm, (int)M, (int)F, mask_rows, total, mask_mode, noise_sign);
  return p2phd::check_launch("spectro_encode");
}

extern "C" int p2phd_spectro_encode(const float* spec, int64_t B, int64_t F, int64_t M, float alpha, float min_value,
                                    int mask_rows, const float* noise, float* log_spectro, float* pha, float* norm4,
                                    float* partials, void* stream) {
  return p2phd_spectro_encode_ex(spec, B, F, M, 2, alpha, min_value, mask_rows, 2, noise, nullptr, log_spectro, pha, norm4,
                                 partials, stream);
}

extern "C" int p2phd_spectro_decode(const float* log_spectro, const float* norm_min_max, int64_t B, int64_t F, int64_t M,
                                    float alpha, float min_value, float* spec, void* stream) {
  P2PHD_REQUIRE(B >= 0 && F >= 1 && M >= 1, "spectro_decode: bad geometry");
  P2PHD_REQUIRE(alpha != 0.5f, "spectro_decode: alpha = 0.5 makes the explicit encoding singular");
  if (B == 0) return P2PHD_OK;
  P2PHD_REQUIRE(log_spectro && norm_min_max && spec, "spectro_decode: null pointer");
  dim3 grid((unsigned)((M + 31) / 32), (unsigned)((F + 31) / 32), (unsigned)B);
  hipLaunchKernelGGL(decode_kernel, grid, dim3(32, 8), 0, (hipStream_t)stream, log_spectro, norm_min_max, spec, (int)F, (int)M, alpha, min_value);
  return p2phd::check_launch("spectro_decode");
}

extern "C" int p2phd_spectro_decode_signed(const float* log_spectro, const float* pha, const float* norm_min_max, int64_t B,
                                           int64_t F, int64_t M, int channels, int keep_rows, float min_value, float scale,
                                           float* spec, void* stream) {
  P2PHD_REQUIRE(B >= 0 && F >= 1 && M >= 1, "spectro_decode_signed: bad geometry");
  P2PHD_REQUIRE(channels == 1 || channels == 2, "spectro_decode_signed: channels must be 1 or 2, got %d", channels);
  P2PHD_REQUIRE(keep_rows >= 0 && keep_rows <= M, "spectro_decode_signed: keep_rows %d outside [0, %lld]", keep_rows, (long long)M);
  if (B == 0) return P2PHD_OK;
  P2PHD_REQUIRE(log_spectro && pha && norm_min_max && spec, "spectro_decode_signed: null pointer");
  P2PHD_REQUIRE(B < 65536 && (F + 31) / 32 < 65536, "spectro_decode_signed: grid too large");
  dim3 grid((unsigned)((M + 31) / 32), (unsigned)((F + 31) / 32), (unsigned)B);
  hipLaunchKernelGGL(decode_signed_kernel, grid, dim3(32, 8), 0, (hipStream_t)stream, log_spectro, pha, norm_min_max, spec, (int)F,
                     (int)M, channels, keep_rows, min_value, scale);
  return p2phd::check_launch("spectro_decode_signed");
}
