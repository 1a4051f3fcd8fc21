// The discriminator's LAST layer: Conv2d(8 ndf = 512, 1, kernel 4, stride 1, padding 2), no normalisation, no activation
// (models/networks.py:361-363, once per scale and per discriminator pass), forward and input gradient, 16-bit storage.
//
// 1.2 GFLOP on a 146 MB tensor (2B = 64 samples at configs[1]): HBM-bound, 30 us at 5 TB/s.  As W-folded gather-GEMMs the
// forward re-gathered the input once per kernel ROW through 128 x 32 tiles (92 us) and the input gradient ran a K = 128 GEMM on
// 128 x 128 tiles that are all prologue and epilogue (139 us; profiles/r04_step_kernel_trace_summary.csv).  Both are ONE pass
// over the wide tensor here, with the 16 taps as an MFMA dimension:
//   forward     P[pixel][tap] = sum_c x[pixel][c] w[c][tap]      M = 16 input pixels of a wave, N = 16 taps, K = C (v_mfma 16x16x32);
//               x is read once, straight into A fragments; the weights live in registers as B fragments.  A second, tiny launch
//               adds the 16 shifted partials of every output pixel: y[ho][wo] = b + sum_{th,tw} P[ho + th - 2][wo + tw - 2][th, tw].
//   input grad  dx[pixel][c] = sum_tap dyG[pixel][tap] w[tap][c]  M = 16 input pixels, K = 16 taps (+ 16 zeros), N = 128 channels
//               per wave (a wave owns a QUARTER of the channels of its pixels: 32 B-fragment registers, no LDS for weights);
//               dyG[pixel][tap] = dy[hi - th + 2][wi - tw + 2] gathered from the one-channel gradient (2.3 MB: cache hits).
//               The 16 x 128 block is transposed through a 4 KB LDS patch of the wave's own into 16-byte pieces of NHWC rows,
//               with the skip / parked gradient added (`addend`) and, optionally, the first pass of the PRODUCER's
//               InstanceNorm backward riding on the stores (conv.hip's fused store loop: p2phd_conv_dgrad_bsum) -- a lane keeps
//               one 8-channel column for all its rows, so the sums stay in registers for the wave's whole run of pixels.
#include "convplan.h"

namespace {

using namespace p2phd;

typedef p2phd_h16 bf16_t;
typedef __attribute__((ext_vector_type(8))) bf16_t bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

constexpr unsigned kOOB = 0xFFFFFFF0u;
constexpr int kMaxKS = 16;                                    // forward: C / 32 k-steps, at most 512 channels (64 weight registers)
constexpr int kQCH = 128;                                     // input gradient: channels per wave
constexpr int kStagePitchG = kQCH * 2 + 16;                   // staged pixel row of the input gradient (bytes)

// ---- weights --------------------------------------------------------------------------------------------------------------
// forward:  wf[(ks * 64 + lane) * 8 + e]  = w[0][c = 32 ks + 8 (lane >> 4) + e][th][tw],  tap = 4 th + tw = lane & 15
// dgrad:    wg[(nb * 64 + lane) * 8 + e]  = w[0][c = 16 nb + (lane & 15)][th][tw],         tap = 8 (lane >> 4) + e  (zero for taps >= 16)
__global__ void dlast_pack_kernel(const float* __restrict__ w, bf16_t* __restrict__ wf, bf16_t* __restrict__ wg, int C) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = idx & 63, blk = idx >> 6;
  if (wf != nullptr && blk < C / 32) {
    bf16_t v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (bf16_t)w[(size_t)(32 * blk + 8 * (lane >> 4) + e) * 16 + (lane & 15)];
    *reinterpret_cast<uint4*>(wf + (size_t)idx * 8) = *reinterpret_cast<const uint4*>(v);
  }
  if (wg != nullptr && blk < C / 16) {
    bf16_t v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int tap = 8 * (lane >> 4) + e;
      v[e] = (bf16_t)(tap < 16 ? w[(size_t)(16 * blk + (lane & 15)) * 16 + tap] : 0.f);
    }
    *reinterpret_cast<uint4*>(wg + (size_t)idx * 8) = *reinterpret_cast<const uint4*>(v);
  }
}

// ---- forward, pass 1: per-pixel tap partials ------------------------------------------------------------------------------
struct DLastFwdArgs {
  const bf16_t* x;        // [npix_in][C]
  const bf16_t* wf;
  float* part;            // [npix_in][16]
  long npix_in;
  int C;
  unsigned x_bytes;
};

template <int KS>
__global__ __launch_bounds__(256) void dlast_fwd_partial_kernel(const DLastFwdArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i16 = lane & 15, kq = lane >> 4;
  const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, (int)a.x_bytes, 0x00020000);
  bf16x8 bfrag[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) bfrag[ks] = *reinterpret_cast<const bf16x8*>(a.wf + ((size_t)ks * 64 + lane) * 8);
  const long nblocks = (a.npix_in + 15) / 16;
  const long stride = (long)gridDim.x * 4;
  const unsigned rowB = (unsigned)a.C * 2u;
  for (long b = (long)blockIdx.x * 4 + wave; b < nblocks; b += stride) {
    const long p = b * 16 + i16;
    const unsigned base = p < a.npix_in ? (unsigned)p * rowB + (unsigned)kq * 16u : kOOB;
    u32x4 af[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) af[ks] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(base == kOOB ? kOOB : base + (unsigned)ks * 64u), 0, 0);
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) acc = p2phd_mfma_16x16x32(*reinterpret_cast<const bf16x8*>(&af[ks]), bfrag[ks], acc);
    // D layout: lane (column = tap i16, row group kq) holds pixels 4 kq + j: 64-byte rows of P
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const long q = b * 16 + 4 * kq + j;
      if (q < a.npix_in) a.part[q * 16 + i16] = acc[j];
    }
  }
}

// ---- forward, pass 2: y[n, ho, wo, 0] = bias + sum over the 16 taps of the shifted partials; pad channels 1..7 = 0 -----------
__global__ __launch_bounds__(256) void dlast_fwd_gather_kernel(const float* __restrict__ part, const float* __restrict__ bias, bf16_t* __restrict__ y,
                                                               int N, int H, int W, int Ho, int Wo) {
  const long total = (long)N * Ho * Wo;
  for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < total; p += (long)gridDim.x * blockDim.x) {
    const int n = (int)(p / ((long)Ho * Wo));
    const int r = (int)(p - (long)n * Ho * Wo);
    const int ho = r / Wo, wo = r - ho * Wo;
    float s = bias ? bias[0] : 0.f;
#pragma unroll
    for (int th = 0; th < 4; ++th) {
      const int hi = ho + th - 2;
#pragma unroll
      for (int tw = 0; tw < 4; ++tw) {
        const int wi = wo + tw - 2;
        if (hi >= 0 && hi < H && wi >= 0 && wi < W) s += part[((size_t)(n * H + hi) * W + wi) * 16 + th * 4 + tw];
      }
    }
    bf16_t v[8];
    v[0] = (bf16_t)s;
#pragma unroll
    for (int e = 1; e < 8; ++e) v[e] = (bf16_t)0.f;
    *reinterpret_cast<uint4*>(y + (size_t)p * 8) = *reinterpret_cast<const uint4*>(v);
  }
}

// ---- input gradient -----------------------------------------------------------------------------------------------------------
struct DLastGradArgs {
  const bf16_t* dy;       // [N, Ho, Wo, 8] (channel 0)
  const bf16_t* wg;
  const bf16_t* addend;   // [N, H, W, C] or nullptr
  bf16_t* dx;             // [N, H, W, C]
  int N, H, W, Ho, Wo, C;
  int bpw, slots;         // 16-pixel blocks per wave, waves (= partial slots) per sample and channel quarter
  // fused first pass of the producer's InstanceNorm backward (nullptr: off)
  const bf16_t* bs_y;     // producer's pre-normalisation output [N, H, W, C]
  const float* bs_stats;  // [N][C][2] (mean, M2)
  float* bs_out;          // [N][slots][C][2]
  float bs_inv_hw, bs_eps, bs_slope;
};

__global__ __launch_bounds__(256) void dlast_dgrad_kernel(const DLastGradArgs a) {
  __shared__ __attribute__((aligned(16))) char stage_all[4 * 16 * kStagePitchG];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i16 = lane & 15, kq = lane >> 4;
  char* stage = stage_all + wave * 16 * kStagePitchG;
  // wave -> (sample n, slot, channel quarter cq)
  const int quarters = a.C / kQCH;
  long wid = (long)blockIdx.x * 4 + wave;
  const int cq = (int)(wid % quarters); wid /= quarters;
  const int slot = (int)(wid % a.slots);
  const int n = (int)(wid / a.slots);
  if (n >= a.N) return;
  const int HW = a.H * a.W;
  const int c0 = cq * kQCH;
  bf16x8 bfrag[kQCH / 16];
#pragma unroll
  for (int nb = 0; nb < kQCH / 16; ++nb) bfrag[nb] = *reinterpret_cast<const bf16x8*>(a.wg + ((size_t)(c0 / 16 + nb) * 64 + lane) * 8);
  // store phase: lane = (pixel row group lane >> 4: rows 4 u + (lane >> 4), piece column lane & 15 = channels c0 + 8 (lane & 15) ..)
  const int pc = lane & 15, rg = lane >> 4;
  const int cch = c0 + 8 * pc;
  float a1[8], a2[8], mean_b[8], rstd_b[8];
  const bool sums = a.bs_out != nullptr;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    a1[e] = a2[e] = 0.f;
    mean_b[e] = 0.f; rstd_b[e] = 0.f;
    if (sums) {
      const float2 ms = *reinterpret_cast<const float2*>(a.bs_stats + 2 * ((size_t)n * a.C + cch + e));
      mean_b[e] = ms.x;
      rstd_b[e] = rsqrtf(fmaxf(ms.y * a.bs_inv_hw, 0.f) + a.bs_eps);
    }
  }
  const bf16_t* dyn = a.dy + (size_t)n * a.Ho * a.Wo * 8;
  const size_t sample_off = (size_t)n * HW * a.C;
  for (int bi = 0; bi < a.bpw; ++bi) {
    const int p0 = (slot * a.bpw + bi) * 16;                  // first pixel of the block inside the sample
    if (p0 >= HW) break;
    // A fragment: this lane's pixel p0 + i16, taps 8 kq .. 8 kq + 7 (kq < 2), gathered from dy
    bf16x8 af;
    {
      const int p = p0 + i16;
      const int hi = p / a.W, wi = p - hi * a.W;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int tap = 8 * kq + e, th = tap >> 2, tw = tap & 3;
        const int ho = hi - th + 2, wo = wi - tw + 2;
        const bool ok = kq < 2 && p < HW && ho >= 0 && ho < a.Ho && wo >= 0 && wo < a.Wo;
        af[e] = ok ? dyn[((size_t)ho * a.Wo + wo) * 8] : (bf16_t)0.f;
      }
    }
#pragma unroll
    for (int nb = 0; nb < kQCH / 16; ++nb) {
      f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
      acc = p2phd_mfma_16x16x32(af, bfrag[nb], acc);
      // D layout: lane (column = channel 16 nb + i16, row group kq) holds pixels 4 kq + j
#pragma unroll
      for (int j = 0; j < 4; ++j) *reinterpret_cast<bf16_t*>(stage + (4 * kq + j) * kStagePitchG + (nb * 16 + i16) * 2) = (bf16_t)acc[j];
    }
    __builtin_amdgcn_wave_barrier();
    // 16 rows x 16 pieces: 4 passes of 64 lanes; the lane keeps its piece column, so channel constants and sums stay in registers
    uint4 yv[4], av[4];
    bool ok[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int row = 4 * u + rg;
      ok[u] = p0 + row < HW;
      const size_t off = sample_off + (size_t)(ok[u] ? p0 + row : p0) * a.C + cch;     // clamped: always loadable
      if (sums) yv[u] = *reinterpret_cast<const uint4*>(a.bs_y + off);
      if (a.addend != nullptr) av[u] = *reinterpret_cast<const uint4*>(a.addend + off);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int row = 4 * u + rg;
      uint4 v = *reinterpret_cast<const uint4*>(stage + row * kStagePitchG + pc * 16);
      if (a.addend != nullptr) {
        bf16_t* vv = reinterpret_cast<bf16_t*>(&v);
        const bf16_t* aa = reinterpret_cast<const bf16_t*>(&av[u]);
#pragma unroll
        for (int e = 0; e < 8; ++e) vv[e] = (bf16_t)((float)vv[e] + (float)aa[e]);
      }
      if (!ok[u]) continue;
      *reinterpret_cast<uint4*>(a.dx + sample_off + (size_t)(p0 + row) * a.C + cch) = v;
      if (sums) {
        const bf16_t* gg = reinterpret_cast<const bf16_t*>(&v);     // the ROUNDED gradient: what the apply pass will read
        const bf16_t* yy = reinterpret_cast<const bf16_t*>(&yv[u]);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float yh = ((float)yy[e] - mean_b[e]) * rstd_b[e];
          const float gp = (float)gg[e] * (yh > 0.f ? 1.f : a.bs_slope);
          a1[e] += gp; a2[e] += gp * yh;
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
  if (sums) {
    // fold the four row groups of a piece column in a fixed order (lanes pc, pc + 16, pc + 32, pc + 48), then one plain store per
    // channel into the wave's slot of the partial table: no atomics, bit-reproducible
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float s1 = a1[e], s2 = a2[e];
      s1 += __shfl_xor(s1, 16); s2 += __shfl_xor(s2, 16);
      s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
      if (rg == 0) *reinterpret_cast<float2*>(a.bs_out + (((size_t)n * a.slots + slot) * a.C + cch + e) * 2) = make_float2(s1, s2);
    }
  }
}

}  // namespace

namespace p2phd {

int g_opt_dlast = 1;

bool dlast_ok(const p2phd_conv_desc* c, bool ignore_option) {
  return (ignore_option || g_opt_dlast != 0) && c->dtype == P2PHD_BF16 && !c->transposed && c->K == 1 && c->C % kQCH == 0 && c->C >= kQCH &&
         c->C <= 32 * kMaxKS && c->R == 4 && c->S == 4 && c->stride == 1 && c->pad == 2 && c->pad_mode == 0 &&
         (size_t)c->N * c->H * c->W * c->C * 2 < 0xFFFFFFF0ull;
}

// which: 0 = forward fragments [C / 32][64][8], 1 = input-gradient fragments [C / 16][64][8]
size_t dlast_packed_elems(const p2phd_conv_desc* c, int which) { return (size_t)(which == 0 ? c->C / 32 : c->C / 16) * 64 * 8; }

int dlast_pack(const p2phd_conv_desc* c, int which, const float* w, void* wfrag, hipStream_t st) {
  const int total = (which == 0 ? c->C / 32 : c->C / 16) * 64;
  hipLaunchKernelGGL(dlast_pack_kernel, dim3((total + 255) / 256), dim3(256), 0, st, w, which == 0 ? (bf16_t*)wfrag : nullptr,
                     which == 1 ? (bf16_t*)wfrag : nullptr, c->C);
  return check_launch("dlast_pack");
}

size_t dlast_fwd_workspace_floats(const p2phd_conv_desc* c) { return (size_t)c->N * c->H * c->W * 16; }

// y [N, Ho, Wo, 8] = conv4x4 p2 (x [N, H, W, C]) + bias (channel 0; pad channels zero); part: dlast_fwd_workspace_floats of scratch
int dlast_fwd(const p2phd_conv_desc* c, const void* x, const void* wf, const float* bias, void* y, float* part, hipStream_t st) {
  DLastFwdArgs a{};
  a.x = (const bf16_t*)x; a.wf = (const bf16_t*)wf; a.part = part;
  a.npix_in = (long)c->N * c->H * c->W; a.C = c->C;
  a.x_bytes = (unsigned)((size_t)a.npix_in * c->C * 2);
  const long nblocks = (a.npix_in + 15) / 16;
  const int cus = g_opt_cus > 0 ? g_opt_cus : device_cus();
  const unsigned wgs = (unsigned)std::max<long>(1, std::min<long>((nblocks + 3) / 4, (long)cus * 3));
  switch (c->C / 32) {
    case 4: hipLaunchKernelGGL(dlast_fwd_partial_kernel<4>, dim3(wgs), dim3(256), 0, st, a); break;
    case 8: hipLaunchKernelGGL(dlast_fwd_partial_kernel<8>, dim3(wgs), dim3(256), 0, st, a); break;
    case 12: hipLaunchKernelGGL(dlast_fwd_partial_kernel<12>, dim3(wgs), dim3(256), 0, st, a); break;
    case 16: hipLaunchKernelGGL(dlast_fwd_partial_kernel<16>, dim3(wgs), dim3(256), 0, st, a); break;
    default: set_error("dlast_fwd: unsupported channel count %d", c->C); return P2PHD_EUNSUPPORTED;
  }
  if (int rc = check_launch("dlast_fwd(partials)")) return rc;
  const int Ho = c->H + 2 * c->pad - c->R + 1, Wo = c->W + 2 * c->pad - c->S + 1;
  const long total = (long)c->N * Ho * Wo;
  hipLaunchKernelGGL(dlast_fwd_gather_kernel, dim3((unsigned)std::min<long>((total + 255) / 256, 4096)), dim3(256), 0, st, (const float*)part, bias,
                     (bf16_t*)y, c->N, c->H, c->W, Ho, Wo);
  return check_launch("dlast_fwd(gather)");
}

// geometry of the input-gradient launch: 16-pixel blocks per wave and waves (partial slots) per sample
void dlast_dgrad_plan(const p2phd_conv_desc* c, int* bpw, int* slots) {
  const int nblk = (c->H * c->W + 15) / 16;
  const int b = nblk >= 64 ? 8 : (nblk >= 16 ? 4 : 1);
  if (bpw) *bpw = b;
  if (slots) *slots = (nblk + b - 1) / b;
}

size_t dlast_bsum_table_floats(const p2phd_conv_desc* c) {
  int slots = 0;
  dlast_dgrad_plan(c, nullptr, &slots);
  return (size_t)c->N * slots * c->C * 2;
}

// dx [N, H, W, C] = input gradient (+ addend); bs_out != nullptr: partial sums [N][slots][C][2] of the producer's InstanceNorm backward
int dlast_dgrad(const p2phd_conv_desc* c, const void* dy, const void* wg, const void* addend, void* dx, const void* bs_y,
                const float* bs_stats, float* bs_out, float bs_inv_hw, float bs_eps, float bs_slope, hipStream_t st) {
  DLastGradArgs a{};
  a.dy = (const bf16_t*)dy; a.wg = (const bf16_t*)wg; a.addend = (const bf16_t*)addend; a.dx = (bf16_t*)dx;
  a.N = c->N; a.H = c->H; a.W = c->W; a.C = c->C;
  a.Ho = c->H + 2 * c->pad - c->R + 1; a.Wo = c->W + 2 * c->pad - c->S + 1;
  dlast_dgrad_plan(c, &a.bpw, &a.slots);
  a.bs_y = (const bf16_t*)bs_y; a.bs_stats = bs_stats; a.bs_out = bs_out;
  a.bs_inv_hw = bs_inv_hw; a.bs_eps = bs_eps; a.bs_slope = bs_slope;
  const long waves = (long)c->N * a.slots * (c->C / kQCH);
  hipLaunchKernelGGL(dlast_dgrad_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, st, a);
  return check_launch("dlast_dgrad");
}

}  // namespace p2phd
