// The discriminator's FIRST layer, forward: Conv2d(input_nc + output_nc = 4, ndf = 64, kernel 4, stride 2, padding 2) + LeakyReLU(0.2)
// (models/networks.py:342-344, once per scale and per discriminator pass), 16-bit storage.
//
// 17 GFLOP on 0.4 GB at configs[1] (2B = 64 samples: 134 MB in -- 4 channels at channel pitch 8 --, 272 MB out): HBM-bound,
// 81 us at 5 TB/s.  As a gather-GEMM (128 x 64 tiles with TWO K slabs each: 16 taps x channel pitch 8) it ran at 1.8 TB/s: 33 k tiles
// that are all prologue and epilogue (profiles/r04_step_kernel_trace_summary.csv: 236 us).
// Here a WAVEFRONT owns 16 consecutive output pixels (of the flattened [N, Ho, Wo] index: consecutive NHWC rows of 128 bytes):
//   * K is ordered (kernel row th, kernel column tw, channel c8): the 8 channel slots of an input pixel are ONE 16-byte load, so
//     the A fragment of v_mfma_f32_16x16x32_bf16 for kernel row th -- lane (pixel i, k-quarter tw) -- is the input pixel
//     (2 ho + th - 2, 2 wo + tw - 2), fetched straight into registers by one buffer load per lane (zero padding = the buffer's
//     range check); the four pad channels multiply packed zeros.  No LDS on the operand path, no folded copy;
//   * the 64 x 128 weight matrix lives in registers as B fragments for the life of the wave (64 VGPRs);
//   * bias + activation on the accumulators, the 16 x 64 output block is transposed through a 2.3 KB LDS patch of the wave's own
//     and leaves as two 16-byte pieces per lane: 1 KB of contiguous NHWC per store instruction;
//   * the next block's four loads are issued before the current block's MFMAs (12 waves per CU keep ~50 KB in flight).
#include "convplan.h"

namespace {

using namespace p2phd;

typedef p2phd_h16 bf16_t;
typedef __attribute__((ext_vector_type(8))) bf16_t bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

constexpr unsigned kOOB = 0xFFFFFFF0u;
constexpr int kStagePitch = 64 * 2 + 16;                       // bytes per staged pixel row (16-byte pad: the 2-byte column writes of 4 pixels spread over banks)

struct DFirstArgs {
  const bf16_t* x;        // [N, H, W, 8]
  const bf16_t* wf;       // fragment-ordered weights (dfirst_pack_kernel)
  const float* bias;      // [64] or nullptr
  bf16_t* y;              // [N, Ho, Wo, 64]
  int N, H, W, Ho, Wo, act;
  long npix;              // N * Ho * Wo
  unsigned in_bytes;
};

// wf[((nb * 4 + th) * 64 + lane) * 8 + e] = w[k = 16 nb + (lane & 15)][c = e][th][tw = lane >> 4], zero for the pad channels e >= C
__global__ void dfirst_pack_kernel(const float* __restrict__ w, bf16_t* __restrict__ wf, int K, int C) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (K / 16) * 4 * 64) return;
  const int lane = idx & 63, th = (idx >> 6) & 3, nb = idx >> 8;
  const int k = 16 * nb + (lane & 15), tw = lane >> 4;
  bf16_t v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = (bf16_t)(e < C ? w[((size_t)(k * C + e) * 4 + th) * 4 + tw] : 0.f);
  *reinterpret_cast<uint4*>(wf + (size_t)idx * 8) = *reinterpret_cast<const uint4*>(v);
}

template <int NBLK>
__global__ __launch_bounds__(256) void dfirst_fwd_kernel(const DFirstArgs a) {
  __shared__ __attribute__((aligned(16))) char stage_all[4 * 16 * kStagePitch];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i16 = lane & 15, kq = lane >> 4;
  char* stage = stage_all + wave * 16 * kStagePitch;
  const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, (int)a.in_bytes, 0x00020000);

  bf16x8 bfrag[NBLK][4];
#pragma unroll
  for (int nb = 0; nb < NBLK; ++nb)
#pragma unroll
    for (int th = 0; th < 4; ++th) bfrag[nb][th] = *reinterpret_cast<const bf16x8*>(a.wf + ((size_t)(nb * 4 + th) * 64 + lane) * 8);
  float bv[NBLK];
#pragma unroll
  for (int nb = 0; nb < NBLK; ++nb) bv[nb] = a.bias ? a.bias[nb * 16 + i16] : 0.f;
  const float slope = a.act == P2PHD_ACT_RELU ? 0.f : (a.act == P2PHD_ACT_LRELU ? 0.2f : 1.f);

  const long nblocks = (a.npix + 15) / 16;
  const long stride = (long)gridDim.x * 4;
  const int HoWo = a.Ho * a.Wo;
  // byte offsets of this lane's four input pixels (kernel rows 0..3 at kernel column kq) for the block's pixel i16
  auto offsets = [&](long b, unsigned* off) {
    const long p = b * 16 + i16;
    const bool live = p < a.npix;
    const int n = (int)(p / HoWo);
    const int r = (int)(p - (long)n * HoWo);
    const int ho = r / a.Wo, wo = r - ho * a.Wo;
    const int wi = 2 * wo + kq - 2;
    const bool wok = live && wi >= 0 && wi < a.W;
#pragma unroll
    for (int th = 0; th < 4; ++th) {
      const int hi = 2 * ho + th - 2;
      off[th] = (wok && hi >= 0 && hi < a.H) ? (unsigned)(((n * a.H + hi) * a.W + wi)) * 16u : kOOB;
    }
  };
  long b = (long)blockIdx.x * 4 + wave;
  u32x4 cur[4], nxt[4];
  if (b < nblocks) {
    unsigned off[4];
    offsets(b, off);
#pragma unroll
    for (int th = 0; th < 4; ++th) cur[th] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off[th], 0, 0);
  }
  for (; b < nblocks; b += stride) {
    const long bn = b + stride;
    if (bn < nblocks) {                                         // the next block's pixels: in flight while this block computes
      unsigned off[4];
      offsets(bn, off);
#pragma unroll
      for (int th = 0; th < 4; ++th) nxt[th] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off[th], 0, 0);
    }
    f32x4 acc[NBLK];
#pragma unroll
    for (int nb = 0; nb < NBLK; ++nb) acc[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int th = 0; th < 4; ++th) {
      const bf16x8 af = *reinterpret_cast<const bf16x8*>(&cur[th]);
#pragma unroll
      for (int nb = 0; nb < NBLK; ++nb) acc[nb] = p2phd_mfma_16x16x32(af, bfrag[nb][th], acc[nb]);
    }
    // D layout: lane (column = channel i16 of block nb, row group kq) holds pixels 4 kq + j.  Bias, activation, 16-bit rounding,
    // then through the wave's LDS patch into whole 16-byte pieces of NHWC rows
#pragma unroll
    for (int nb = 0; nb < NBLK; ++nb)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float v = acc[nb][j] + bv[nb];
        v = v > 0.f ? v : slope * v;
        *reinterpret_cast<bf16_t*>(stage + (4 * kq + j) * kStagePitch + (nb * 16 + i16) * 2) = (bf16_t)v;
      }
    // (the patch is private to the wave: its LDS operations execute in order, no workgroup barrier; the wave barrier only keeps
    // the compiler from moving the reads above the writes)
    __builtin_amdgcn_wave_barrier();
    const long p0 = b * 16;
#pragma unroll
    for (int u = 0; u < (NBLK * 16 * 2 / 16) * 16 / 64; ++u) {   // 16 pixels x (NBLK * 2) pieces over 64 lanes
      const int q = u * 64 + lane;
      const int px = q / (NBLK * 2), pc = q - px * (NBLK * 2);
      const uint4 v = *reinterpret_cast<const uint4*>(stage + px * kStagePitch + pc * 16);
      if (p0 + px < a.npix) *reinterpret_cast<uint4*>(a.y + ((size_t)(p0 + px) * (NBLK * 16) + pc * 8)) = v;
    }
    __builtin_amdgcn_wave_barrier();
    if (bn < nblocks) {
#pragma unroll
      for (int th = 0; th < 4; ++th) cur[th] = nxt[th];
    }
  }
}

}  // namespace

namespace p2phd {

int g_opt_dfirst = 1;

bool dfirst_ok(const p2phd_conv_desc* c, bool ignore_option) {
  return (ignore_option || g_opt_dfirst != 0) && c->dtype == P2PHD_BF16 && !c->transposed && c->C >= 1 && c->C <= 8 && c->K == 64 &&
         c->R == 4 && c->S == 4 && c->stride == 2 && c->pad == 2 && c->pad_mode == 0 && c->H >= 2 && c->W >= 2 &&
         (size_t)c->N * c->H * c->W * 16 < 0xFFFFFFF0ull;
}

size_t dfirst_packed_elems(const p2phd_conv_desc* c) { return (size_t)(c->K / 16) * 4 * 64 * 8; }

int dfirst_pack(const p2phd_conv_desc* c, const float* w, void* wf, hipStream_t st) {
  const int total = (c->K / 16) * 4 * 64;
  hipLaunchKernelGGL(dfirst_pack_kernel, dim3((total + 255) / 256), dim3(256), 0, st, w, (bf16_t*)wf, c->K, c->C);
  return check_launch("dfirst_pack");
}

// y [N, Ho, Wo, 64] = act(conv4x4 s2 p2 (x [N, H, W, 8]) + bias)
int dfirst_fwd(const p2phd_conv_desc* c, const void* x, const void* wf, const float* bias, int act, void* y, hipStream_t st) {
  DFirstArgs a{};
  a.x = (const bf16_t*)x; a.wf = (const bf16_t*)wf; a.bias = bias; a.y = (bf16_t*)y;
  a.N = c->N; a.H = c->H; a.W = c->W;
  a.Ho = (c->H + 2 * c->pad - c->R) / c->stride + 1; a.Wo = (c->W + 2 * c->pad - c->S) / c->stride + 1;
  a.act = act;
  a.npix = (long)c->N * a.Ho * a.Wo;
  a.in_bytes = (unsigned)((size_t)c->N * c->H * c->W * 16);
  const long nblocks = (a.npix + 15) / 16;
  const int cus = g_opt_cus > 0 ? g_opt_cus : device_cus();
  const long wgs = std::min<long>((nblocks + 3) / 4, (long)cus * 3);      // 3 workgroups of 4 waves per CU (130 VGPRs), grid-stride over the blocks
  hipLaunchKernelGGL(dfirst_fwd_kernel<4>, dim3((unsigned)std::max<long>(wgs, 1)), dim3(256), 0, st, a);
  return check_launch("dfirst_fwd");
}

}  // namespace p2phd
