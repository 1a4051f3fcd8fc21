// "Marching" kernels for the stride-2 3x3 layers at the OUTER end of the generator (models/networks.py:194-195 Conv2d(ngf, 2 ngf,
// 3, stride 2, padding 1) + InstanceNorm + ReLU and :205-206 ConvTranspose2d(2 ngf, ngf, 3, stride 2, padding 1,
// output_padding 1), ngf = 48 at configs[1]) and for their input gradients, bf16.
//
// These four launches per step move 0.6 GB each for 87 GFLOP: they are HBM-bound (120 us at 5 TB/s), and as tiles of the
// generic gather-GEMM they ran at a third of that (profiles/r03_outer_layer_probe.log: 24 k of a tile's 36 k cycles are gather
// table, ring fill, statistics and store loop, with one workgroup per CU and nothing to overlap them with).  Here:
//   * a workgroup (12 waves) owns a COLUMN STRIP of one sample and marches down it, one output row per step; the input rows
//     it needs live in a 5-row LDS ring, so every input byte is fetched ONCE (two new rows per step, coalesced 16-byte
//     loads issued a whole step before they are written to LDS), and the address arithmetic of a step is a handful of adds
//     -- no gather table;
//   * the packed weights never touch LDS: every wave keeps the B fragments of ITS 16 output channels for all 9 taps in
//     registers for the life of the workgroup (60 VGPRs at 48 input channels), A fragments come from the ring with
//     conflict-free ds_read_b128 (pixel pitch padded by 16 bytes), v_mfma_f32_16x16x32_bf16;
//   * the output tile of a step is staged in LDS and leaves one step later as whole 16-byte pieces of contiguous NHWC rows
//     (one piece per thread), so ONE barrier per step orders ring writes, staging and stores.  (The staging stores are where the
//     kernel's SQ_LDS_BANK_CONFLICT count comes from -- lanes 2 m and 2 m + 1 write the two 16-bit halves of one dword -- , not the
//     fragment reads; a variant that packs the pair into one dword store through DPP changed the counter and not the time: the
//     kernel waits for HBM.  DESIGN.md section 6, round 5);
//   * InstanceNorm partials: every lane keeps running (count, mean, M2) of its channel over the whole march (Chan's
//     update per step), merged across the four lanes of a channel at the end: one table slot per wave and launch, no atomics;
//   * input-gradient launches can carry the producer's InstanceNorm-backward sums (the fused form of conv.hip's store
//     loop): the storing thread owns one 8-channel column for the whole march, so the sums stay in registers.
#include "convplan.h"

namespace {

using namespace p2phd;

typedef p2phd_h16 bf16_t;                 // the library's 16-bit storage type: bf16, or fp16 in the -DP2PHD_F16 build (common.h)
typedef __attribute__((ext_vector_type(8))) bf16_t bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

constexpr int kThreads = 768, kWaves = 12;
constexpr unsigned kOOB = 0xFFFFFFF0u;

__device__ __forceinline__ void chan_merge(float& na, float& ma, float& qa, float nb, float mb, float qb) {
  const float n = na + nb;
  const float d = mb - ma, f = n > 0.f ? nb / n : 0.f;
  ma += d * f;
  qa += qb + d * d * na * f;
  na = n;
}

struct MarchArgs {
  const bf16_t* in;       // [N, Hin, Win, CI]
  const bf16_t* wf;       // fragment-ordered weights (march_pack_kernel)
  const float* bias;      // [CO] or nullptr
  bf16_t* out;            // [N, Ho, Wo, CO]
  float* table;           // InstanceNorm partials [N][slots][CO][2] or nullptr
  int N, Hin, Win, Ho, Wo;
  int strips, nseg, seg_rows, slots;
  unsigned in_bytes;
  // fused first pass of the PRODUCER's InstanceNorm backward (input-gradient launches; conv.hip GDesc::bs_*)
  const bf16_t* bs_y;     // pre-normalisation output of the producer, shape of `out`
  const float* bs_stats;  // [N][CO][2] (mean, M2)
  float* bs_out;          // [N][strips * nseg][CO][2] or nullptr
  float bs_inv_hw, bs_eps, bs_slope;
  // LAZY input (round 4): `in` is the RAW output of an InstanceNorm block; this kernel applies (x - mean) * rstd and the
  // activation while it stages rows into LDS -- the values p2phd_instnorm_act_fwd would have written, bit for bit, without
  // that pass over the plane.  in_stats [N][CI][2] (mean, M2) of the producer; zeros of the padding stay zeros.
  const float* in_stats;
  float in_inv_hw, in_eps, in_slope;
};

// (mean, rstd) of the lazily normalised operand's channels of sample n -> LDS table [C][2]
__device__ __forceinline__ void lazy_table_fill(float* tab, const float* stats, int n, int Cc, float inv_hw, float eps, int tid) {
  if (tid < Cc) {
    const float2 ms = *reinterpret_cast<const float2*>(stats + 2 * ((size_t)n * Cc + tid));
    tab[2 * tid] = ms.x;
    tab[2 * tid + 1] = rsqrtf(fmaxf(ms.y * inv_hw, 0.f) + eps);
  }
}
// one 16-byte chunk (channels c0 .. c0 + 7) normalised + activated exactly as in_act_fwd_kernel does (norm.hip): (y - mean) * rstd
// in f32, then the activation, then ONE rounding to bf16; !ok -> zeros.  This is VALU work on every staged element, so it is
// kept short: bf16 -> f32 is a shift / mask on the packed dword, subtraction and multiplication go out as packed f32 pairs
// (v_pk_add_f32 / v_pk_mul_f32: same roundings as the scalar forms), ReLU is one max, the padding mask is applied to the four
// result dwords.  Table: [C][2] = (mean, rstd) interleaved, i.e. one float4 = two channels.
typedef __attribute__((ext_vector_type(2))) float f32x2;
__device__ __forceinline__ u32x4 lazy_norm8(u32x4 raw, const float* tab, int c0, float slope, bool ok) {
  u32x4 o;
#pragma unroll
  for (int h = 0; h < 4; ++h) {                                   // dword h = channels c0 + 2 h, c0 + 2 h + 1
    const f32x4 t = *reinterpret_cast<const f32x4*>(tab + 2 * (c0 + 2 * h));   // (mean0, rstd0, mean1, rstd1)
    const unsigned w = raw[h];
    f32x2 v;
    {
      float lo, hi;
      p2phd_unpack2(w, lo, hi);
      v = f32x2{lo, hi};
    }
    v = (v - f32x2{t[0], t[2]}) * f32x2{t[1], t[3]};
    if (slope == 0.f) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); }                 // (uniform branch)
    else { v[0] = v[0] > 0.f ? v[0] : slope * v[0]; v[1] = v[1] > 0.f ? v[1] : slope * v[1]; }
    typedef __attribute__((ext_vector_type(2))) bf16_t bf16x2;
    const bf16x2 r = {(bf16_t)v[0], (bf16_t)v[1]};
    o[h] = ok ? *reinterpret_cast<const unsigned*>(&r) : 0u;
  }
  return o;
}

// geometry of one (CI, CO, WS) instance of the stride-2 gather ("S") kernel
template <int CI, int CO, int WS>
struct SGeom {
  static constexpr int CQ = CI / 8;                         // 16-byte chunks per input pixel
  static constexpr int KSR = (3 * CQ + 3) / 4;              // k-steps (32 channels-of-taps) per tap ROW; pad chunks carry zero weights
  static constexpr int KS = 3 * KSR;
  static constexpr int NWN = CO / 16, NWM = kWaves / NWN;   // waves over output channels / over the strip's pixels
  static constexpr int MBW = WS / 16 / NWM;                 // 16-pixel blocks per wave and step
  static constexpr int RW = 2 * WS + 1;                     // input pixels per ring row
  static constexpr int PXB = CI * 2 + 16;                   // ring pixel pitch (bytes): stride-2 fragment reads hit 16 distinct 16-byte slots
  static constexpr int ROWB = RW * PXB;
  static constexpr int RING = 5 * ROWB;
  static constexpr int SPXB = CO * 2 + 16;                  // staging pixel pitch
  static constexpr int STAGEB = WS * SPXB;
  static constexpr int TABB = CI * 2 * 4;                   // (mean, rstd) of a lazily normalised input
  static constexpr int LDS = RING + 2 * STAGEB + TABB;
  static constexpr int NCH = 2 * RW * CQ;                   // chunks of the two new rows of a step
  static constexpr int LPT = (NCH + kThreads - 1) / kThreads;
  static_assert(CI % 16 == 0 && CO % 16 == 0 && kWaves % NWN == 0 && (WS / 16) % NWM == 0, "march: wave grid");
  static_assert(WS * (CO / 8) == kThreads, "march: one staged piece per thread");
  static_assert(ROWB % 16 == 0 && (3 * CQ) % 4 != 1 && (3 * CQ) % 4 != 3, "march: pad chunks come in pairs");
};

// wf[((wn * KS + i) * 64 + lane) * 8 + e] = w[o = 16 wn + (lane & 15)][ci = 8 q + e][r][s3] for k-step i = r * KSR + ii, chunk
// c = 4 ii + (lane >> 4) of tap row r (s3 = c / CQ, q = c % CQ), zero for the pad chunks c >= 3 CQ.
// Master tensor: element (o, ci, r, s) at o * s_o + ci * s_i + r * 3 + s -- Conv2d weight [K][C][3][3] (o = K) for the forward,
// ConvTranspose2d weight [Cin][Cout][3][3] (o = Cin) for its input gradient: the same index pattern.
template <int CI, int CO>
__global__ void march_s_pack_kernel(const float* __restrict__ w, bf16_t* __restrict__ wf, long s_o, long s_i) {
  constexpr int CQ = CI / 8, KSR = (3 * CQ + 3) / 4, KS = 3 * KSR;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (CO / 16) * KS * 64) return;
  const int lane = idx & 63, i = (idx >> 6) % KS, wn = (idx >> 6) / KS;
  const int o = 16 * wn + (lane & 15), kq = lane >> 4;
  const int r = i / KSR, c = 4 * (i % KSR) + kq;
  bf16_t v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    float f = 0.f;
    if (c < 3 * CQ) {
      const int s3 = c / CQ, q = c % CQ;
      f = w[(long)o * s_o + (long)(8 * q + e) * s_i + r * 3 + s3];
    }
    v[e] = (bf16_t)f;
  }
  *reinterpret_cast<uint4*>(wf + (size_t)idx * 8) = *reinterpret_cast<const uint4*>(v);
}

// out[n, ho, wo, k] = sum_{r,s,c} in[n, 2 ho + r - 1, 2 wo + s - 1, c] * W[k][c][r][s]   (zeros outside the image)
template <int CI, int CO, int WS, bool BSUM, bool LAZY>
__global__ __launch_bounds__(kThreads) void march_s_kernel(const MarchArgs a) {
  typedef SGeom<CI, CO, WS> G;
  constexpr int CQ = G::CQ, KSR = G::KSR, KS = G::KS, NWN = G::NWN, MBW = G::MBW, RW = G::RW, PXB = G::PXB, ROWB = G::ROWB;
  constexpr int SPXB = G::SPXB, STAGEB = G::STAGEB, LPT = G::LPT, NCH = G::NCH;
  extern __shared__ float4 smem_raw[];
  char* smem = reinterpret_cast<char*>(smem_raw);
  char* stage = smem + G::RING;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave % NWN, wm = wave / NWN;
  const int n16 = lane & 15, kq = lane >> 4;

  // workgroup -> (sample, strip, segment)
  int wg = (int)blockIdx.x;
  const int seg = wg % a.nseg; wg /= a.nseg;
  const int strip = wg % a.strips;
  const int n = wg / a.strips;
  const int wo0 = strip * WS, h_first = seg * a.seg_rows;
  const int Hin = a.Hin, Win = a.Win;

  // B fragments of this wave's 16 output channels: resident for the whole march
  bf16x8 bfrag[KS];
#pragma unroll
  for (int i = 0; i < KS; ++i)
    bfrag[i] = *reinterpret_cast<const bf16x8*>(a.wf + ((size_t)(wn * KS + i) * 64 + lane) * 8);
  const float bias_n = a.bias != nullptr ? a.bias[16 * wn + n16] : 0.f;

  // A-fragment byte offsets inside a ring row, per k-step of a tap row: lane (m = n16, kq) reads chunk c = 4 ii + kq of pixel
  // 2 m + s3 (pad chunks re-read the chunk two to the left: finite data under zero weights)
  unsigned aoff[KSR];
#pragma unroll
  for (int ii = 0; ii < KSR; ++ii) {
    int c = 4 * ii + kq;
    if (c >= 3 * CQ) c -= 2;
    const int s3 = c / CQ, q = c - s3 * CQ;
    aoff[ii] = (unsigned)((2 * n16 + s3) * PXB + q * 16 + wm * MBW * 32 * PXB);
  }

  // the two new input rows of a step, as 16-byte chunks: chunk j = tid + 768 u -> (row rr, pixel x, channel chunk q)
  const auto rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, (int)a.in_bytes, 0x00020000);
  unsigned colB[LPT], lw[LPT];
  int rr_of[LPT], q_of[LPT];
#pragma unroll
  for (int u = 0; u < LPT; ++u) {
    const int j = tid + kThreads * u;
    const int rr = j / (RW * CQ), rem = j - rr * (RW * CQ);
    const int x = rem / CQ, q = rem - x * CQ;
    const int wi = 2 * wo0 - 1 + x;
    const bool ok = j < NCH && wi >= 0 && wi < Win;
    rr_of[u] = rr; q_of[u] = q;
    colB[u] = ok ? (unsigned)((wi * CI + 8 * q) * 2) : kOOB;
    lw[u] = (unsigned)(min(x, RW - 1) * PXB + q * 16);
  }
  float* in_tab = reinterpret_cast<float*>(smem + G::RING + 2 * STAGEB);
  if constexpr (LAZY) {
    lazy_table_fill(in_tab, a.in_stats, n, CI, a.in_inv_hw, a.in_eps, tid);
    __syncthreads();
  }
  const unsigned sampleB = (unsigned)((size_t)n * Hin * Win * CI * 2);
  const unsigned rowpitchB = (unsigned)(Win * CI * 2);
  u32x4 ld[LPT];
  // loads of abs rows (a_first, a_first + 1) of this segment: input row hi = 2 h_first - 1 + a
  auto issue_rows = [&](int a_first) {
#pragma unroll
    for (int u = 0; u < LPT; ++u) {
      const int hi = 2 * h_first - 1 + a_first + rr_of[u];
      const bool ok = hi >= 0 && hi < Hin && colB[u] != kOOB;
      const unsigned off = ok ? sampleB + (unsigned)hi * rowpitchB + colB[u] : kOOB;
      ld[u] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, (int)off, 0, 0);
    }
  };
  auto write_rows = [&](int a_first) {
#pragma unroll
    for (int u = 0; u < LPT; ++u) {
      const int slot = (a_first + rr_of[u] + 5) % 5;
      u32x4 v = ld[u];
      if constexpr (LAZY) {
        const int hi = 2 * h_first - 1 + a_first + rr_of[u];
        v = lazy_norm8(v, in_tab, 8 * q_of[u], a.in_slope, hi >= 0 && hi < Hin && colB[u] != kOOB);
      }
      if (tid + kThreads * u < NCH) *reinterpret_cast<u32x4*>(smem + slot * ROWB + lw[u]) = v;
    }
  };

  // prologue: abs rows -1 (unused), 0, then 1, 2
  issue_rows(-1);
  write_rows(-1);
  issue_rows(1);
  write_rows(1);
  __syncthreads();

  f32x4 acc[MBW];
  float st_n = 0.f, st_mean = 0.f, st_m2 = 0.f;               // running statistics of this lane's channel
  // fused InstanceNorm-backward sums: this thread stores piece column `pcol` of every row
  constexpr int CPR = CO / 8;
  const int ppx = tid / CPR, pcol = tid - ppx * CPR;
  float bs_a1[8], bs_a2[8], bs_mean[8], bs_rstd[8];
  constexpr bool bsum = BSUM;                                   // (its own instance: the sums cost 36 registers)
  if (bsum) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float2 ms = *reinterpret_cast<const float2*>(a.bs_stats + 2 * ((size_t)n * CO + 8 * pcol + e));
      bs_mean[e] = ms.x;
      bs_rstd[e] = rsqrtf(fmaxf(ms.y * a.bs_inv_hw, 0.f) + a.bs_eps);
      bs_a1[e] = bs_a2[e] = 0.f;
    }
  }
  const size_t out_sample = (size_t)n * a.Ho * a.Wo * CO;

  auto store_tile = [&](int s_prev, u32x4 yv) {
    // tile of step s_prev (staged one step ago): one 16-byte piece per thread, contiguous NHWC rows
    const u32x4 v = *reinterpret_cast<const u32x4*>(stage + (s_prev & 1) * STAGEB + ppx * SPXB + pcol * 16);
    const size_t o = out_sample + ((size_t)(h_first + s_prev) * a.Wo + wo0 + ppx) * CO + 8 * pcol;
    *reinterpret_cast<u32x4*>(a.out + o) = v;
    if (bsum) {
      const bf16_t* gg = reinterpret_cast<const bf16_t*>(&v);      // the ROUNDED gradient: what the apply pass will read
      const bf16_t* yy = reinterpret_cast<const bf16_t*>(&yv);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float yh = ((float)yy[e] - bs_mean[e]) * bs_rstd[e];
        const float gp = (float)gg[e] * (yh > 0.f ? 1.f : a.bs_slope);
        bs_a1[e] += gp; bs_a2[e] += gp * yh;
      }
    }
  };

  const int nsteps = a.seg_rows;
  int a0 = 0;                                                   // (2 s) % 5
  for (int s = 0; s < nsteps; ++s) {
    // (the producer's pre-normalisation piece of the row stored in this step: requested BEFORE the row loads, so that waiting
    // for it leaves those in flight -- vmcnt retires in issue order)
    u32x4 yv = {0u, 0u, 0u, 0u};
    if (bsum && s > 0)
      yv = *reinterpret_cast<const u32x4*>(a.bs_y + out_sample + ((size_t)(h_first + s - 1) * a.Wo + wo0 + ppx) * CO + 8 * pcol);
    if (s + 1 < nsteps) issue_rows(2 * s + 3);

    // ---- this step's output row: 16 channels x MBW pixel blocks per wave, K = 9 taps x CI ----
#pragma unroll
    for (int b = 0; b < MBW; ++b) acc[b] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      int slot = a0 + r;
      slot = slot >= 5 ? slot - 5 : slot;
      const char* rowp = smem + slot * ROWB;
#pragma unroll
      for (int ii = 0; ii < KSR; ++ii) {
#pragma unroll
        for (int b = 0; b < MBW; ++b) {
          const bf16x8 af = *reinterpret_cast<const bf16x8*>(rowp + aoff[ii] + b * 32 * PXB);
          acc[b] = p2phd_mfma_16x16x32(af, bfrag[r * KSR + ii], acc[b]);
        }
      }
    }

    // ---- bias, statistics, staging (lane = channel n16, registers = pixels 4 kq + e of block b) ----
    {
      float v[MBW][4];
      float sum = 0.f;
#pragma unroll
      for (int b = 0; b < MBW; ++b)
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[b][e] = acc[b][e] + bias_n; sum += v[b][e]; }
      if (a.table != nullptr) {
        // Chan's update with equal batches: after s steps the lane holds n_a = 4 MBW s values, the batch has n_b = 4 MBW, so
        // n_b / n = 1 / (s + 1) and n_a n_b / n = 4 MBW s / (s + 1): one reciprocal per step, none per lane
        const float mb = sum * (1.f / (4 * MBW));
        float q = 0.f;
#pragma unroll
        for (int b = 0; b < MBW; ++b)
#pragma unroll
          for (int e = 0; e < 4; ++e) { const float dlt = v[b][e] - mb; q += dlt * dlt; }
        const float f = __builtin_amdgcn_rcpf((float)(s + 1)), d = mb - st_mean;
        st_mean += d * f;
        st_m2 += q + d * d * ((float)(4 * MBW * s) * f);
        st_n = (float)(4 * MBW * (s + 1));
      }
      char* sp = stage + (s & 1) * STAGEB + (16 * wn + n16) * 2;
#pragma unroll
      for (int b = 0; b < MBW; ++b)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          *reinterpret_cast<bf16_t*>(sp + (16 * (wm * MBW + b) + 4 * kq + e) * SPXB) = (bf16_t)v[b][e];
    }

    if (s > 0) store_tile(s - 1, yv);
    if (s + 1 < nsteps) write_rows(2 * s + 3);
    __syncthreads();
    a0 += 2;
    a0 = a0 >= 5 ? a0 - 5 : a0;
  }
  {
    u32x4 yv = {0u, 0u, 0u, 0u};
    if (bsum)
      yv = *reinterpret_cast<const u32x4*>(a.bs_y + out_sample + ((size_t)(h_first + nsteps - 1) * a.Wo + wo0 + ppx) * CO + 8 * pcol);
    store_tile(nsteps - 1, yv);
  }

  if (a.table != nullptr) {
    // the four lanes (kq) of a channel -> one (sum, M2) partial per wave: slot = ((strip * nseg + seg) * NWM + wm)
    float nn = st_n, mm = st_mean, qq = st_m2;
#pragma unroll
    for (int o = 16; o < 64; o <<= 1) {
      const float n2 = __shfl_xor(nn, o), m2 = __shfl_xor(mm, o), q2 = __shfl_xor(qq, o);
      chan_merge(nn, mm, qq, n2, m2, q2);
    }
    if (kq == 0) {
      const int slot = (strip * a.nseg + seg) * G::NWM + wm;
      float* sp = a.table + 2 * (((size_t)n * a.slots + slot) * CO + 16 * wn + n16);
      sp[0] = mm * nn;
      sp[1] = qq;
    }
  }
  if (bsum) {
    // fold the sums of the WS threads of a piece column in a fixed order (no atomics): [tid][16] floats through the ring
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int e = 0; e < 8; ++e) { red[tid * 16 + e] = bs_a1[e]; red[tid * 16 + 8 + e] = bs_a2[e]; }
    __syncthreads();
    if (tid < 2 * CO) {
      const int ch = tid >> 1, which = tid & 1, pc = ch >> 3, e = ch & 7;
      float sum = 0.f;
      for (int p = 0; p < WS; ++p) sum += red[(p * CPR + pc) * 16 + which * 8 + e];
      a.bs_out[(((size_t)n * (a.strips * a.nseg) + strip * a.nseg + seg) * CO + ch) * 2 + which] = sum;
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------------
// "U" kernel: the transposed form -- out[n, 2 a + pi, 2 b + pj, k] = sum over the taps (r, s) of parity class (pi, pj) of
// in[n, a + dr, b + dc, c] * W[c][k][r][s]: ConvTranspose2d(CI, CO, 3, stride 2, padding 1, output_padding 1) forward, and the
// input gradient of Conv2d(CO, CI, 3, stride 2, padding 1).  Class (0,0) has one tap, (0,1) and (1,0) two, (1,1) four:
//   pi = 0: (dr 0, r 1);   pi = 1: (dr 0, r 2), (dr 1, r 0);   the same for (pj, dc, s).
// A workgroup marches down a strip of WS INPUT pixels: step a reads input rows a, a + 1 (3-row ring, one new row per step) and
// writes output rows 2 a, 2 a + 1 of 2 WS pixels.  12 waves = 3 channel blocks x 2 pixel groups x 2 ROLES: role A computes
// class (1,1) (4 taps), role B the other three classes (5 taps, the (0,0)-neighbour fragments shared by all three) -- 12 / 15
// B fragments resident per wave, no zero blocks multiplied.
// ------------------------------------------------------------------------------------------------------------------------
template <int CI, int CO, int WS>
struct UGeom {
  static constexpr int CQ = CI / 8;
  static constexpr int KN = CQ / 4;                         // k-steps per neighbour pixel
  static constexpr int NB = CO / 16;                        // channel blocks
  static constexpr int NWM = kWaves / (2 * NB);             // pixel groups
  static constexpr int MBW = WS / 16 / NWM;
  static constexpr int RW = WS + 1;
  static constexpr int PXB = CI * 2 + 32;                   // pixel pitch = 16 * 14 (mod 256): the 16-lane groups of ds_read_b128 hit 16 slots
  static constexpr int ROWB = RW * PXB;
  static constexpr int RING = 3 * ROWB;
  static constexpr int SPXB = CO * 2 + 16;
  static constexpr int STAGEB = 2 * (2 * WS) * SPXB;        // two output rows of 2 WS pixels
  static constexpr int TABB = CO * 2 * 4;                   // (mean, rstd) of the fused backward sums
  static constexpr int TABI = CI * 2 * 4;                   // (mean, rstd) of a lazily normalised input
  static constexpr int LDS = RING + 2 * STAGEB + TABB + TABI;
  static constexpr int NCH = RW * CQ;
  static constexpr int LPT = (NCH + kThreads - 1) / kThreads;
  static constexpr int FA = 4 * KN, FB = 5 * KN;            // resident B fragments of a role-A / role-B wave
  static_assert(CQ % 4 == 0 && kWaves % (2 * NB) == 0 && (WS / 16) % NWM == 0, "march(U): wave grid");
  static_assert(2 * WS * (CO / 8) == kThreads, "march(U): one staged piece per thread and output row");
  static_assert(ROWB % 16 == 0 && RING % 16 == 0 && STAGEB % 16 == 0, "march(U): alignment");
};

// tap of class parity p reached through neighbour offset d (0 / 1): kernel coordinate, or -1
__host__ __device__ constexpr int u_tap(int p, int d) { return p == 0 ? (d == 0 ? 1 : -1) : (d == 0 ? 2 : 0); }

// wf: role A, block nb: fragments [(dr, dc) in 00 01 10 11][kk]; role B, block nb: class 00: [00][kk]; class 01: [00][kk], [01][kk];
// class 10: [00][kk], [10][kk].  Fragment = 64 lanes x 8 bf16: W[ci = 32 kk + 8 (lane >> 4) + e][o = 16 nb + (lane & 15)][r][s].
// Master tensor element (ci, o, r, s) at ci * s_i + o * s_o + 3 r + s.
template <int CI, int CO>
__global__ void march_u_pack_kernel(const float* __restrict__ w, bf16_t* __restrict__ wf, long s_i, long s_o) {
  constexpr int KN = CI / 32, NB = CO / 16, FA = 4 * KN, FB = 5 * KN;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= NB * (FA + FB) * 64) return;
  const int lane = idx & 63, f = (idx >> 6) % (FA + FB), nb = (idx >> 6) / (FA + FB);
  int pi, pj, dr, dc, kk;
  if (f < FA) { pi = pj = 1; dr = (f / KN) >> 1; dc = (f / KN) & 1; kk = f % KN; }
  else {
    const int g = (f - FA) / KN;                            // 0: cls00/d00  1: cls01/d00  2: cls01/d01  3: cls10/d00  4: cls10/d10
    kk = (f - FA) % KN;
    pi = g >= 3 ? 1 : 0; pj = (g == 1 || g == 2) ? 1 : 0;
    dr = g == 4 ? 1 : 0; dc = g == 2 ? 1 : 0;
  }
  const int r = u_tap(pi, dr), s3 = u_tap(pj, dc);
  const int o = 16 * nb + (lane & 15), kq = lane >> 4;
  bf16_t v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = (bf16_t)w[(long)(32 * kk + 8 * kq + e) * s_i + (long)o * s_o + 3 * r + s3];
  *reinterpret_cast<uint4*>(wf + (size_t)idx * 8) = *reinterpret_cast<const uint4*>(v);
}

template <int CI, int CO, int WS, bool BSUM, bool LAZY>
__global__ __launch_bounds__(kThreads) void march_u_kernel(const MarchArgs a) {
  typedef UGeom<CI, CO, WS> G;
  constexpr int CQ = G::CQ, KN = G::KN, NB = G::NB, MBW = G::MBW, RW = G::RW, PXB = G::PXB, ROWB = G::ROWB;
  constexpr int SPXB = G::SPXB, STAGEB = G::STAGEB, LPT = G::LPT, NCH = G::NCH, FA = G::FA, FB = G::FB;
  extern __shared__ float4 smem_raw[];
  char* smem = reinterpret_cast<char*>(smem_raw);
  char* stage = smem + G::RING;
  float* bs_tab = reinterpret_cast<float*>(smem + G::RING + 2 * STAGEB);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nb = wave % NB, role = (wave / NB) & 1, wm = wave / (2 * NB);
  const int n16 = lane & 15, kq = lane >> 4;

  int wg = (int)blockIdx.x;
  const int seg = wg % a.nseg; wg /= a.nseg;
  const int strip = wg % a.strips;
  const int n = wg / a.strips;
  const int wi0 = strip * WS, i_first = seg * a.seg_rows;
  const int Hin = a.Hin, Win = a.Win, Hout = a.Ho, Wout = a.Wo;

  // resident B fragments (role A: FA, role B: FB; the array is sized for the larger)
  bf16x8 bfrag[FB];
  {
    const bf16_t* wfw = a.wf + ((size_t)nb * (FA + FB) + (role ? FA : 0)) * 64 * 8;
#pragma unroll
    for (int f = 0; f < FB; ++f)
      bfrag[f] = (role || f < FA) ? *reinterpret_cast<const bf16x8*>(wfw + ((size_t)f * 64 + lane) * 8) : bf16x8{};
  }
  const float bias_n = a.bias != nullptr ? a.bias[16 * nb + n16] : 0.f;
  const unsigned abase = (unsigned)((16 * wm * MBW + n16) * PXB + kq * 16);

  // the new input row of a step as 16-byte chunks
  const auto rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, (int)a.in_bytes, 0x00020000);
  unsigned colB[LPT], lw[LPT];
  int q_of[LPT];
#pragma unroll
  for (int u = 0; u < LPT; ++u) {
    const int j = tid + kThreads * u;
    const int x = j / CQ, q = j - x * CQ;
    const int wi = wi0 + x;
    q_of[u] = q;
    colB[u] = (j < NCH && wi < Win) ? (unsigned)((wi * CI + 8 * q) * 2) : kOOB;
    lw[u] = (unsigned)(min(x, RW - 1) * PXB + q * 16);
  }
  float* in_tab = reinterpret_cast<float*>(smem + G::RING + 2 * STAGEB + G::TABB);
  if constexpr (LAZY) {
    lazy_table_fill(in_tab, a.in_stats, n, CI, a.in_inv_hw, a.in_eps, tid);
    __syncthreads();
  }
  const unsigned sampleB = (unsigned)((size_t)n * Hin * Win * CI * 2);
  const unsigned rowpitchB = (unsigned)(Win * CI * 2);
  u32x4 ld[LPT];
  auto issue_row = [&](int arel) {                                // input row i_first + arel (zeros past the image)
    const int hi = i_first + arel;
#pragma unroll
    for (int u = 0; u < LPT; ++u) {
      const unsigned off = (hi < Hin && colB[u] != kOOB) ? sampleB + (unsigned)hi * rowpitchB + colB[u] : kOOB;
      ld[u] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, (int)off, 0, 0);
    }
  };
  auto write_row = [&](int arel) {
    const int slot = arel % 3;
#pragma unroll
    for (int u = 0; u < LPT; ++u) {
      u32x4 v = ld[u];
      if constexpr (LAZY) v = lazy_norm8(v, in_tab, 8 * q_of[u], a.in_slope, i_first + arel < Hin && colB[u] != kOOB);
      if (tid + kThreads * u < NCH) *reinterpret_cast<u32x4*>(smem + slot * ROWB + lw[u]) = v;
    }
  };

  constexpr int CPR = CO / 8;
  const int ppx = tid / CPR, pcol = tid - ppx * CPR;            // this thread's piece of BOTH output rows of a step
  constexpr bool bsum = BSUM;
  float bs_a1[8], bs_a2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) bs_a1[e] = bs_a2[e] = 0.f;
  if (bsum && tid < CO) {
    const float2 ms = *reinterpret_cast<const float2*>(a.bs_stats + 2 * ((size_t)n * CO + tid));
    bs_tab[2 * tid] = ms.x;
    bs_tab[2 * tid + 1] = rsqrtf(fmaxf(ms.y * a.bs_inv_hw, 0.f) + a.bs_eps);
  }
  issue_row(0);
  write_row(0);
  issue_row(1);
  write_row(1);
  __syncthreads();

  const size_t out_sample = (size_t)n * Hout * Wout * CO;
  auto piece_off = [&](int s_step, int rho) -> size_t {
    return out_sample + ((size_t)(2 * (i_first + s_step) + rho) * Wout + 2 * wi0 + ppx) * CO + 8 * pcol;
  };
  auto store_tile = [&](int s_prev, u32x4 y0, u32x4 y1) {
#pragma unroll
    for (int rho = 0; rho < 2; ++rho) {
      const u32x4 v = *reinterpret_cast<const u32x4*>(stage + (s_prev & 1) * STAGEB + (rho * 2 * WS + ppx) * SPXB + pcol * 16);
      *reinterpret_cast<u32x4*>(a.out + piece_off(s_prev, rho)) = v;
      if (bsum) {
        const u32x4 yv = rho ? y1 : y0;
        const bf16_t* gg = reinterpret_cast<const bf16_t*>(&v);
        const bf16_t* yy = reinterpret_cast<const bf16_t*>(&yv);
        const f32x4 t0 = *reinterpret_cast<const f32x4*>(bs_tab + 16 * pcol), t1 = *reinterpret_cast<const f32x4*>(bs_tab + 16 * pcol + 4);
        const f32x4 t2 = *reinterpret_cast<const f32x4*>(bs_tab + 16 * pcol + 8), t3 = *reinterpret_cast<const f32x4*>(bs_tab + 16 * pcol + 12);
        const float mr[16] = {t0[0], t0[1], t0[2], t0[3], t1[0], t1[1], t1[2], t1[3], t2[0], t2[1], t2[2], t2[3], t3[0], t3[1], t3[2], t3[3]};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float yh = ((float)yy[e] - mr[2 * e]) * mr[2 * e + 1];
          const float gp = (float)gg[e] * (yh > 0.f ? 1.f : a.bs_slope);
          bs_a1[e] += gp; bs_a2[e] += gp * yh;
        }
      }
    }
  };

  // running statistics per class this wave computes (role A: class 11 in slot 0; role B: classes 00, 01, 10)
  float st_n[3] = {0.f, 0.f, 0.f}, st_mean[3] = {0.f, 0.f, 0.f}, st_m2[3] = {0.f, 0.f, 0.f};
  const int nsteps = a.seg_rows;
  for (int s = 0; s < nsteps; ++s) {
    u32x4 y0 = {0u, 0u, 0u, 0u}, y1 = {0u, 0u, 0u, 0u};
    if (bsum && s > 0) {
      y0 = *reinterpret_cast<const u32x4*>(a.bs_y + piece_off(s - 1, 0));
      y1 = *reinterpret_cast<const u32x4*>(a.bs_y + piece_off(s - 1, 1));
    }
    if (s + 1 < nsteps) issue_row(s + 2);
    const char* row0 = smem + (s % 3) * ROWB + abase;
    const char* row1 = smem + ((s + 1) % 3) * ROWB + abase;
    char* sp = stage + (s & 1) * STAGEB + (16 * nb + n16) * 2;
    // finish one class of one pixel block: bias, statistics, staging at output pixel (row pi, 2 b + pj)
    auto finish = [&](const f32x4& acc, int b, int pi, int pj, int slot) {
      float v[4], sum = 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[e] = acc[e] + bias_n; sum += v[e]; }
      if (a.table != nullptr) {
        // Chan's update with equal batches of 4 (see march_s_kernel): batch number MBW s + b of this class
        const float mb = sum * 0.25f;
        float q = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float dlt = v[e] - mb; q += dlt * dlt; }
        const int k = MBW * s + b;
        const float f = __builtin_amdgcn_rcpf((float)(k + 1)), d = mb - st_mean[slot];   // (v_rcp_f32: 1 ulp; an IEEE division is ten instructions, six times per step)
        st_mean[slot] += d * f;
        st_m2[slot] += q + d * d * ((float)(4 * k) * f);
        st_n[slot] = (float)(4 * (k + 1));
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int bpx = 16 * (wm * MBW + b) + 4 * kq + e;
        *reinterpret_cast<bf16_t*>(sp + (pi * 2 * WS + 2 * bpx + pj) * SPXB) = (bf16_t)v[e];
      }
    };
    if (role == 0) {
#pragma unroll
      for (int b = 0; b < MBW; ++b) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int d = 0; d < 4; ++d)
#pragma unroll
          for (int kk = 0; kk < KN; ++kk) {
            const char* rp = (d >> 1) ? row1 : row0;
            const bf16x8 af = *reinterpret_cast<const bf16x8*>(rp + b * 16 * PXB + (d & 1) * PXB + kk * 64);
            acc = p2phd_mfma_16x16x32(af, bfrag[d * KN + kk], acc);
          }
        finish(acc, b, 1, 1, 0);
      }
    } else {
#pragma unroll
      for (int b = 0; b < MBW; ++b) {
        f32x4 c00 = {0.f, 0.f, 0.f, 0.f}, c01 = c00, c10 = c00;
#pragma unroll
        for (int kk = 0; kk < KN; ++kk) {
          const bf16x8 a00 = *reinterpret_cast<const bf16x8*>(row0 + b * 16 * PXB + kk * 64);
          const bf16x8 a01 = *reinterpret_cast<const bf16x8*>(row0 + b * 16 * PXB + PXB + kk * 64);
          const bf16x8 a10 = *reinterpret_cast<const bf16x8*>(row1 + b * 16 * PXB + kk * 64);
          c00 = p2phd_mfma_16x16x32(a00, bfrag[kk], c00);
          c01 = p2phd_mfma_16x16x32(a00, bfrag[KN + kk], c01);
          c10 = p2phd_mfma_16x16x32(a00, bfrag[3 * KN + kk], c10);
          c01 = p2phd_mfma_16x16x32(a01, bfrag[2 * KN + kk], c01);
          c10 = p2phd_mfma_16x16x32(a10, bfrag[4 * KN + kk], c10);
        }
        finish(c00, b, 0, 0, 0);
        finish(c01, b, 0, 1, 1);
        finish(c10, b, 1, 0, 2);
      }
    }
    if (s > 0) store_tile(s - 1, y0, y1);
    if (s + 1 < nsteps) write_row(s + 2);
    __syncthreads();
  }
  {
    u32x4 y0 = {0u, 0u, 0u, 0u}, y1 = {0u, 0u, 0u, 0u};
    if (bsum) {
      y0 = *reinterpret_cast<const u32x4*>(a.bs_y + piece_off(nsteps - 1, 0));
      y1 = *reinterpret_cast<const u32x4*>(a.bs_y + piece_off(nsteps - 1, 1));
    }
    store_tile(nsteps - 1, y0, y1);
  }

  if (a.table != nullptr) {
    // table [N][slots][4 classes][CO][2]; slot = (strip * nseg + seg) * NWM + wm; class index = 2 pi + pj
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      if (role == 0 && k > 0) break;
      float nn = st_n[k], mm = st_mean[k], qq = st_m2[k];
#pragma unroll
      for (int o = 16; o < 64; o <<= 1) {
        const float n2 = __shfl_xor(nn, o), m2 = __shfl_xor(mm, o), q2 = __shfl_xor(qq, o);
        chan_merge(nn, mm, qq, n2, m2, q2);
      }
      if (kq == 0) {
        const int cls = role == 0 ? 3 : k;                       // role B: k = 0, 1, 2 = classes 00, 01, 10
        const int slot = (strip * a.nseg + seg) * G::NWM + wm;
        float* tp = a.table + 2 * ((((size_t)n * a.slots + slot) * 4 + cls) * CO + 16 * nb + n16);
        tp[0] = mm * nn;
        tp[1] = qq;
      }
    }
  }
  if (bsum) {
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);                 // [tid][16] floats: 48 KiB over ring + staging
#pragma unroll
    for (int e = 0; e < 8; ++e) { red[tid * 16 + e] = bs_a1[e]; red[tid * 16 + 8 + e] = bs_a2[e]; }
    __syncthreads();
    if (tid < 2 * CO) {
      const int ch = tid >> 1, which = tid & 1, pc = ch >> 3, e = ch & 7;
      float sum = 0.f;
      for (int p = 0; p < 2 * WS; ++p) sum += red[(p * CPR + pc) * 16 + which * 8 + e];
      a.bs_out[(((size_t)n * (a.strips * a.nseg) + strip * a.nseg + seg) * CO + ch) * 2 + which] = sum;
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------------
// "W" kernel: the weight gradient of both layers --
//   dW[o][i][r][s] = sum_{n, a, b} T96[n, a, b, o] * T48[n, 2 a + r - 1, 2 b + s - 1, i]
// with (T96, T48) = (dy, x) for Conv2d(48, 96, 3, s2, p1) and (x, dy) for ConvTranspose2d(96, 48, 3, s2, p1, op1); both master
// tensors are [96-side][48-side][3][3].  GEMM view: M = 96 (o), N = 9 taps x 48 (i), reduction over the pixels of the small
// plane.  The same march as the S kernel (5-row ring of T48, one T96 row per step, every byte fetched once); the whole 96 x 432
// result lives in the accumulators of the 12 waves (6 row blocks x 2 halves of the 27 column blocks: 56 VGPRs), both operands
// are read pixel-major out of LDS with the transposing ds_read_b64_tr_b16 (the reduction index of the MFMA is the pixel), and a
// workgroup leaves ONE slab at the end of its march; march_w_reduce_kernel adds the slabs in index order (no atomics).
// LAZY = 1 / 2: T48 / T96 is the raw output of an InstanceNorm block, normalised + activated while it is staged (MarchArgs).
// ------------------------------------------------------------------------------------------------------------------------
struct MarchWArgs {
  const bf16_t* t96;      // [N, Hs, Ws, 96]
  const bf16_t* t48;      // [N, 2 Hs, 2 Ws, 48]
  float* slab;            // [workgroups][96][432]
  int N, Hs, Ws, strips, nseg, seg_rows;
  unsigned t96_bytes, t48_bytes;
  const float* stats;     // lazily normalised operand: [N][C][2] of its producer
  float inv_hw, eps, slope;
};
struct WGeom {
  static constexpr int WS = 64, RW = 2 * WS + 1;
  static constexpr int PX48 = 48 * 2 + 16, ROW48 = RW * PX48, RING = 5 * ROW48;
  static constexpr int PX96 = 96 * 2 + 32, ROW96 = WS * PX96;
  static constexpr int TABB = 96 * 2 * 4;
  static constexpr int LDS = RING + 2 * ROW96 + TABB;
  static constexpr int NCH48 = 2 * RW * 6, LPT = (NCH48 + kThreads - 1) / kThreads;
  static constexpr int NBLK = 27, NBW = 14;                 // column blocks (tap x channel third) in all / per wave half
  static_assert(ROW48 % 16 == 0 && ROW96 % 16 == 0 && WS * 12 == kThreads, "march(W): geometry");
};

typedef __attribute__((ext_vector_type(4))) short s16x4;
__device__ __forceinline__ bf16x8 tr_pair(const char* p0, const char* p1) {
  typedef __attribute__((address_space(3))) s16x4* lp;
  bf16x8 v;
  *reinterpret_cast<s16x4*>(&v) = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)p0);
  *(reinterpret_cast<s16x4*>(&v) + 1) = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)p1);
  return v;
}

template <int LAZY>
__global__ __launch_bounds__(kThreads) void march_w_kernel(const MarchWArgs a) {
  typedef WGeom G;
  constexpr int WS = G::WS, RW = G::RW, PX48 = G::PX48, ROW48 = G::ROW48, PX96 = G::PX96, ROW96 = G::ROW96, LPT = G::LPT, NCH48 = G::NCH48;
  extern __shared__ float4 smem_raw[];
  char* smem = reinterpret_cast<char*>(smem_raw);
  char* buf96 = smem + G::RING;
  float* tab = reinterpret_cast<float*>(smem + G::RING + 2 * ROW96);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int mb = wave % 6, nh = wave / 6;
  int wg = (int)blockIdx.x;
  const int seg = wg % a.nseg; wg /= a.nseg;
  const int strip = wg % a.strips;
  const int n = wg / a.strips;
  const int w0 = strip * WS, h_first = seg * a.seg_rows;
  const int Hb = 2 * a.Hs, Wb = 2 * a.Ws;

  if constexpr (LAZY != 0) {
    lazy_table_fill(tab, a.stats, n, LAZY == 1 ? 48 : 96, a.inv_hw, a.eps, tid);
    __syncthreads();
  }
  const auto rs48 = __builtin_amdgcn_make_buffer_rsrc((void*)a.t48, 0, (int)a.t48_bytes, 0x00020000);
  const auto rs96 = __builtin_amdgcn_make_buffer_rsrc((void*)a.t96, 0, (int)a.t96_bytes, 0x00020000);
  // T48: the two new rows of a step as 16-byte chunks (as march_s_kernel); T96: one row of 64 pixels = one chunk per thread
  unsigned colB[LPT], lw[LPT];
  int rr_of[LPT], q_of[LPT];
#pragma unroll
  for (int u = 0; u < LPT; ++u) {
    const int j = tid + kThreads * u;
    const int rr = j / (RW * 6), rem = j - rr * (RW * 6);
    const int x = rem / 6, q = rem - x * 6;
    const int wi = 2 * w0 - 1 + x;
    rr_of[u] = rr; q_of[u] = q;
    colB[u] = (j < NCH48 && wi >= 0 && wi < Wb) ? (unsigned)((wi * 48 + 8 * q) * 2) : kOOB;
    lw[u] = (unsigned)(min(x, RW - 1) * PX48 + q * 16);
  }
  const unsigned sample48 = (unsigned)((size_t)n * Hb * Wb * 48 * 2), pitch48 = (unsigned)(Wb * 48 * 2);
  const int px96 = tid / 12, q96 = tid - px96 * 12;
  const unsigned base96 = (unsigned)((((size_t)n * a.Hs + h_first) * a.Ws + w0 + px96) * 96 * 2 + q96 * 16), pitch96 = (unsigned)(a.Ws * 96 * 2);
  const unsigned lw96 = (unsigned)(px96 * PX96 + q96 * 16);
  u32x4 ld48[LPT], ld96;
  auto issue = [&](int a_first, int srow, bool both) {              // T48 abs rows a_first, a_first + 1; T96 row srow of this segment
#pragma unroll
    for (int u = 0; u < LPT; ++u) {
      const int hi = 2 * h_first - 1 + a_first + rr_of[u];
      const bool ok = hi >= 0 && hi < Hb && colB[u] != kOOB;
      ld48[u] = __builtin_amdgcn_raw_buffer_load_b128(rs48, (int)(ok ? sample48 + (unsigned)hi * pitch48 + colB[u] : kOOB), 0, 0);
    }
    if (both) ld96 = __builtin_amdgcn_raw_buffer_load_b128(rs96, (int)(base96 + (unsigned)srow * pitch96), 0, 0);
  };
  auto write = [&](int a_first, int srow, bool both) {
#pragma unroll
    for (int u = 0; u < LPT; ++u) {
      const int slot = (a_first + rr_of[u] + 5) % 5;
      u32x4 v = ld48[u];
      if constexpr (LAZY == 1) {
        const int hi = 2 * h_first - 1 + a_first + rr_of[u];
        v = lazy_norm8(v, tab, 8 * q_of[u], a.slope, hi >= 0 && hi < Hb && colB[u] != kOOB);
      }
      if (tid + kThreads * u < NCH48) *reinterpret_cast<u32x4*>(smem + slot * ROW48 + lw[u]) = v;
    }
    if (both) {
      u32x4 v = ld96;
      if constexpr (LAZY == 2) v = lazy_norm8(v, tab, 8 * q96, a.slope, true);
      *reinterpret_cast<u32x4*>(buf96 + (srow & 1) * ROW96 + lw96) = v;
    }
  };
  issue(-1, 0, true); write(-1, 0, true);
  issue(1, 0, false); write(1, 0, false);
  __syncthreads();

  // transposing reads: lane 4 q + p of 16-lane group g supplies the address of pixel row (4 g + q [+ 16]) at columns 4 p .. 4 p + 3
  const int g = lane >> 4, q4 = (lane & 15) >> 2, p4 = lane & 3;
  const unsigned aA = (unsigned)((4 * g + q4) * PX96 + (16 * mb + 4 * p4) * 2);
  const unsigned aB = (unsigned)(2 * (4 * g + q4) * PX48 + 4 * p4 * 2);
  f32x4 acc[G::NBW];
#pragma unroll
  for (int i = 0; i < G::NBW; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nsteps = a.seg_rows;
  int a0 = 0;
  for (int s = 0; s < nsteps; ++s) {
    if (s + 1 < nsteps) issue(2 * s + 3, s + 1, true);
    const char* A = buf96 + (s & 1) * ROW96 + aA;
    bf16x8 af[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) af[kk] = tr_pair(A + (32 * kk) * PX96, A + (32 * kk + 16) * PX96);
    int sl[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) { sl[r] = a0 + r; sl[r] = sl[r] >= 5 ? sl[r] - 5 : sl[r]; }
    auto body = [&](auto half_tag) {
      constexpr int NH = decltype(half_tag)::value;
#pragma unroll
      for (int i = 0; i < G::NBW; ++i) {
        constexpr int dummy = 0; (void)dummy;
        const int nbk = G::NBW * NH + i;                             // compile-time after unrolling
        if (nbk < G::NBLK) {
          const int t = nbk / 3, j = nbk - 3 * t, r = t / 3, s3 = t - 3 * r;
          const char* B = smem + sl[r] * ROW48 + aB + s3 * PX48 + 32 * j;
#pragma unroll
          for (int kk = 0; kk < 2; ++kk) {
            const bf16x8 bfr = tr_pair(B + 2 * (32 * kk) * PX48, B + 2 * (32 * kk + 16) * PX48);
            acc[i] = p2phd_mfma_16x16x32(af[kk], bfr, acc[i]);
          }
        }
      }
    };
    if (nh == 0) body(std::integral_constant<int, 0>{}); else body(std::integral_constant<int, 1>{});
    if (s + 1 < nsteps) write(2 * s + 3, s + 1, true);
    __syncthreads();
    a0 += 2;
    a0 = a0 >= 5 ? a0 - 5 : a0;
  }
  // this workgroup's slab: lane (n16 = column within the block, g) holds rows 4 g + e of block i
  float* sb = a.slab + (size_t)blockIdx.x * (96 * 432);
  const int n16 = lane & 15;
#pragma unroll
  for (int i = 0; i < G::NBW; ++i) {
    const int nbk = G::NBW * nh + i;
    if (nbk < G::NBLK) {
      const int col = (nbk / 3) * 48 + (nbk % 3) * 16 + n16;
#pragma unroll
      for (int e = 0; e < 4; ++e) sb[(16 * mb + 4 * g + e) * 432 + col] = acc[i][e];
    }
  }
}

// dw[o][i][r][s] (+)= sum_z slab[z][o][(3 r + s) * 48 + i], z in index order (fixed: reproducible)
__global__ __launch_bounds__(256) void march_w_reduce_kernel(const float* __restrict__ slab, int nslabs, float* __restrict__ dw, int accumulate) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= 96 * 432) return;
  float sum = 0.f;
  for (int z0 = 0; z0 < nslabs; z0 += 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = slab[(size_t)min(z0 + u, nslabs - 1) * (96 * 432) + e];
#pragma unroll
    for (int u = 0; u < 8; ++u) sum += z0 + u < nslabs ? v[u] : 0.f;
  }
  const int o = e / 432, col = e - o * 432, t = col / 48, i = col - t * 48;
  float* d = dw + (size_t)o * 432 + i * 9 + t;
  *d = accumulate ? *d + sum : sum;
}

// segments per strip: enough workgroups for two rounds of the CUs when the plane allows, rows per segment >= 4
int pick_segments(int N, int strips, int Ho) {
  const int cus = g_opt_cus > 0 ? g_opt_cus : device_cus();
  int best = 1;
  for (int nseg = 1; nseg <= Ho; ++nseg) {
    if (Ho % nseg != 0 || Ho / nseg < 4) continue;
    best = nseg;
    if ((long)N * strips * nseg >= cus) break;
  }
  return best;
}

}  // namespace

namespace p2phd {

// kind of marching kernel a layer's launch takes (0 = none).  which: 0 = forward, 1 = input gradient.
//   1 = "S" 48 -> 96 gather: Conv2d(48, 96, 3, s2, p1) forward; input gradient of ConvTranspose2d(96, 48, 3, s2, p1, op1)
//   2 = "U" 96 -> 48 transposed form: that ConvTranspose2d's forward; that Conv2d's input gradient
// march_shape_kind: the shape rule alone (what the packed buffer and the workspace must hold whatever the option says at the
// moment of the pack); march_kind: what a launch takes now.
int march_shape_kind(const p2phd_conv_desc* c, int which) {
  if (c->dtype != P2PHD_BF16 || c->R != 3 || c->S != 3 || c->stride != 2 || c->pad != 1 || c->pad_mode != 0) return 0;
  const bool conv = !c->transposed && c->C == 48 && c->K == 96 && c->H % 2 == 0 && c->W % 128 == 0 && c->H >= 8;
  const bool convt = c->transposed && c->opad == 1 && c->C == 96 && c->K == 48 && c->W % 64 == 0 && c->H >= 4;
  if (which == 0) return conv ? 1 : (convt ? 2 : 0);
  if (which == 1) return convt ? 1 : (conv ? 2 : 0);
  return 0;
}
int march_kind(const p2phd_conv_desc* c, int which) { return g_opt_march == 0 ? 0 : march_shape_kind(c, which); }

namespace {
typedef SGeom<48, 96, 64> GS;
typedef UGeom<96, 48, 64> GU;
struct MarchGeom { int Hin, Win, Ho, Wo, strips, nseg, seg_rows; };
// geometry of the launch: the S kernel marches over OUTPUT rows of its half-resolution plane, the U kernel over INPUT rows
MarchGeom march_geom(const p2phd_conv_desc* c, int kind, int which) {
  MarchGeom g{};
  const bool conv = !c->transposed;                              // the layer is the Conv2d (else the ConvTranspose2d)
  const int Hs = conv ? c->H / 2 : c->H, Ws = conv ? c->W / 2 : c->W;   // the 96-channel (small) plane
  const int Hb = 2 * Hs, Wb = 2 * Ws;                            // the 48-channel (big) plane
  (void)which;
  if (kind == 1) { g.Hin = Hb; g.Win = Wb; g.Ho = Hs; g.Wo = Ws; }
  else { g.Hin = Hs; g.Win = Ws; g.Ho = Hb; g.Wo = Wb; }
  g.strips = Ws / 64;
  g.nseg = pick_segments(c->N, g.strips, Hs);
  g.seg_rows = Hs / g.nseg;
  return g;
}
}  // namespace

size_t march_packed_elems(const p2phd_conv_desc* c, int which) {
  const int kind = march_shape_kind(c, which);
  if (kind == 1) return (size_t)(96 / 16) * GS::KS * 64 * 8;
  if (kind == 2) return (size_t)GU::NB * (GU::FA + GU::FB) * 64 * 8;
  return 0;
}

int march_pack(const p2phd_conv_desc* c, int which, const float* w, void* wf, hipStream_t st) {
  const int kind = march_shape_kind(c, which);
  // both layers keep their weights as [96-side index][48-side index][3][3] or the reverse:
  //   Conv2d(48, 96): [K = 96][C = 48][3][3];  ConvTranspose2d(96, 48): [Cin = 96][Cout = 48][3][3]  -- the same strides
  if (kind == 1) {                                               // out = the 96 side (o), in = the 48 side (ci)
    const int total = (96 / 16) * GS::KS * 64;
    hipLaunchKernelGGL((march_s_pack_kernel<48, 96>), dim3((total + 255) / 256), dim3(256), 0, st, w, (bf16_t*)wf, (long)48 * 9, (long)9);
    return check_launch("march_pack");
  }
  if (kind == 2) {                                               // in = the 96 side (ci), out = the 48 side (o)
    const int total = GU::NB * (GU::FA + GU::FB) * 64;
    hipLaunchKernelGGL((march_u_pack_kernel<96, 48>), dim3((total + 255) / 256), dim3(256), 0, st, w, (bf16_t*)wf, (long)48 * 9, (long)9);
    return check_launch("march_pack");
  }
  set_error("march_pack: layer has no marching kernel");
  return P2PHD_EINVAL;
}

// geometry of the launch `march_run` will make: statistics slots per sample, classes per slot, pixels per (slot, class) and
// per-class plane size (what launch_stats_merge needs); fused-sums partial rows per sample
void march_plan(const p2phd_conv_desc* c, int which, int* slots, int* ncls, int* slot_rows, long* npix_cls, int* bs_tiles) {
  const int kind = march_kind(c, which);
  const MarchGeom g = march_geom(c, kind, which);
  const int nwm = kind == 1 ? GS::NWM : GU::NWM;
  if (slots) *slots = g.strips * g.nseg * nwm;
  if (ncls) *ncls = kind == 1 ? 1 : 4;
  if (slot_rows) *slot_rows = g.seg_rows * (64 / nwm);
  if (npix_cls) *npix_cls = kind == 1 ? (long)g.Ho * g.Wo : (long)g.Hin * g.Win;
  if (bs_tiles) *bs_tiles = g.strips * g.nseg;
}

int march_run(const p2phd_conv_desc* c, int which, const void* in, const void* wf, const float* bias, void* out, float* table,
              const void* bs_y, const float* bs_stats, float* bs_out, float bs_inv_hw, float bs_eps, float bs_slope, hipStream_t st,
              const float* in_stats, float in_slope, float in_eps) {
  const int kind = march_kind(c, which);
  P2PHD_REQUIRE(kind == 1 || kind == 2, "march_run: layer has no marching kernel");
  const MarchGeom g = march_geom(c, kind, which);
  MarchArgs a{};
  a.in = (const bf16_t*)in; a.wf = (const bf16_t*)wf; a.bias = bias; a.out = (bf16_t*)out; a.table = table;
  a.N = c->N; a.Hin = g.Hin; a.Win = g.Win; a.Ho = g.Ho; a.Wo = g.Wo;
  a.strips = g.strips; a.nseg = g.nseg; a.seg_rows = g.seg_rows;
  a.slots = g.strips * g.nseg * (kind == 1 ? GS::NWM : GU::NWM);
  const size_t ib = (size_t)a.N * a.Hin * a.Win * (kind == 1 ? 48 : 96) * 2;
  P2PHD_REQUIRE(ib < 0xFFFFFFF0ull, "march: tensor larger than 4 GiB");
  a.in_bytes = (unsigned)ib;
  a.bs_y = (const bf16_t*)bs_y; a.bs_stats = bs_stats; a.bs_out = bs_out;
  a.bs_inv_hw = bs_inv_hw; a.bs_eps = bs_eps; a.bs_slope = bs_slope;
  a.in_stats = in_stats; a.in_slope = in_slope; a.in_eps = in_eps; a.in_inv_hw = 1.f / ((float)a.Hin * (float)a.Win);
  P2PHD_REQUIRE(in_stats == nullptr || bs_out == nullptr, "march_run: a lazily normalised input goes with forward launches");
  const dim3 grid((unsigned)(a.N * a.strips * a.nseg));
  ++g_launch_count[LC_MARCH];
  auto launch = [&](auto kern, int lds, const char* what) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL(kern, grid, dim3(kThreads), lds, st, a);
    return check_launch(what);
  };
  if (kind == 1) {
    if (in_stats) return launch(march_s_kernel<48, 96, 64, false, true>, GS::LDS, "march_s(lazy)");
    return bs_out ? launch(march_s_kernel<48, 96, 64, true, false>, GS::LDS, "march_s") : launch(march_s_kernel<48, 96, 64, false, false>, GS::LDS, "march_s");
  }
  if (in_stats) return launch(march_u_kernel<96, 48, 64, false, true>, GU::LDS, "march_u(lazy)");
  return bs_out ? launch(march_u_kernel<96, 48, 64, true, false>, GU::LDS, "march_u") : launch(march_u_kernel<96, 48, 64, false, false>, GU::LDS, "march_u");
}

// ---- weight gradient of both layers (march_w_kernel) --------------------------------------------------------------------
bool march_w_ok(const p2phd_conv_desc* c) { return march_kind(c, 0) != 0; }   // same layers, same geometry rule as the forward

size_t march_w_workspace_floats(const p2phd_conv_desc* c) {
  if (march_shape_kind(c, 0) == 0) return 0;
  const MarchGeom g = march_geom(c, 1, 0);
  return (size_t)c->N * g.strips * g.nseg * 96 * 432;
}

// x / dy in the layer's own orientation; x_stats != nullptr: x is the RAW output of its producer's InstanceNorm block
int march_w_run(const p2phd_conv_desc* c, const void* x, const void* dy, float* dw, int accumulate, const float* x_stats, float x_slope,
                float x_eps, float* slabs, hipStream_t st) {
  P2PHD_REQUIRE(march_w_ok(c), "march_w_run: layer has no marching weight-gradient kernel");
  const bool conv = !c->transposed;
  const MarchGeom g = march_geom(c, 1, 0);                        // Ho x Wo = the 96-channel plane, Hin x Win = the 48-channel plane
  MarchWArgs a{};
  a.t96 = (const bf16_t*)(conv ? dy : x); a.t48 = (const bf16_t*)(conv ? x : dy); a.slab = slabs;
  a.N = c->N; a.Hs = g.Ho; a.Ws = g.Wo; a.strips = g.strips; a.nseg = g.nseg; a.seg_rows = g.seg_rows;
  const size_t b96 = (size_t)a.N * a.Hs * a.Ws * 96 * 2, b48 = (size_t)a.N * 4 * a.Hs * a.Ws * 48 * 2;
  P2PHD_REQUIRE(b96 < 0xFFFFFFF0ull && b48 < 0xFFFFFFF0ull, "march(W): tensor larger than 4 GiB");
  a.t96_bytes = (unsigned)b96; a.t48_bytes = (unsigned)b48;
  a.stats = x_stats; a.slope = x_slope; a.eps = x_eps;
  a.inv_hw = conv ? 1.f / ((float)(2 * a.Hs) * (float)(2 * a.Ws)) : 1.f / ((float)a.Hs * (float)a.Ws);   // plane of x
  const int wgs = a.N * a.strips * a.nseg;
  ++g_launch_count[LC_MARCH_W];
  auto launch = [&](auto kern) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, WGeom::LDS);
    hipLaunchKernelGGL(kern, dim3((unsigned)wgs), dim3(kThreads), WGeom::LDS, st, a);
  };
  if (x_stats == nullptr) launch(march_w_kernel<0>);
  else if (conv) launch(march_w_kernel<1>);                       // x is the 48-channel operand
  else launch(march_w_kernel<2>);
  if (int rc = check_launch("march_w")) return rc;
  hipLaunchKernelGGL(march_w_reduce_kernel, dim3((96 * 432 + 255) / 256), dim3(256), 0, st, slabs, wgs, dw, accumulate);
  return check_launch("march_w_reduce");
}

}  // namespace p2phd
