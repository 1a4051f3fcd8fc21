// Dedicated kernels for the 7x7 reflect-padded convolutions at the two ends of the generator (models/networks.py:190
// Conv2d(2, ngf, 7) behind ReflectionPad2d(3); :207 Conv2d(ngf, 2, 7) + Tanh), bf16, full-resolution planes.
//
// These layers have 39 GFLOP of arithmetic per launch at B = 32 and move 0.47 GB: they are HBM-bound by a factor of
// three even on padded MFMA tiles.  Run through the generic gather-GEMM (W-fold + 128x64 tiles) they were bound by the
// gather path instead -- every output pixel re-gathered 7 x 32 bytes of a materialised 16-channel image -- at 0.8 TB/s.
// Here a workgroup owns an 8 x 128 pixel tile of one sample:
//   c7_in_fwd  (2 -> C_out):  the 15 x 136 input halo is read ONCE into LDS, compacted to its two channels (4 bytes
//     per pixel, reflection applied while filling).  GEMM view per wave: A = weights [16 channels x K], B = patches
//     [K x 16 pixels] with K ordered (dw, dh, c) = 8 x 8 x 2 (taps 7 and beyond carry zero weights), so a lane's B
//     fragment of v_mfma_f32_16x16x32_bf16 is four aligned 4-byte LDS reads (rows dh0..dh0+3 of one pixel) and the A
//     fragments -- the whole weight matrix -- stay in registers for the life of the wave.  The C tile comes out
//     [channel][pixel]; it is staged through LDS and leaves as whole 1 KiB runs of NHWC rows.
//   InstanceNorm statistics: every lane keeps shifted sums of its channels over the tile, merged across lanes with
//     Chan's update once per wave, stored as this wave's slot of the same partial table the generic conv uses.
#include "convplan.h"

namespace {

typedef p2phd_h16 bf16_t;                 // the library's 16-bit storage type: bf16, or fp16 in the -DP2PHD_F16 build (common.h)
typedef __attribute__((ext_vector_type(8))) bf16_t bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

constexpr int TH = 8, TW = 128;            // output tile
constexpr int LROWS = TH + 7;              // input rows held (3 above, 3 below, 1 for the zero-weight tap row 7)
constexpr int LPITCH = 140;                // dwords per LDS input row: >= TW + 8, and 4 * LPITCH % 32 == 16 (bank spread)
// bytes per pixel in the C staging image: NB * 32 of channels + padding that keeps rows 16-byte aligned and spreads the
// 8-byte writes of 16 pixel lanes over the banks (pitch in dwords = 12, 20, 28, 44: 2-way at worst)
constexpr int srow_bytes(int nb) { return nb == 4 ? 176 : nb * 32 + 16; }

// Hand-over of LDS data between the lanes of ONE wave: LDS executes a wave's operations in order, so all that is needed is
// that the compiler keeps the order (memory clobber) and that earlier LDS operations have completed.  Deliberately NOT a
// wavefront-scope fence: that lowers to s_waitcnt vmcnt(0) as well, i.e. it waits for every global load and store in
// flight -- the row prefetch of c7_out_fwd and the output stores of c7_in_fwd ran at one memory round trip per step.
__device__ __forceinline__ void lds_wave_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

__device__ __forceinline__ int reflect_idx(int i, int n) {
  if (i < 0) i = -i;
  if (i >= n) i = 2 * (n - 1) - i;
  return i;
}

// Weights in fragment order: wf[((half * NB + nb) * 4 + s) * 64 + lane] = 8 bf16:
//   A[row = l & 15][k = 32 s + 8 (l >> 4) + j],  k = dw * 16 + dh * 2 + c,  value w[16 (half * NB + nb) + row][c][dh][dw]
// dgrad = 1: the input gradient of Conv2d(K, 2, 7) as a 2 -> K convolution with w_eff[ch][c][dh][dw] = w[c][ch][6-dh][6-dw]
__global__ void c7_pack_in_kernel(const float* __restrict__ w, bf16_t* __restrict__ wf, int K, int nblocks, int dgrad) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= nblocks * 4 * 64) return;
  const int lane = idx & 63, s = (idx >> 6) & 3, blk = idx >> 8;
  const int ch = 16 * blk + (lane & 15), g = lane >> 4;
  bf16_t v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int dw = 2 * s + (g >> 1), dh = 4 * (g & 1) + (j >> 1), c = j & 1;
    const bool ok = ch < K && dh < 7 && dw < 7;
    const int src = dgrad ? ((c * K + ch) * 7 + (6 - dh)) * 7 + (6 - dw) : ((ch * 2 + c) * 7 + dh) * 7 + dw;
    v[j] = (bf16_t)(ok ? w[src] : 0.f);
  }
  *reinterpret_cast<uint4*>(wf + (size_t)idx * 8) = *reinterpret_cast<const uint4*>(v);
}

__device__ __forceinline__ void chan_merge(float& na, float& ma, float& qa, float nb, float mb, float qb) {
  const float n = na + nb;
  const float d = mb - ma, f = nb / n;
  ma += d * f;
  qa += qb + d * d * na * f;
  na = n;
}

// x [N,H,W,8] bf16 (channels 0,1 used) -> y [N,H,W,Cp] bf16, Cp = K (multiple of 16); grid (tiles, N, channel halves)
// ZERO = true: zero padding instead of reflection (the input-gradient use, see c7_out_dgrad)
template <int NB, bool ZERO>
__global__ __launch_bounds__(256) void c7_in_fwd_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ wf,
                                                        const float* __restrict__ bias, bf16_t* __restrict__ y,
                                                        float* __restrict__ table, int H, int W, int Cp, int slots) {
  __shared__ unsigned s_in[LROWS * LPITCH];
  __shared__ __attribute__((aligned(16))) float s_bias[NB * 16];
  constexpr int SROW = srow_bytes(NB);
  __shared__ __attribute__((aligned(16))) unsigned char s_out[4][64 * SROW];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles_w = W / TW;
  const int th = blockIdx.x / tiles_w, tw = blockIdx.x - th * tiles_w;
  const int n = blockIdx.y, half = blockIdx.z;
  const int h0 = th * TH, w0 = tw * TW;
  const int ch0 = half * NB * 16;

  // weights of this channel half: NB x 4 fragments, resident for the whole tile
  bf16x8 wa[NB][4];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int s = 0; s < 4; ++s)
      wa[nb][s] = *reinterpret_cast<const bf16x8*>(wf + ((size_t)((half * NB + nb) * 4 + s) * 64 + lane) * 8);

  // input halo, compacted to (c0, c1) per pixel, reflection folded into the fill
  const unsigned* xin = reinterpret_cast<const unsigned*>(x) + (size_t)n * H * W * 4;      // 4 dwords per pixel
  {
    // all loads first (unconditional, clamped), then the LDS writes: a load inside the loop body with its own wait
    // costs one memory round trip per iteration
    constexpr int NE = LROWS * (TW + 8), NIT = (NE + 255) / 256;
    unsigned v[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int e = min(tid + 256 * it, NE - 1);
      const int r = e / (TW + 8), c = e - r * (TW + 8);
      int hi = h0 + r - 3, wi = w0 + c - 3;
      bool ok = true;
      if constexpr (ZERO) { ok = hi >= 0 && hi < H && wi >= 0 && wi < W; hi = ok ? hi : 0; wi = ok ? wi : 0; }
      else { hi = reflect_idx(hi, H); wi = reflect_idx(wi, W); }
      const unsigned t = xin[((size_t)hi * W + wi) * 4];
      v[it] = ok ? t : 0u;
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int e = tid + 256 * it;
      const int r = e / (TW + 8), c = e - r * (TW + 8);
      if (e < NE) s_in[r * LPITCH + c] = v[it];
    }
  }
  if (tid < NB * 16) s_bias[tid] = bias != nullptr ? bias[ch0 + tid] : 0.f;
  __syncthreads();

  const int p = lane & 15, g = lane >> 4;
  // B fragment of k-step s for the 16-pixel block at column c16: rows (dh0 .. dh0+3) of pixel c16 + p + dw,
  // dw = 2 s + (g >> 1), dh0 = 4 (g & 1); element (row - 3) is the tap offset, the image starts 3 pixels left / above
  const int lane_off = (4 * (g & 1)) * LPITCH + p + (g >> 1);

  // shifted InstanceNorm sums of this lane's channels over the pixels it sees (16 per group)
  float sh[NB][4], s1[NB][4], s2[NB][4];
  bool first = true;

  unsigned char* stage = s_out[wave];
  // 16 groups of 64 pixels per tile (8 rows x 2), four per wave
#pragma unroll 1
  for (int q = 0; q < 4; ++q) {
    const int grp = wave * 4 + q;
    const int row = grp >> 1, col0 = (grp & 1) * 64;
    f32x4 acc[4][NB];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) acc[mb][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
    // 16 (block, k-step) pairs; the fragment of pair i + 1 is read while the three MFMAs of pair i run
    const unsigned* gbase = s_in + row * LPITCH + col0 + lane_off;
    unsigned f[2][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) f[0][i] = gbase[i * LPITCH];
#pragma unroll
    for (int st = 0; st < 16; ++st) {
      const int mb = st >> 2, s = st & 3;
      if (st + 1 < 16) {
        const int mb1 = (st + 1) >> 2, s1n = (st + 1) & 3;
#pragma unroll
        for (int i = 0; i < 4; ++i) f[(st + 1) & 1][i] = gbase[16 * mb1 + i * LPITCH + 2 * s1n];
      }
      const bf16x8 bfr = *reinterpret_cast<const bf16x8*>(f[st & 1]);
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
        acc[mb][nb] = p2phd_mfma_16x16x32(wa[nb][s], bfr, acc[mb][nb]);
    }
    // epilogue of the group: bias, statistics, bf16, stage [pixel][channel]
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const float4 b4 = *reinterpret_cast<const float4*>(s_bias + 16 * nb + 4 * g);
        float v[4] = {acc[mb][nb][0] + b4.x, acc[mb][nb][1] + b4.y, acc[mb][nb][2] + b4.z, acc[mb][nb][3] + b4.w};
        if (first && mb == 0) {
#pragma unroll
          for (int r = 0; r < 4; ++r) { sh[nb][r] = v[r]; s1[nb][r] = 0.f; s2[nb][r] = 0.f; }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) { const float d = v[r] - sh[nb][r]; s1[nb][r] += d; s2[nb][r] += d * d; }
        bf16_t o[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (bf16_t)v[r];
        *reinterpret_cast<uint2*>(stage + (16 * mb + p) * SROW + (16 * nb + 4 * g) * 2) = *reinterpret_cast<const uint2*>(o);
      }
    }
    first = false;
    lds_wave_sync();
    // 64 pixels x (NB * 32) bytes leave as consecutive 16-byte pieces of consecutive NHWC rows
    constexpr int CPP = NB * 2;                                  // 16-byte pieces per pixel (of this channel half)
    bf16_t* orow = y + (((size_t)n * H + h0 + row) * W + w0 + col0) * Cp + ch0;
#pragma unroll
    for (int it = 0; it < CPP; ++it) {
      const int piece = lane + 64 * it;
      const int px = piece / CPP, pc = piece - px * CPP;
      const uint4 v = *reinterpret_cast<const uint4*>(stage + px * SROW + pc * 16);
      *reinterpret_cast<uint4*>(orow + (size_t)px * Cp + pc * 8) = v;
    }
    lds_wave_sync();
  }

  if (table != nullptr) {
    // per lane: 16 values per channel about the shift sh -> (n, mean, M2); Chan merge across the 16 pixel lanes
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        // equal counts at every level of the tree (16, 32, 64, 128 values a side): Chan's update is then symmetric in
        // its two arguments -- mean = (a + b) / 2, M2 = qa + qb + (b - a)^2 n / 2 -- so both partners of a shuffle
        // compute the same numbers with no ordering logic and no division
        float cm = sh[nb][r] + s1[nb][r] * (1.f / 16.f);
        float cq = s2[nb][r] - s1[nb][r] * s1[nb][r] * (1.f / 16.f);
        float half_n = 8.f;
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) {
          const float om = __shfl_xor(cm, o), oq = __shfl_xor(cq, o);
          const float d = om - cm;
          cq = (cq + oq) + d * d * half_n;
          cm = 0.5f * (cm + om);
          half_n *= 2.f;
        }
        const float cn = 256.f;
        if (p == 0) {
          const int c = ch0 + 16 * nb + 4 * g + r;
          float* sp = table + 2 * (((size_t)n * slots + (size_t)blockIdx.x * 4 + wave) * Cp + c);
          sp[0] = cm * cn;                                       // the table holds (sum, squared deviations) per slot
          sp[1] = cq;
        }
      }
  }
}

// Reflection adjoint of the input gradient of Conv2d(C, 2, 7) behind ReflectionPad2d(3): c7_in_fwd<ZERO> has written
// dxpad at the interior positions; the padded frame folds onto the interior pixels within 3 of an edge,
//   dx[i][j] += sum over the other preimages (ph, pw) of (i, j) under the reflection of dxpad[ph][pw],
//   dxpad[p][c] = sum_{dh,dw,n} dy[ph - dh][pw - dw][n] w[n][c][dh][dw]   (dy zero outside the image).
// One thread per (border pixel, 8-channel piece); the weights sit in LDS as [tap][n][C] f32, only the taps that reach into
// the image are visited, dx is updated 16 bytes at a time.
__global__ __launch_bounds__(256) void c7_out_dgrad_fix_kernel(const bf16_t* __restrict__ dy, const float* __restrict__ w,
                                                               bf16_t* __restrict__ dx, int N, int H, int W, int C) {
  extern __shared__ float s_w[];                                  // [49][2][C]
  for (int e = threadIdx.x; e < 49 * 2 * C; e += 256) {
    const int c = e % C, n = (e / C) & 1, t = e / (2 * C);
    s_w[e] = w[(n * C + c) * 49 + t];
  }
  __syncthreads();
  const int cpr = C / 8;
  const int nborder = 6 * W + 6 * (H - 6);
  const long total = (long)N * nborder * cpr;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int pc = (int)(idx % cpr);
  long r = idx / cpr;
  const int e = (int)(r % nborder), n = (int)(r / nborder);
  int i, j;
  if (e < 6 * W) {
    const int k = e / W;
    i = k < 3 ? 1 + k : H - 7 + k;                              // rows 1,2,3, H-4,H-3,H-2
    j = e - k * W;
  } else {
    const int e2 = e - 6 * W, k = e2 % 6, q = e2 / 6;           // the H - 6 rows outside the border rows: 0, 4 .. H-5, H-1
    i = q == 0 ? 0 : (q <= H - 8 ? q + 3 : H - 1);
    j = k < 3 ? 1 + k : W - 7 + k;
  }
  // preimages in padded coordinates: self, and the mirror if (i, j) lies within 3 of an edge (not on it)
  int ph[2], pw[2], nh = 1, nw = 1;
  ph[0] = i + 3; pw[0] = j + 3;
  if (i >= 1 && i <= 3) ph[nh++] = 3 - i; else if (i >= H - 4 && i <= H - 2) ph[nh++] = 2 * H + 1 - i;
  if (j >= 1 && j <= 3) pw[nw++] = 3 - j; else if (j >= W - 4 && j <= W - 2) pw[nw++] = 2 * W + 1 - j;
  float acc[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) acc[k] = 0.f;
  const bf16_t* dyn = dy + (size_t)n * H * W * 8;               // channel pitch 8, channels 0,1
  for (int a = 0; a < nh; ++a)
    for (int b = 0; b < nw; ++b) {
      if (a == 0 && b == 0) continue;
      const int dh_lo = max(0, ph[a] - H + 1), dh_hi = min(6, ph[a]);
      const int dw_lo = max(0, pw[b] - W + 1), dw_hi = min(6, pw[b]);
      for (int dh = dh_lo; dh <= dh_hi; ++dh)
        for (int dw = dw_lo; dw <= dw_hi; ++dw) {
          const unsigned d2 = *reinterpret_cast<const unsigned*>(dyn + ((size_t)(ph[a] - dh) * W + (pw[b] - dw)) * 8);
          float d0, d1;
          p2phd_unpack2(d2, d0, d1);
          const float* w0 = s_w + ((dh * 7 + dw) * 2) * C + pc * 8;
          const float4 a0 = *reinterpret_cast<const float4*>(w0), a1 = *reinterpret_cast<const float4*>(w0 + 4);
          const float4 b0 = *reinterpret_cast<const float4*>(w0 + C), b1 = *reinterpret_cast<const float4*>(w0 + C + 4);
          acc[0] += d0 * a0.x + d1 * b0.x; acc[1] += d0 * a0.y + d1 * b0.y; acc[2] += d0 * a0.z + d1 * b0.z; acc[3] += d0 * a0.w + d1 * b0.w;
          acc[4] += d0 * a1.x + d1 * b1.x; acc[5] += d0 * a1.y + d1 * b1.y; acc[6] += d0 * a1.z + d1 * b1.z; acc[7] += d0 * a1.w + d1 * b1.w;
        }
    }
  bf16_t* o = dx + (((size_t)n * H + i) * W + j) * C + pc * 8;  // channel pitch = C (multiple of 16)
  uint4 v = *reinterpret_cast<const uint4*>(o);
  bf16_t* vv = reinterpret_cast<bf16_t*>(&v);
#pragma unroll
  for (int k = 0; k < 8; ++k) vv[k] = (bf16_t)((float)vv[k] + acc[k]);
  *reinterpret_cast<uint4*>(o) = v;
}

// ------------------------------------------------------------------------------------------------------
// Conv2d(C, 2, 7) behind ReflectionPad2d(3) (+ bias, Tanh): the generator head, forward.
// The horizontal taps ride on the GEMM's reduction axis, the vertical ones on its rows:
//   Z[(dh, n)][qw] = sum_{dw, c} w[n][c][dh][dw] x[r][qw + dw - 3][c]        (A = 16 x 7C weights, resident in registers;
//                                                                            B fragment = one 16-byte piece of pixel qw+dw-3)
//   out[r - dh + 3][qw][n] += Z[(dh, n)][qw]                                 (14 LDS adds per input pixel)
// A workgroup owns a strip of 128 columns and marches down kRowsOut output rows: every input row is read from HBM once
// (plus 6 halo rows per segment), staged in LDS one row ahead, and each wave keeps to its own 32 columns, so output
// rows are finished, biased, squashed and stored by the wave that accumulated them: one barrier per row.
// ------------------------------------------------------------------------------------------------------
constexpr int kRowsOut = 32;
constexpr int kStripW = 128;

// native vector type, not HIP's uint4 struct: a struct copied global -> private -> LDS stays two memcpys in the IR that
// SROA does not break up, i.e. both prefetch sets lived in scratch memory (and scratch traffic queues on vmcnt right
// behind the prefetch it was meant to hide: the kernel ran at one memory round trip per row)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int NLD>
__device__ __forceinline__ void row_fetch(u32x4 (&v)[NLD], const bf16_t* __restrict__ row, const int (&src)[NLD]) {
#pragma unroll
  for (int i = 0; i < NLD; ++i) v[i] = *reinterpret_cast<const u32x4*>(row + src[i]);
}
template <int NLD>
__device__ __forceinline__ void row_commit(unsigned char* __restrict__ dst, const u32x4 (&v)[NLD], const int (&off)[NLD]) {
#pragma unroll
  for (int i = 0; i < NLD; ++i)
    if (off[i] >= 0) *reinterpret_cast<u32x4*>(dst + off[i]) = v[i];
}

// wf[s][lane] = 8 bf16: A[row = l & 15][k = 32 s + 8 (l >> 4) + j], row = dh * 2 + n, k = dw * C + c
__global__ void c7_pack_out_kernel(const float* __restrict__ w, bf16_t* __restrict__ wf, int C, int ksteps) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= ksteps * 64) return;
  const int lane = idx & 63, s = idx >> 6;
  const int row = lane & 15, g = lane >> 4;
  const int dh = row >> 1, n = row & 1;
  bf16_t v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int k = 32 * s + 8 * g + j;
    const int dw = k / C, c = k - dw * C;
    const bool ok = dh < 7 && dw < 7;
    v[j] = (bf16_t)(ok ? w[((n * C + c) * 7 + dh) * 7 + dw] : 0.f);
  }
  *reinterpret_cast<uint4*>(wf + (size_t)idx * 8) = *reinterpret_cast<const uint4*>(v);
}

template <int C>
__global__ __launch_bounds__(256) void c7_out_fwd_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ wf,
                                                         const float* __restrict__ bias, bf16_t* __restrict__ y, int H, int W,
                                                         int act, int abl) {
  constexpr int KS = (7 * C + 31) / 32;                         // k-steps of 32
  constexpr int PPX = C / 8;                                    // 16-byte pieces per pixel
  constexpr int ROWPX = kStripW + 6;
  constexpr int ROWB = ROWPX * C * 2;                           // bytes of one staged input row
  constexpr int NLD = (ROWPX * PPX + 255) / 256;                // pieces per thread and row
  __shared__ __attribute__((aligned(16))) unsigned char s_x[2][ROWB + 16];   // + one zero piece for the K tail
  __shared__ float s_acc[8][kStripW][2];                        // output rows in flight (ring of 8)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int strips = W / kStripW;
  const int st = blockIdx.x % strips, seg = blockIdx.x / strips;
  const int n = blockIdx.y;
  const int w0 = st * kStripW, q0 = seg * kRowsOut;
  const int q1 = min(q0 + kRowsOut, H);

  bf16x8 wa[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) wa[s] = *reinterpret_cast<const bf16x8*>(wf + ((size_t)s * 64 + lane) * 8);
  for (int e = tid; e < 8 * kStripW * 2; e += 256) (&s_acc[0][0][0])[e] = 0.f;
  if (tid < 2) { *reinterpret_cast<uint4*>(s_x[tid] + ROWB) = make_uint4(0u, 0u, 0u, 0u); }

  // this thread's pieces of a staged row: pixel (with the column reflection folded in) and piece index
  int src_off[NLD], dst_off[NLD];
#pragma unroll
  for (int i = 0; i < NLD; ++i) {
    const int e = tid + 256 * i;
    const int px = e / PPX, pc = e - px * PPX;
    const int wi = reflect_idx(w0 + min(px, ROWPX - 1) - 3, W);
    src_off[i] = (wi * C + pc * 8);
    dst_off[i] = e < ROWPX * PPX ? (px * C + pc * 8) * 2 : -1;
  }
  const bf16_t* xn = x + (size_t)n * H * W * C;
  const int p = lane & 15, g = lane >> 4;
  const int r_first = q0 - 3, r_last = q1 - 1 + 3;             // input rows this segment touches
  // rows are fetched TWO steps ahead into two alternating register sets (one step of ~0.3 us does not cover the memory
  // latency; with four workgroups per CU two rows each keep ~100 KB in flight per CU)
  u32x4 preA[NLD], preB[NLD];
  // rows past the segment are clamped to its last row (loaded, never used): every step runs the same code
  auto rowp = [&](int r) { return xn + (size_t)reflect_idx(min(r, r_last), H) * W * C; };
  row_fetch<NLD>(preA, rowp(r_first), src_off);
  __syncthreads();
  row_commit<NLD>(s_x[0], preA, dst_off);
  row_fetch<NLD>(preA, rowp(r_first + 1), src_off);             // committed at the first step
  row_fetch<NLD>(preB, rowp(r_first + 2), src_off);             // committed at the second step
  const float b0 = bias != nullptr ? bias[0] : 0.f, b1 = bias != nullptr ? bias[1] : 0.f;

  // always_inline: called twice per iteration; out of line, every array it captures (the weights!) lives in scratch
  auto compute_row = [&](int r, int buf) __attribute__((always_inline)) {
    const unsigned char* xr = s_x[buf];
    f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    const int qwA = (wave * 2) * 16, qwB = qwA + 16;
    // the two 16-column blocks of the wave as two independent accumulation chains
    if (!(abl & 1))
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      // k = dw * C + c and the staged row is [pixel][C]: the fragment of k-chunk kc for output column qw is the 16
      // bytes at element (qw * C + kc) -- a shift by dw pixels IS a shift by dw * C elements; the K tail reads zeros
      const int kc = 32 * s + 8 * g;
      const int offA = kc < 7 * C ? ((qwA + p) * C + kc) * 2 : ROWB;
      const int offB = kc < 7 * C ? ((qwB + p) * C + kc) * 2 : ROWB;
      const bf16x8 fa = *reinterpret_cast<const bf16x8*>(xr + offA);
      const bf16x8 fb = *reinterpret_cast<const bf16x8*>(xr + offB);
      acc[0] = p2phd_mfma_16x16x32(wa[s], fa, acc[0]);
      acc[1] = p2phd_mfma_16x16x32(wa[s], fb, acc[1]);
    }
    // rows (dh, n) = 4 g + reg: dh = 2 g + (reg >> 1), n = reg & 1; output row q = r - dh + 3
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int dh = 2 * g + (reg >> 1), nn = reg & 1;
        const int q = r - dh + 3;
        if (dh < 7 && q >= q0 && q < q1 && !(abl & 2)) s_acc[q & 7][(cb ? qwB : qwA) + p][nn] += acc[cb][reg];
      }
    // output row q = r - 3 has now received its 7 contributions; each wave finishes its own 32 columns
    const int q = r - 3;
    if (q >= q0 && q < q1) {
      lds_wave_sync();
      if (lane < 32) {
        const int col = wave * 32 + lane;
        float v0 = s_acc[q & 7][col][0] + b0, v1 = s_acc[q & 7][col][1] + b1;
        s_acc[q & 7][col][0] = 0.f; s_acc[q & 7][col][1] = 0.f;
        if (act == P2PHD_ACT_TANH) { v0 = tanhf(v0); v1 = tanhf(v1); }
        else if (act == P2PHD_ACT_RELU) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); }
        else if (act == P2PHD_ACT_LRELU) { v0 = v0 > 0.f ? v0 : 0.2f * v0; v1 = v1 > 0.f ? v1 : 0.2f * v1; }
        const bf16_t o0 = (bf16_t)v0, o1 = (bf16_t)v1;
        const unsigned lo = (unsigned)__builtin_bit_cast(unsigned short, o0) | ((unsigned)__builtin_bit_cast(unsigned short, o1) << 16);
        *reinterpret_cast<uint4*>(y + (((size_t)n * H + q) * W + w0 + col) * 8) = make_uint4(lo, 0u, 0u, 0u);
      }
      lds_wave_sync();
    }
  };

  // two steps per iteration so that the register sets are indexed statically
  for (int r = r_first; r <= r_last; r += 2) {
    __syncthreads();                                            // row r is staged; everybody is done with the other buffer
    row_commit<NLD>(s_x[1], preA, dst_off);
    if (!(abl & 4)) row_fetch<NLD>(preA, rowp(r + 3), src_off);
    compute_row(r, 0);
    __syncthreads();
    row_commit<NLD>(s_x[0], preB, dst_off);
    if (!(abl & 4)) row_fetch<NLD>(preB, rowp(r + 4), src_off);
    if (r + 1 <= r_last) compute_row(r + 1, 1);
  }
}

}  // namespace

namespace p2phd {

bool c7_out_dgrad_ok(const p2phd_conv_desc* c, bool ignore_option) {
  return (ignore_option || !g_opt_c7_generic) && c->dtype == P2PHD_BF16 && !c->transposed && c->K == 2 && c->R == 7 && c->S == 7 && c->stride == 1 &&
         c->pad == 3 && c->pad_mode == 1 && c->C % 16 == 0 && c->C >= 16 && c->C <= 128 && (c->C <= 64 || c->C % 32 == 0) &&
         c->H % TH == 0 && c->W % TW == 0 && c->H >= 8 && c->W >= 8;
}

size_t c7_out_dgrad_packed_elems(const p2phd_conv_desc* c) { return (size_t)(c->C / 16) * 4 * 64 * 8; }

int c7_out_dgrad_pack(const p2phd_conv_desc* c, const float* w, void* wf, hipStream_t st) {
  const int nblocks = c->C / 16;
  const int total = nblocks * 4 * 64;
  hipLaunchKernelGGL(c7_pack_in_kernel, dim3((total + 255) / 256), dim3(256), 0, st, w, (bf16_t*)wf, c->C, nblocks, 1);
  return check_launch("c7_pack(dgrad)");
}

// dx [N,H,W,C] = input gradient of Conv2d(C, 2, 7) behind ReflectionPad2d(3); dy [N,H,W,8], w = master weights [2][C][7][7]
int c7_out_dgrad(const p2phd_conv_desc* c, const void* dy, const void* wf, const float* w_master, void* dx, hipStream_t st) {
  const int halves = c->C > 64 ? 2 : 1;
  const int nb = c->C / 16 / halves;
  dim3 grid((unsigned)((c->H / TH) * (c->W / TW)), (unsigned)c->N, (unsigned)halves);
#define P2PHD_C7D(NBV) hipLaunchKernelGGL((c7_in_fwd_kernel<NBV, true>), grid, dim3(256), 0, st, (const bf16_t*)dy, (const bf16_t*)wf, \
                                          (const float*)nullptr, (bf16_t*)dx, (float*)nullptr, c->H, c->W, cpitch(c->C), 0)
  switch (nb) {
    case 1: P2PHD_C7D(1); break;
    case 2: P2PHD_C7D(2); break;
    case 3: P2PHD_C7D(3); break;
    case 4: P2PHD_C7D(4); break;
    default: set_error("c7_out_dgrad: unsupported channel count %d", c->C); return P2PHD_EUNSUPPORTED;
  }
#undef P2PHD_C7D
  if (int rc = check_launch("c7_out_dgrad")) return rc;
  const long total = (long)c->N * (6 * c->W + 6 * (c->H - 6)) * (c->C / 8);
  hipLaunchKernelGGL(c7_out_dgrad_fix_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 49 * 2 * c->C * sizeof(float), st, (const bf16_t*)dy, w_master,
                     (bf16_t*)dx, c->N, c->H, c->W, c->C);
  return check_launch("c7_out_dgrad_fix");
}

bool c7_out_ok(const p2phd_conv_desc* c, bool ignore_option) {
  return (ignore_option || !g_opt_c7_generic) && c->dtype == P2PHD_BF16 && !c->transposed && c->K == 2 && c->R == 7 && c->S == 7 && c->stride == 1 &&
         c->pad == 3 && c->pad_mode == 1 && (c->C == 32 || c->C == 48 || c->C == 64 || c->C == 96 || c->C == 128) &&
         c->W % kStripW == 0 && c->H >= 8;
}

size_t c7_out_packed_elems(const p2phd_conv_desc* c) { return (size_t)((7 * c->C + 31) / 32) * 64 * 8; }

int c7_out_pack(const p2phd_conv_desc* c, const float* w, void* wf, hipStream_t st) {
  const int ksteps = (7 * c->C + 31) / 32;
  hipLaunchKernelGGL(c7_pack_out_kernel, dim3((ksteps * 64 + 255) / 256), dim3(256), 0, st, w, (bf16_t*)wf, c->C, ksteps);
  return check_launch("c7_pack(out)");
}

// y [N,H,W,8] = act(conv7x7(reflect_pad3(x [N,H,W,C])) + bias), two output channels
int c7_out_fwd(const p2phd_conv_desc* c, const void* x, const void* wf, const float* bias, int act, void* y, hipStream_t st) {
  dim3 grid((unsigned)((c->W / kStripW) * ((c->H + kRowsOut - 1) / kRowsOut)), (unsigned)c->N);
#define P2PHD_C7O(CV) hipLaunchKernelGGL(c7_out_fwd_kernel<CV>, grid, dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)wf, bias, \
                                         (bf16_t*)y, c->H, c->W, act, g_opt_c7_abl)
  switch (c->C) {
    case 32: P2PHD_C7O(32); break;
    case 48: P2PHD_C7O(48); break;
    case 64: P2PHD_C7O(64); break;
    case 96: P2PHD_C7O(96); break;
    case 128: P2PHD_C7O(128); break;
    default: set_error("c7_out_fwd: unsupported channel count %d", c->C); return P2PHD_EUNSUPPORTED;
  }
#undef P2PHD_C7O
  return check_launch("c7_out_fwd");
}

bool c7_in_ok(const p2phd_conv_desc* c, bool ignore_option) {
  return (ignore_option || !g_opt_c7_generic) && c->dtype == P2PHD_BF16 && !c->transposed && c->C == 2 && c->R == 7 && c->S == 7 && c->stride == 1 &&
         c->pad == 3 && c->pad_mode == 1 && c->K % 16 == 0 && c->K >= 16 && c->K <= 128 && (c->K <= 64 || c->K % 32 == 0) &&
         c->H % TH == 0 && c->W % TW == 0 && c->H > 3 && c->W > 3;
}

size_t c7_in_packed_elems(const p2phd_conv_desc* c) { return (size_t)(c->K / 16) * 4 * 64 * 8; }

int c7_in_pack(const p2phd_conv_desc* c, const float* w, void* wf, hipStream_t st) {
  const int nblocks = c->K / 16;
  const int total = nblocks * 4 * 64;
  hipLaunchKernelGGL(c7_pack_in_kernel, dim3((total + 255) / 256), dim3(256), 0, st, w, (bf16_t*)wf, c->K, nblocks, 0);
  return check_launch("c7_pack");
}

int c7_in_slots(const p2phd_conv_desc* c) { return (c->H / TH) * (c->W / TW) * 4; }

// y = conv7x7(reflect_pad3(x)) + bias; `table` (optional): [N][slots][Cp][2] statistics partials, 256 pixels per slot
int c7_in_fwd(const p2phd_conv_desc* c, const void* x, const void* wf, const float* bias, void* y, float* table, hipStream_t st) {
  const int halves = c->K > 64 ? 2 : 1;
  const int nb = c->K / 16 / halves;
  dim3 grid((unsigned)((c->H / TH) * (c->W / TW)), (unsigned)c->N, (unsigned)halves);
  const int slots = c7_in_slots(c);
#define P2PHD_C7(NBV) hipLaunchKernelGGL((c7_in_fwd_kernel<NBV, false>), grid, dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)wf, bias, \
                                         (bf16_t*)y, table, c->H, c->W, cpitch(c->K), slots)
  switch (nb) {
    case 1: P2PHD_C7(1); break;
    case 2: P2PHD_C7(2); break;
    case 3: P2PHD_C7(3); break;
    case 4: P2PHD_C7(4); break;
    default: set_error("c7_in_fwd: unsupported channel count %d", c->K); return P2PHD_EUNSUPPORTED;
  }
#undef P2PHD_C7
  return check_launch("c7_in_fwd");
}

}  // namespace p2phd
