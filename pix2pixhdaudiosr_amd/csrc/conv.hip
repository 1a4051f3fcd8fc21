// Convolution family of the pix2pixHD generator / discriminator for gfx950, as implicit GEMMs on MFMA.
//
// Reference layers covered (models/networks.py): Conv2d 7x7 s1 behind ReflectionPad2d(3) (:190,207),
// Conv2d 3x3 s2 p1 (:194), Conv2d 3x3 s1 behind ReflectionPad2d(1) (:231,246), ConvTranspose2d 3x3 s2 p1
// op1 (:205), Conv2d 4x4 s2/s1 p2 (:342-361), each with forward, input gradient and weight gradient.
//
// One primitive serves all of them: a GATHER CONVOLUTION over NHWC activations
//     out[n, ho*om+oo, wo*om'+oo', k] = sum_{tap t} sum_c in[n, ho*s + dh(t), wo*s + dw(t), c] * Wp[k][t][c]
// with zero or reflect boundary handling folded into the gather index.  Forward convs, the input gradient of
// stride-1 convs and of ConvTranspose2d are single launches; ConvTranspose2d forward and the input gradient
// of stride-2 convs are run as stride^2 sub-pixel classes (each a stride-1 gather with its own tap subset and
// an interleaved output lattice), so no zero-stuffed tensor and no col2im scatter ever exists.
//
// GEMM view: M = output pixels (tiles never straddle samples), N = output channels, K = taps x channels.
// 128 x BN x (128 bytes of K) tiles; 4 wavefronts; A (gathered pixels) and B (packed weights, K-contiguous
// rows) are staged straight global -> LDS with global_load_lds_dwordx4 in 16-byte pieces (coalesced along the
// channel axis = the frequency-major NHWC inner dimension), double-buffered with the next tile's loads issued
// before the MFMAs of the current one; LDS rows are 128 B with a 16-byte-chunk XOR swizzle ((row>>1)&7), applied
// to the per-lane SOURCE address because the LDS side of a direct load is lane-linear, so the ds_read_b128
// fragment reads of v_mfma_f32_32x32x16_bf16 are bank-conflict free.  fp32 mode (parity runs) uses the exact
// v_mfma_f32_32x32x2_f32 on the same tiles.  The epilogue adds bias, accumulates the per-(n,channel) sum and
// sum of squares InstanceNorm needs (wave reduction + one float atomic per wave and channel), applies an
// optional activation, stages the tile in LDS and writes whole 16-byte pieces of NHWC rows.
//
// The weight gradient is a second kernel: M = out channels, N = taps x in channels, reduction over pixels.
// Both operands then have the reduction index as the slow LDS dimension; bf16 fragments are fetched with the
// gfx950 transposing read ds_read_b64_tr_b16, so no transposed copy of the activations is made.
#include "common.h"
#include "convplan.h"

namespace {

using p2phd::GDesc;

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;

__device__ uint4 g_zero_page[4];   // zero-initialised: source of every out-of-image / padding piece

template <typename T> struct Elem;
template <> struct Elem<float> { static constexpr int EPP = 4; };    // elements per 16-byte piece
template <> struct Elem<bf16_t> { static constexpr int EPP = 8; };

__device__ __forceinline__ float to_f(float v) { return v; }
__device__ __forceinline__ float to_f(bf16_t v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f(float v);
template <> __device__ __forceinline__ float from_f<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f<bf16_t>(float v) { return (bf16_t)v; }

__device__ __forceinline__ int reflect_idx(int i, int n) {
  if (i < 0) i = -i;
  if (i >= n) i = 2 * (n - 1) - i;
  return i;
}

__device__ __forceinline__ float apply_act(float v, int act) {
  switch (act) {
    case P2PHD_ACT_LRELU: return v > 0.f ? v : 0.2f * v;
    case P2PHD_ACT_TANH: return tanhf(v);
    case P2PHD_ACT_RELU: return v > 0.f ? v : 0.f;
    default: return v;
  }
}

constexpr int kBM = 128;
constexpr int kRowBytes = 128;   // bytes of K per LDS tile row

// ------------------------------------------------------------------------------------------------------
// gather convolution
// ------------------------------------------------------------------------------------------------------
template <typename T, int BN, int WGM, int WGN, int MR, int NR>
__global__ __launch_bounds__(256) void gconv_kernel(const GDesc d, const T* __restrict__ in, const T* __restrict__ wp,
                                                    const float* __restrict__ bias, const T* __restrict__ addend,
                                                    T* __restrict__ out, float* __restrict__ stats) {
  constexpr int EPP = Elem<T>::EPP;
  constexpr int BK = 8 * EPP;
  constexpr int BM = kBM;
  static_assert(WGM * WGN == 4 && WGM * MR * 32 == BM && WGN * NR * 32 == BN, "tile config");
  constexpr int STAGE = (BM + BN) * kRowBytes;
  constexpr int NB = BN / 32;      // B pieces per thread per step

  // descriptor fields used in loops live in registers (a by-value struct that is captured by reference ends
  // up in scratch memory)
  const int Cp = d.Cp_in, KK = d.KK, Wg = d.Wg, T_taps = d.nth * d.ntw, npix = d.Hg * d.Wg;
  const int Cp_out = d.Cp_out, Kout = d.Kout, act = d.act;

  extern __shared__ float4 smem_raw[];
  char* smem = reinterpret_cast<char*>(smem_raw);
  int* tab = reinterpret_cast<int*>(smem);
  char* stages = smem + ((T_taps * BM * 4 + 15) & ~15);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WGN, wn = wave % WGN;
  const int mtiles = (npix + BM - 1) / BM;
  const int n = blockIdx.x / mtiles;
  const int p_base = (blockIdx.x - n * mtiles) * BM;
  const int n0 = blockIdx.y * BN;

  {  // gather table: input pixel index (or -1) per (tap, tile row)
    const int Hin = d.Hin, Win = d.Win, sh = d.sh, sw = d.sw, ntw = d.ntw, pad_mode = d.pad_mode;
    const int dh0 = d.dh0, dhs = d.dh_step, dw0 = d.dw0, dws = d.dw_step;
    for (int e = tid; e < T_taps * BM; e += 256) {
      const int t = e / BM, r = e - t * BM;
      const int p = p_base + r;
      int off = -1;
      if (p < npix) {
        const int ho = p / Wg, wo = p - ho * Wg;
        const int ta = t / ntw, tb = t - ta * ntw;
        int hi = ho * sh + dh0 + ta * dhs;
        int wi = wo * sw + dw0 + tb * dws;
        if (pad_mode == 1) { hi = reflect_idx(hi, Hin); wi = reflect_idx(wi, Win); }
        if (hi >= 0 && hi < Hin && wi >= 0 && wi < Win) off = (n * Hin + hi) * Win + wi;
      }
      tab[e] = off;
    }
  }
  __syncthreads();

  // Direct global -> LDS staging (global_load_lds_dwordx4): one wave instruction fills 8 consecutive 128-byte
  // tile rows linearly (lane l -> row l>>3, slot l&7).  The bank-conflict swizzle therefore sits on the SOURCE:
  // the lane that owns slot s of row r fetches logical chunk s ^ ((r>>1)&7), and fragment reads undo it.
  const int rbase = tid >> 3;                                 // rows rbase + 32 i
  const int kchunk = (tid & 7) ^ ((rbase >> 1) & 7);          // logical 16-byte chunk of the K slab
  int a_t = (kchunk * EPP) / Cp, a_c = (kchunk * EPP) % Cp;
  int cur_t = -1;
  int rowoff[4] = {-1, -1, -1, -1};
  const T* bsrc = wp + (size_t)(n0 + rbase) * KK + kchunk * EPP;
  const size_t brow = (size_t)32 * KK;
  const T* zero = reinterpret_cast<const T*>(g_zero_page);
  typedef __attribute__((address_space(3))) void* lds_ptr;
  typedef const __attribute__((address_space(1))) void* gbl_ptr;

  auto issue = [&](int stage) {
    char* A = stages + stage * STAGE + (8 * wave) * kRowBytes;
    char* B = A + BM * kRowBytes;
    if (a_t != cur_t) {
      cur_t = a_t;
#pragma unroll
      for (int i = 0; i < 4; ++i) rowoff[i] = a_t < T_taps ? tab[a_t * BM + rbase + 32 * i] : -1;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const T* src = rowoff[i] >= 0 ? in + (size_t)rowoff[i] * Cp + a_c : zero;
      __builtin_amdgcn_global_load_lds((gbl_ptr)src, (lds_ptr)(A + 32 * i * kRowBytes), 16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < NB; ++i)
      __builtin_amdgcn_global_load_lds((gbl_ptr)(bsrc + i * brow), (lds_ptr)(B + 32 * i * kRowBytes), 16, 0, 0);
    bsrc += BK;
    a_c += BK;
    while (a_c >= Cp) { a_c -= Cp; ++a_t; }
  };

  f32x16 acc[MR][NR];
#pragma unroll
  for (int i = 0; i < MR; ++i)
#pragma unroll
    for (int j = 0; j < NR; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int lr = lane & 31, lh = lane >> 5;
  // fragment read offsets: row * 128 + ((jj ^ swz(row)) << 4) with jj = 2 ks + lh; swz(row) = (row >> 1) & 7
  int aoff[MR], asw[MR], boff[NR], bsw[NR];
#pragma unroll
  for (int i = 0; i < MR; ++i) {
    const int row = wm * (MR * 32) + i * 32 + lr;
    aoff[i] = row * kRowBytes; asw[i] = (row >> 1) & 7;
  }
#pragma unroll
  for (int j = 0; j < NR; ++j) {
    const int row = wn * (NR * 32) + j * 32 + lr;
    boff[j] = BM * kRowBytes + row * kRowBytes; bsw[j] = (row >> 1) & 7;
  }

  auto compute = [&](int stage) {
    const char* S = stages + stage * STAGE;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int jj = 2 * ks + lh;
      uint4 af[MR], bfr[NR];
#pragma unroll
      for (int i = 0; i < MR; ++i) af[i] = *reinterpret_cast<const uint4*>(S + aoff[i] + ((jj ^ asw[i]) << 4));
#pragma unroll
      for (int j = 0; j < NR; ++j) bfr[j] = *reinterpret_cast<const uint4*>(S + boff[j] + ((jj ^ bsw[j]) << 4));
#pragma unroll
      for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int j = 0; j < NR; ++j) {
          if constexpr (sizeof(T) == 2) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<bf16x8*>(&af[i]),
                                                                *reinterpret_cast<bf16x8*>(&bfr[j]), acc[i][j], 0, 0, 0);
          } else {
            // exact f32 MFMA; any k permutation is fine as long as A and B share it
            const f32x4 a4 = *reinterpret_cast<f32x4*>(&af[i]);
            const f32x4 b4 = *reinterpret_cast<f32x4*>(&bfr[j]);
#pragma unroll
            for (int e = 0; e < 4; ++e)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[e], b4[e], acc[i][j], 0, 0, 0);
          }
        }
    }
  };

  const int nsteps = KK / BK;
  issue(0);
  __syncthreads();                       // emits s_waitcnt vmcnt(0): the LDS-DMA of stage 0 has landed
  for (int s = 0; s < nsteps; ++s) {
    if (s + 1 < nsteps) issue((s + 1) & 1);
    compute(s & 1);
    __syncthreads();
  }

  // ---- epilogue: bias, InstanceNorm partial sums, activation, LDS-staged coalesced store ----
  constexpr int CROW = BN * (int)sizeof(T) + 16;            // padded C-tile row
  char* ct = stages;
#pragma unroll
  for (int j = 0; j < NR; ++j) {
    const int col = wn * (NR * 32) + j * 32 + lr;
    const int k = n0 + col;
    const float bv = (bias != nullptr && k < Kout) ? bias[k] : 0.f;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < MR; ++i) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = wm * (MR * 32) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
        float v = acc[i][j][e] + bv;
        if (p_base + row < npix) { s1 += v; s2 += v * v; }
        v = apply_act(v, act);
        *reinterpret_cast<T*>(ct + row * CROW + col * (int)sizeof(T)) = from_f<T>(v);
      }
    }
    if (stats != nullptr) {
      s1 += __shfl_xor(s1, 32);
      s2 += __shfl_xor(s2, 32);
      if (lh == 0 && k < Kout) {
        atomicAdd(&stats[2 * ((size_t)n * Cp_out + k)], s1);
        atomicAdd(&stats[2 * ((size_t)n * Cp_out + k) + 1], s2);
      }
    }
  }
  __syncthreads();
  constexpr int CPR = BN / EPP;                              // 16-byte pieces per C-tile row
  const int Hout = d.Hout, Wout = d.Wout, ohm = d.oh_mul, oho = d.oh_off, owm = d.ow_mul, owo = d.ow_off;
  for (int q = tid; q < BM * CPR; q += 256) {
    const int row = q / CPR, pc = q - row * CPR;
    const int p = p_base + row;
    const int k = n0 + pc * EPP;
    if (p >= npix || k >= Cp_out) continue;
    const int ho = p / Wg, wo = p - ho * Wg;
    const size_t opix = ((size_t)n * Hout + (ho * ohm + oho)) * Wout + (wo * owm + owo);
    uint4 v = *reinterpret_cast<const uint4*>(ct + row * CROW + pc * 16);
    if (addend != nullptr) {
      const uint4 a = *reinterpret_cast<const uint4*>(addend + opix * Cp_out + k);
      T* vv = reinterpret_cast<T*>(&v);
      const T* aa = reinterpret_cast<const T*>(&a);
#pragma unroll
      for (int e = 0; e < EPP; ++e) vv[e] = from_f<T>(to_f(vv[e]) + to_f(aa[e]));
    }
    *reinterpret_cast<uint4*>(out + opix * Cp_out + k) = v;
  }
}

// ------------------------------------------------------------------------------------------------------
// weight gradient:  dWp[m][t*Cg + c] (+)= sum_p rows[p][m] * gather[pix(p,t)][c]
//   rows   : [N*Hg*Wg][Cp_r]   the tensor on the pixel grid (dy for Conv2d, x for ConvTranspose2d)
//   gather : [N,Hin,Win,Cp_in] the tensor reached through the taps
// ------------------------------------------------------------------------------------------------------
template <typename T, int TM>
__global__ __launch_bounds__(256) void wgrad_kernel(const GDesc d, const T* __restrict__ rows, const T* __restrict__ gat,
                                                    float* __restrict__ dwp, int Cp_r, int steps_per_split, int use_atomic) {
  constexpr int EPP = Elem<T>::EPP;
  constexpr int BKP = sizeof(T) == 2 ? 64 : 32;             // pixels per K-step
  constexpr int TN = 128;
  static_assert(TM == 128 || TM == 32, "row tile");
  constexpr int WAVES_M = TM == 128 ? 2 : 1, WAVES_N = 4 / WAVES_M;
  constexpr int MI = TM / WAVES_M / 32, NI = TN / WAVES_N / 32;
  // gather-operand tile [BKP][TN]
  constexpr int ROWB = TN * (int)sizeof(T);                 // 256 B (bf16) / 512 B (f32) LDS rows, unpadded
  constexpr int TILE = BKP * ROWB;
  constexpr int CPR = TN / EPP;                             // 16-byte pieces per row: 16 / 32
  constexpr int PPT = BKP * CPR / 256;                      // pieces per thread per tile (4)
  constexpr int RSTEP = 256 / CPR;                          // row distance between a thread's pieces
  constexpr int RPW = 64 / CPR;                             // rows one wave instruction fills: 4 / 2
  // rows-operand tile [BKP][TM]
  constexpr int ROWA = TM * (int)sizeof(T);
  constexpr int TILEA = BKP * ROWA;
  constexpr int CPRA = TM / EPP;
  constexpr int PPTA = BKP * CPRA / 256;                    // 4 (TM 128) or 1 (TM 32)
  constexpr int RSTEPA = 256 / CPRA;
  constexpr int RPWA = 64 / CPRA;
  constexpr int STAGE = TILE + TILEA;

  extern __shared__ float4 smem_raw[];
  char* smem = reinterpret_cast<char*>(smem_raw);
  // loop-resident descriptor fields in registers (see gconv_kernel)
  const int Hg = d.Hg, Wg = d.Wg, Hin = d.Hin, Win = d.Win, Cpi = d.Cp_in, sh = d.sh, sw = d.sw, pad_mode = d.pad_mode, KK = d.KK;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int j0 = blockIdx.x * TN;                           // first kk column
  const int m0 = blockIdx.y * TM;                           // first output row
  const int npix = Hg * Wg;
  const long P = (long)d.N * npix;
  const int total_steps = (int)((P + BKP - 1) / BKP);
  const int s_begin = blockIdx.z * steps_per_split;
  int s_end = s_begin + steps_per_split;
  if (s_end > total_steps) s_end = total_steps;
  if (s_begin >= s_end) return;

  // Direct global -> LDS staging: one wave instruction fills RPW consecutive tile rows linearly.  bf16 tiles are
  // read back with the transposing ds_read_b64_tr_b16, whose 32-lane half touches 4 pixel rows x 64 B: the 16-byte
  // chunk index is XORed with (row & 3) << 2 so the four rows land in different quarters of the 256-byte bank row.
  // As in gconv the swizzle is applied to the per-lane SOURCE column.
  const int rb = tid / CPR;                                 // tile rows rb + RSTEP * i
  const int slot = tid % CPR;
  const int chunk = sizeof(T) == 2 ? (slot ^ ((rb & 3) << 2)) : slot;
  const T* zero = reinterpret_cast<const T*>(g_zero_page);
  const int rbA = tid / CPRA, slotA = tid % CPRA;
  const int chunkA = (sizeof(T) == 2 && TM == 128) ? (slotA ^ ((rbA & 3) << 2)) : slotA;
  const int mcol = m0 + chunkA * EPP;                       // rows-operand column, fixed per thread
  const bool mvalid = mcol < Cp_r;
  const int kk = j0 + chunk * EPP;                          // gather-operand (tap, channel), fixed per thread
  const int T_taps = d.nth * d.ntw;
  const int g_t = kk / Cpi, g_c = kk - g_t * Cpi;
  const bool gvalid = g_t < T_taps;
  const int ta = g_t / d.ntw, tb = g_t - ta * d.ntw;
  const int dh = d.dh0 + ta * d.dh_step, dw = d.dw0 + tb * d.dw_step;
  typedef __attribute__((address_space(3))) void* lds_ptr;
  typedef const __attribute__((address_space(1))) void* gbl_ptr;

  // per-piece pixel coordinates, advanced incrementally (no division in the loop)
  int pn0, pn1, pn2, pn3, ph0, ph1, ph2, ph3, pw0, pw1, pw2, pw3;
  long pbase = (long)s_begin * BKP + rb;
  {
    auto split = [&](long p, int& n_, int& h_, int& w_) {
      const long nn = p / npix;
      const int rem = (int)(p - nn * npix);
      n_ = (int)nn; h_ = rem / Wg; w_ = rem - h_ * Wg;
    };
    split(pbase, pn0, ph0, pw0);
    split(pbase + RSTEP, pn1, ph1, pw1);
    split(pbase + 2 * RSTEP, pn2, ph2, pw2);
    split(pbase + 3 * RSTEP, pn3, ph3, pw3);
  }
  static_assert(PPT == 4, "four pieces per thread");

#define P2PHD_WG_PIECE(I, PN, PH, PW)                                                                        \
  {                                                                                                          \
    const T* s2 = zero;                                                                                      \
    const long pp = pbase + (I) * RSTEP;                                                                     \
    if (pp < P) {                                                                                            \
      if (gvalid) {                                                                                          \
        int hi = PH * sh + dh, wi = PW * sw + dw;                                                            \
        if (pad_mode == 1) { hi = reflect_idx(hi, Hin); wi = reflect_idx(wi, Win); }                         \
        if (hi >= 0 && hi < Hin && wi >= 0 && wi < Win) s2 = gat + ((size_t)(PN * Hin + hi) * Win + wi) * Cpi + g_c; \
      }                                                                                                      \
    }                                                                                                        \
    __builtin_amdgcn_global_load_lds((gbl_ptr)s2, (lds_ptr)(G + (I) * RSTEP * ROWB), 16, 0, 0);             \
    PW += BKP;                                                                                               \
    while (PW >= Wg) { PW -= Wg; ++PH; }                                                                     \
    while (PH >= Hg) { PH -= Hg; ++PN; }                                                                     \
  }

  long pstep = (long)s_begin * BKP;
  auto issue = [&](int stage) {
    char* A = smem + stage * STAGE + (RPWA * wave) * ROWA;
    char* G = smem + stage * STAGE + TILEA + (RPW * wave) * ROWB;
#pragma unroll
    for (int i = 0; i < PPTA; ++i) {
      const long pa = pstep + rbA + RSTEPA * i;
      const T* s1 = (mvalid && pa < P) ? rows + (size_t)pa * Cp_r + mcol : zero;
      __builtin_amdgcn_global_load_lds((gbl_ptr)s1, (lds_ptr)(A + i * RSTEPA * ROWA), 16, 0, 0);
    }
    pstep += BKP;
    P2PHD_WG_PIECE(0, pn0, ph0, pw0)
    P2PHD_WG_PIECE(1, pn1, ph1, pw1)
    P2PHD_WG_PIECE(2, pn2, ph2, pw2)
    P2PHD_WG_PIECE(3, pn3, ph3, pw3)
    pbase += BKP;
  };
#undef P2PHD_WG_PIECE

  f32x16 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  auto compute = [&](int stage) {
    const char* A = smem + stage * STAGE;
    const char* G = A + TILEA;
    if constexpr (sizeof(T) == 2) {
      // transposing LDS read: 16-lane group g reads a 4-pixel x 16-channel block, lane i gets channel i.
      // lane 4q+p of the group supplies row (8h + q), logical 8-byte column unit u = 4*(g&1) + p  (u>>1 = 16-B chunk)
      const int g = lane >> 4, i16 = lane & 15, q = i16 >> 2, pq = i16 & 3, h = g >> 1;
      const int u = 4 * (g & 1) + pq;
      const int swz = q << 2;                                // (row & 3) << 2 with row = 16 sub + 8 h + q (+4)
      const int swzA = TM == 128 ? swz : 0;
      typedef __attribute__((address_space(3))) s16x4* trp;
#pragma unroll
      for (int sub = 0; sub < BKP / 16; ++sub) {
        const int prow = 16 * sub + 8 * h + q;
        bf16x8 af[MI], gf[NI];
#pragma unroll
        for (int i = 0; i < MI; ++i) {
          const int ca = (wm * (MI * 32) + i * 32) / 8 + (u >> 1);  // logical chunk
          const char* pa = A + prow * ROWA + ((ca ^ swzA) << 4) + 8 * (u & 1);
          s16x4* ad = reinterpret_cast<s16x4*>(&af[i]);
          ad[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((trp)(pa));
          ad[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((trp)(pa + 4 * ROWA));
        }
#pragma unroll
        for (int j = 0; j < NI; ++j) {
          const int cg = (wn * (NI * 32) + j * 32) / 8 + (u >> 1);
          const char* pg = G + prow * ROWB + ((cg ^ swz) << 4) + 8 * (u & 1);
          s16x4* gd = reinterpret_cast<s16x4*>(&gf[j]);
          gd[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((trp)(pg));
          gd[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((trp)(pg + 4 * ROWB));
        }
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NI; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], gf[j], acc[i][j], 0, 0, 0);
      }
    } else {
      const int lr = lane & 31, lh = lane >> 5;
#pragma unroll 4
      for (int s2 = 0; s2 < BKP / 2; ++s2) {
        const int prow = 2 * s2 + lh;
        float af[MI], gf[NI];
#pragma unroll
        for (int i = 0; i < MI; ++i) af[i] = *reinterpret_cast<const float*>(A + prow * ROWA + (wm * (MI * 32) + i * 32 + lr) * 4);
#pragma unroll
        for (int j = 0; j < NI; ++j) gf[j] = *reinterpret_cast<const float*>(G + prow * ROWB + (wn * (NI * 32) + j * 32 + lr) * 4);
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NI; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], gf[j], acc[i][j], 0, 0, 0);
      }
    }
  };

  issue(0);
  __syncthreads();
  int st = 0;
  for (int s = s_begin; s < s_end; ++s) {
    if (s + 1 < s_end) issue(st ^ 1);
    compute(st);
    __syncthreads();
    st ^= 1;
  }

  const int lr = lane & 31, lh = lane >> 5;
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int col = j0 + wn * (NI * 32) + j * 32 + lr;
      if (col >= KK) continue;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = m0 + wm * (MI * 32) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
        float* dst = dwp + (size_t)row * KK + col;
        if (use_atomic) atomicAdd(dst, acc[i][j][e]);
        else *dst = acc[i][j][e];
      }
    }
}

// ------------------------------------------------------------------------------------------------------
// weight packing: master f32 tensor (generic strides) -> Wp[rows_pad][KK] of T, zero padded
// and the inverse for gradients (packed f32 -> master layout, overwrite)
// ------------------------------------------------------------------------------------------------------
template <typename T>
__global__ void pack_kernel(GDesc d, p2phd::WMap m, const float* __restrict__ w, T* __restrict__ wp, int rows_pad) {
  const long total = (long)rows_pad * d.KK;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int row = (int)(e / d.KK);
    const int kk = (int)(e - (long)row * d.KK);
    const int t = kk / d.Cp_in, c = kk - t * d.Cp_in;
    float v = 0.f;
    if (row < m.rows && t < d.nth * d.ntw && c < m.inner) {
      const int ta = t / d.ntw, tb = t - ta * d.ntw;
      const int r = d.wr0 + ta * d.wr_step, s = d.ws0 + tb * d.ws_step;
      v = w[(row % m.row_mod) * m.s_row + (row / m.row_mod) * m.s_rowq + (c % m.c_mod) * m.s_inner + (c / m.c_mod) * m.s_innerq + r * m.S + s];
    }
    wp[e] = from_f<T>(v);
  }
}

__global__ void unpack_grad_kernel(GDesc d, p2phd::WMap m, const float* __restrict__ dwp, float* __restrict__ dw) {
  const int T_taps = d.nth * d.ntw;
  const long total = (long)m.rows * T_taps * m.inner;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int c = (int)(e % m.inner);
    const long r2 = e / m.inner;
    const int t = (int)(r2 % T_taps);
    const int row = (int)(r2 / T_taps);
    const int ta = t / d.ntw, tb = t - ta * d.ntw;
    const int r = d.wr0 + ta * d.wr_step, s = d.ws0 + tb * d.ws_step;
    dw[(row % m.row_mod) * m.s_row + (row / m.row_mod) * m.s_rowq + (c % m.c_mod) * m.s_inner + (c / m.c_mod) * m.s_innerq + r * m.S + s] =
        dwp[(size_t)row * d.KK + t * d.Cp_in + c];
  }
}

// reflect-pad adjoint: dx[n,i,j,:] = sum over padded positions that mirror onto (i,j) of dxp (+ addend)
template <typename T>
__global__ void reflect_fold_kernel(const T* __restrict__ dxp, const T* __restrict__ addend, T* __restrict__ dx,
                                    int N, int H, int W, int Cp, int P) {
  constexpr int EPP = Elem<T>::EPP;
  const int cpr = Cp / EPP;
  const long total = (long)N * H * W * cpr;
  const int Hp = H + 2 * P, Wp = W + 2 * P;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int pc = (int)(e % cpr);
    long r = e / cpr;
    const int j = (int)(r % W); r /= W;
    const int i = (int)(r % H);
    const int n = (int)(r / H);
    int hs[2], ws[2], nh = 1, nw = 1;
    hs[0] = i + P; ws[0] = j + P;
    if (i >= 1 && i <= P) hs[nh++] = P - i;
    else if (i >= H - 1 - P && i <= H - 2) hs[nh++] = 2 * (H - 1) - i + P;
    if (j >= 1 && j <= P) ws[nw++] = P - j;
    else if (j >= W - 1 - P && j <= W - 2) ws[nw++] = 2 * (W - 1) - j + P;
    float acc[EPP];
#pragma unroll
    for (int k = 0; k < EPP; ++k) acc[k] = 0.f;
    for (int a = 0; a < nh; ++a)
      for (int b = 0; b < nw; ++b) {
        const uint4 v = *reinterpret_cast<const uint4*>(dxp + (((size_t)n * Hp + hs[a]) * Wp + ws[b]) * Cp + pc * EPP);
        const T* vv = reinterpret_cast<const T*>(&v);
#pragma unroll
        for (int k = 0; k < EPP; ++k) acc[k] += to_f(vv[k]);
      }
    const size_t o = (((size_t)n * H + i) * W + j) * Cp + pc * EPP;
    if (addend != nullptr) {
      const uint4 v = *reinterpret_cast<const uint4*>(addend + o);
      const T* vv = reinterpret_cast<const T*>(&v);
#pragma unroll
      for (int k = 0; k < EPP; ++k) acc[k] += to_f(vv[k]);
    }
    uint4 ov;
    T* oo = reinterpret_cast<T*>(&ov);
#pragma unroll
    for (int k = 0; k < EPP; ++k) oo[k] = from_f<T>(acc[k]);
    *reinterpret_cast<uint4*>(dx + o) = ov;
  }
}

// column sums of a [P][Cp] matrix (bias gradient); db must be zeroed beforehand.
// Block = cpg channel pieces x R pixel rows; partial sums meet in LDS so each block issues one atomic per channel.
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ x, long P, int Cp, int K, float* __restrict__ db, int cpg) {
  constexpr int EPP = Elem<T>::EPP;
  __shared__ float red[256 * 8];
  const int cpr = Cp / EPP;
  const int pl = threadIdx.x % cpg, rl = threadIdx.x / cpg, R = 256 / cpg;
  const int pc = blockIdx.y * cpg + pl;
  float acc[EPP];
#pragma unroll
  for (int k = 0; k < EPP; ++k) acc[k] = 0.f;
  if (pc < cpr) {
    for (long p = (long)blockIdx.x * R + rl; p < P; p += (long)gridDim.x * R) {
      const uint4 v = *reinterpret_cast<const uint4*>(x + (size_t)p * Cp + pc * EPP);
      const T* vv = reinterpret_cast<const T*>(&v);
#pragma unroll
      for (int k = 0; k < EPP; ++k) acc[k] += to_f(vv[k]);
    }
  }
#pragma unroll
  for (int k = 0; k < EPP; ++k) red[threadIdx.x * 8 + k] = acc[k];
  __syncthreads();
  if (rl == 0 && pc < cpr) {
#pragma unroll
    for (int k = 0; k < EPP; ++k) {
      const int c = pc * EPP + k;
      if (c >= K) continue;
      float t = 0.f;
      for (int r = 0; r < R; ++r) t += red[(r * cpg + pl) * 8 + k];
      atomicAdd(&db[c], t);
    }
  }
}

// ------------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------------
template <typename T, int BN, int WGM, int WGN, int MR, int NR>
int launch_gconv_cfg(const GDesc& d, const void* in, const void* wp, const float* bias, const void* addend, void* out,
                     float* stats, hipStream_t st) {
  constexpr int STAGE = (kBM + BN) * kRowBytes;
  constexpr int CT = kBM * (BN * (int)sizeof(T) + 16);
  const int tab = (d.nth * d.ntw * kBM * 4 + 15) & ~15;
  const size_t lds = tab + (size_t)(2 * STAGE > CT ? 2 * STAGE : CT);
  auto kern = gconv_kernel<T, BN, WGM, WGN, MR, NR>;
  if (lds > 64 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const int mtiles = (d.Hg * d.Wg + kBM - 1) / kBM;
  const int ntiles = (d.Cp_out + BN - 1) / BN;
  dim3 grid((unsigned)(mtiles * d.N), (unsigned)ntiles);
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, d, (const T*)in, (const T*)wp, bias, (const T*)addend, (T*)out, stats);
  return p2phd::check_launch("gconv");
}

template <typename T>
int launch_gconv_t(const GDesc& d, const void* in, const void* wp, const float* bias, const void* addend, void* out,
                   float* stats, hipStream_t st) {
  // N tile: the 128-wide tile has the best MFMA density (64x64 per wave) and reads the gathered A operand once;
  // narrower tiles only for layers that would leave most of it empty
  const int k = d.Cp_out;
  const int bn = k > 64 ? 128 : (k > 32 ? 64 : 32);
  if (bn == 128) return launch_gconv_cfg<T, 128, 2, 2, 2, 2>(d, in, wp, bias, addend, out, stats, st);
  if (bn == 64) return launch_gconv_cfg<T, 64, 2, 2, 2, 1>(d, in, wp, bias, addend, out, stats, st);
  return launch_gconv_cfg<T, 32, 4, 1, 1, 1>(d, in, wp, bias, addend, out, stats, st);
}

}  // namespace

namespace p2phd {

int launch_gconv(const GDesc& d, int dtype, const void* in, const void* wp, const float* bias, const void* addend,
                 void* out, float* stats, hipStream_t st) {
  if (d.N == 0 || d.Hg * d.Wg == 0) return P2PHD_OK;
  P2PHD_REQUIRE(d.Cp_in % 8 == 0 && d.Cp_out % 8 == 0, "gconv: channel pitch must be a multiple of 8");
  P2PHD_REQUIRE((long)d.N * d.Hin * d.Win < (1l << 31) && (long)d.N * d.Hout * d.Wout < (1l << 31), "gconv: too many pixels");
  if (dtype == P2PHD_BF16) return launch_gconv_t<bf16_t>(d, in, wp, bias, addend, out, stats, st);
  if (dtype == P2PHD_F32) return launch_gconv_t<float>(d, in, wp, bias, addend, out, stats, st);
  set_error("gconv: unsupported dtype %d", dtype);
  return P2PHD_EUNSUPPORTED;
}

template <typename T, int TM>
void launch_wgrad_cfg(const GDesc& d, const void* rows, const void* gat, float* dwp, int Cp_r, int M_rows_pad, int sps,
                      int splits, int use_atomic, hipStream_t st) {
  constexpr int bkp = sizeof(T) == 2 ? 64 : 32;
  constexpr int lds = 2 * bkp * (128 + TM) * (int)sizeof(T);
  auto kern = wgrad_kernel<T, TM>;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  dim3 grid((unsigned)((d.KK + 127) / 128), (unsigned)(M_rows_pad / TM), (unsigned)splits);
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, d, (const T*)rows, (const T*)gat, dwp, Cp_r, sps, use_atomic);
}

int launch_wgrad(const GDesc& d, int dtype, const void* rows, int Cp_r, int M_rows, int M_rows_pad, const void* gat,
                 float* dwp, hipStream_t st) {
  // dwp: [M_rows_pad][KK] f32, M_rows_pad a multiple of 128; M_rows = rows that carry data
  const long P = (long)d.N * d.Hg * d.Wg;
  const int bkp = dtype == P2PHD_BF16 ? 64 : 32;
  const int total_steps = (int)((P + bkp - 1) / bkp);
  const int tm = M_rows <= 32 ? 32 : 128;                       // narrow row tile for folded 2-channel layers
  const int mrows = tm == 32 ? 32 : M_rows_pad;
  const int tiles = (mrows / tm) * ((d.KK + 127) / 128);
  if (total_steps == 0) {
    (void)hipMemsetAsync(dwp, 0, sizeof(float) * (size_t)M_rows_pad * d.KK, st);
    return P2PHD_OK;
  }
  // split the pixel reduction until the grid covers the chip ~3x, keeping >= 8 steps per split
  int splits = 1;
  while (tiles * splits < 768 && total_steps / (splits * 2) >= 8) splits *= 2;
  const int sps = (total_steps + splits - 1) / splits;
  splits = (total_steps + sps - 1) / sps;
  const int use_atomic = splits > 1;
  if (use_atomic) (void)hipMemsetAsync(dwp, 0, sizeof(float) * (size_t)mrows * d.KK, st);
  if (dtype == P2PHD_BF16) {
    if (tm == 32) launch_wgrad_cfg<bf16_t, 32>(d, rows, gat, dwp, Cp_r, mrows, sps, splits, use_atomic, st);
    else launch_wgrad_cfg<bf16_t, 128>(d, rows, gat, dwp, Cp_r, mrows, sps, splits, use_atomic, st);
  } else if (dtype == P2PHD_F32) {
    if (tm == 32) launch_wgrad_cfg<float, 32>(d, rows, gat, dwp, Cp_r, mrows, sps, splits, use_atomic, st);
    else launch_wgrad_cfg<float, 128>(d, rows, gat, dwp, Cp_r, mrows, sps, splits, use_atomic, st);
  } else {
    set_error("wgrad: unsupported dtype %d", dtype);
    return P2PHD_EUNSUPPORTED;
  }
  return check_launch("wgrad");
}

int launch_pack(const GDesc& d, const WMap& m, int dtype, const float* w, void* wp, int rows_pad, hipStream_t st) {
  const long total = (long)rows_pad * d.KK;
  const int blocks = (int)std::min<long>((total + 255) / 256, 4096);
  if (dtype == P2PHD_BF16)
    hipLaunchKernelGGL(pack_kernel<bf16_t>, dim3(blocks), dim3(256), 0, st, d, m, w, (bf16_t*)wp, rows_pad);
  else
    hipLaunchKernelGGL(pack_kernel<float>, dim3(blocks), dim3(256), 0, st, d, m, w, (float*)wp, rows_pad);
  return check_launch("pack_weights");
}

int launch_unpack_grad(const GDesc& d, const WMap& m, const float* dwp, float* dw, hipStream_t st) {
  const long total = (long)m.rows * d.nth * d.ntw * m.inner;
  const int blocks = (int)std::min<long>((total + 255) / 256, 4096);
  hipLaunchKernelGGL(unpack_grad_kernel, dim3(blocks), dim3(256), 0, st, d, m, dwp, dw);
  return check_launch("unpack_grad");
}

int launch_reflect_fold(int dtype, const void* dxp, const void* addend, void* dx, int N, int H, int W, int Cp, int P,
                        hipStream_t st) {
  const int epp = dtype == P2PHD_BF16 ? 8 : 4;
  const long total = (long)N * H * W * (Cp / epp);
  const int blocks = (int)std::min<long>((total + 255) / 256, 8192);
  if (dtype == P2PHD_BF16)
    hipLaunchKernelGGL(reflect_fold_kernel<bf16_t>, dim3(blocks), dim3(256), 0, st, (const bf16_t*)dxp, (const bf16_t*)addend, (bf16_t*)dx, N, H, W, Cp, P);
  else
    hipLaunchKernelGGL(reflect_fold_kernel<float>, dim3(blocks), dim3(256), 0, st, (const float*)dxp, (const float*)addend, (float*)dx, N, H, W, Cp, P);
  return check_launch("reflect_fold");
}

int launch_colsum(int dtype, const void* x, long P, int Cp, int K, float* db, hipStream_t st) {
  (void)hipMemsetAsync(db, 0, sizeof(float) * (size_t)K, st);
  if (P == 0) return P2PHD_OK;
  const int epp = dtype == P2PHD_BF16 ? 8 : 4;
  const int cpr = Cp / epp;
  int cpg = 1;
  while (cpg * 2 <= cpr && cpg * 2 <= 64) cpg *= 2;
  const int R = 256 / cpg;
  const int ygroups = (cpr + cpg - 1) / cpg;
  const int xblocks = (int)std::max<long>(1, std::min<long>((P + R * 32 - 1) / (R * 32), 512));
  dim3 grid(xblocks, ygroups);
  if (dtype == P2PHD_BF16)
    hipLaunchKernelGGL(colsum_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)x, P, Cp, K, db, cpg);
  else
    hipLaunchKernelGGL(colsum_kernel<float>, grid, dim3(256), 0, st, (const float*)x, P, Cp, K, db, cpg);
  return check_launch("colsum");
}

}  // namespace p2phd
