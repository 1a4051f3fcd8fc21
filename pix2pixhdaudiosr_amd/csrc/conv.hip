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
// v_mfma_f32_32x32x2_f32 on the same tiles.  The epilogue adds bias, leaves the per-wave (sum, M2) partials InstanceNorm
// needs in the wave's own slot of a table (merged with Chan's update by a small kernel: no float atomics), applies an
// optional activation, stages the tile in LDS and writes whole 16-byte pieces of NHWC rows.  Round 3: the launch is a 1-D
// grid over tiles whose last, almost empty round is cut along K (split-K tail, launch_gconv_cfg).
//
// The weight gradient is a second kernel: M = out channels, N = taps x in channels, reduction over pixels.
// Both operands then have the reduction index as the slow LDS dimension; bf16 fragments are fetched with the
// gfx950 transposing read ds_read_b64_tr_b16, so no transposed copy of the activations is made.
#include "common.h"
#include "convplan.h"
#include <cmath>
#include <utility>
#include <vector>
#include <type_traits>

namespace {

using p2phd::GDesc;

typedef p2phd_h16 bf16_t;                 // the library's 16-bit storage type: bf16, or fp16 in the -DP2PHD_F16 build (common.h)
typedef __attribute__((ext_vector_type(8))) bf16_t bf16x8;
typedef __attribute__((ext_vector_type(4))) bf16_t bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;


template <typename T> struct Elem;
template <> struct Elem<float> { static constexpr int EPP = 4; };    // elements per 16-byte piece
template <> struct Elem<bf16_t> { static constexpr int EPP = 8; };
// OCP e4m3 operands (block-scaled v_mfma_scale_f32_32x32x64_f8f6f4, unit scales): 16 per 16-byte piece; results leave as bf16
struct fp8_t { unsigned char v; };
template <> struct Elem<fp8_t> { static constexpr int EPP = 16; };
template <typename T> struct OutOf { typedef T type; };
template <> struct OutOf<fp8_t> { typedef bf16_t type; };

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

constexpr int kRowBytes = 128;   // bytes of K per LDS tile row

// ------------------------------------------------------------------------------------------------------
// -DP2PHD_CHECK_WAITS (libp2phd_hip_chk.so, tests/test_gpu_waits.py): a checker for the hand-counted waits of the LDS-DMA
// pipelines.  `s_waitcnt vmcnt(n)` lets the wave's n YOUNGEST vector-memory operations stay in flight; a relaxed wait in
// front of a slab barrier is correct only if none of those n targets a buffer that ANY wave reads behind the barrier.  That is a
// statement about the issue order of every wave, and one wave that issued fewer pieces than its neighbours (round 4: the last
// halo rows belong to waves 0 and 1 only) breaks it without any test noticing on most runs.  In this build every wave logs the
// buffer id of each piece it issues (a 64-bit shift register of 4-bit tags, wave-uniform: scalar registers) and, at every
// relaxed wait, looks at the n youngest tags: a tag inside the `forbid` set raises a device flag (p2phd_wait_check).
// The product build compiles all of it away.
// ------------------------------------------------------------------------------------------------------
#ifdef P2PHD_CHECK_WAITS
__device__ unsigned g_cw_flag[4];     // [0] violations (bit mask of kernel families), [1] relaxed waits checked, [2] first offending (family << 16 | n << 8 | tag), [3] pieces logged
#define P2PHD_CW_DECL unsigned long long cw_log = ~0ull; unsigned cw_pieces = 0
#define P2PHD_CW_ISSUE(tag) do { cw_log = (cw_log << 4) | (unsigned long long)((tag) & 15); ++cw_pieces; } while (0)
#define P2PHD_CW_WAIT(family, n, forbid)                                                                            \
  do {                                                                                                              \
    if ((lane) == 0) {                                                                                              \
      atomicAdd(&g_cw_flag[1], 1u);                                                                                 \
      for (int cw_k = 0; cw_k < (n) && cw_k < 16; ++cw_k) {                                                         \
        const unsigned cw_t = (unsigned)(cw_log >> (4 * cw_k)) & 15u;                                               \
        if (cw_t != 15u && (((forbid) >> cw_t) & 1u)) {                                                             \
          atomicOr(&g_cw_flag[0], 1u << (family));                                                                  \
          atomicCAS(&g_cw_flag[2], 0u, ((unsigned)(family) << 16) | ((unsigned)(n) << 8) | cw_t);                   \
        }                                                                                                           \
      }                                                                                                             \
    }                                                                                                               \
  } while (0)
#define P2PHD_CW_DONE() do { if ((lane) == 0 && cw_pieces) atomicAdd(&g_cw_flag[3], cw_pieces); } while (0)
#else
#define P2PHD_CW_DECL
#define P2PHD_CW_ISSUE(tag) do { } while (0)
#define P2PHD_CW_WAIT(family, n, forbid) do { } while (0)
#define P2PHD_CW_DONE() do { } while (0)
#endif
enum { CW_GCONV = 0, CW_HALO = 1, CW_WGRAD = 2, CW_WGRAD_F32 = 3 };

// ------------------------------------------------------------------------------------------------------
// gather convolution
// ------------------------------------------------------------------------------------------------------
#ifdef P2PHD_PROBE
// experiment builds only (tools/ablate_gconv.sh): per-wave cycle totals of the main loop's wait / barrier / compute parts
constexpr int kProbeSlots = 65536;
__device__ unsigned long long g_probe[kProbeSlots * 8];   // one record per workgroup (wave 0): no atomics in the timed path
#endif

template <typename T, int BM, int BN, int MR, int NR, int NSTAGE, int HALO = 0>
__global__ __launch_bounds__((BM / (MR * 32)) * (BN / (NR * 32)) * 64) void gconv_kernel(const GDesc d, const T* __restrict__ in, const T* __restrict__ wp,
                                                    const float* __restrict__ bias,
                                                    const typename OutOf<T>::type* __restrict__ addend,
                                                    typename OutOf<T>::type* __restrict__ out, float* __restrict__ stats) {
  typedef typename OutOf<T>::type TO;                          // output element (fp8 operands produce bf16)
  constexpr int EPPO = Elem<TO>::EPP;
  constexpr int EPP = Elem<T>::EPP;
  constexpr int BK = 8 * EPP;
  constexpr int WGM = BM / (MR * 32), WGN = BN / (NR * 32);
  constexpr int NT = WGM * WGN * 64;                          // threads: one wave per (MR*32) x (NR*32) sub-tile
  static_assert(NT == BM * 2 || NT == BM, "tile config");
  constexpr int STAGE = (BM + BN) * kRowBytes;
  constexpr int RS = NT / 8;                                  // row distance between a thread's pieces
  constexpr int NB = BN * 8 / NT;                             // B pieces per thread per step
  constexpr int NA = BM * 8 / NT;                             // A pieces per thread per step (4, or 8 with one wave per SIMD)
  static_assert(NB >= 1 && (NA == 4 || NA == 8), "piece distribution");
  constexpr int NLOADS = NA + NB;                             // direct-to-LDS loads per thread per stage

  // descriptor fields used in loops live in registers (a by-value struct that is captured by reference ends
  // up in scratch memory)
  const int Cp = d.Cp_in, KK = d.KK, Wg = d.Wg, T_taps = d.nth * d.ntw, npix = d.Hg * d.Wg;
  const int Cp_out = d.Cp_out, Kout = d.Kout, act = d.act, cls_cp = d.cls_cp, n_extent = d.n_extent, stats_slots = d.stats_slots;

  extern __shared__ float4 smem_raw[];
  char* smem = reinterpret_cast<char*>(smem_raw);
  int* tab = reinterpret_cast<int*>(smem);                    // [T_taps][BM] gathered input pixel (or -1)
  int2* rinfo = reinterpret_cast<int2*>(smem + (HALO ? 0 : ((T_taps * BM * 4 + 15) & ~15)));   // [BM] {sample or -1, ho << 16 | wo}  (HALO: no gather table)
  char* stages = reinterpret_cast<char*>(rinfo + BM);
#ifdef P2PHD_PROBE
  const unsigned long long pr_t0 = __builtin_readcyclecounter();
#endif

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WGN, wn = wave % WGN;
  // M tiling: per sample (tiles never straddle samples; needed for the InstanceNorm sums) or, when no statistics
  // are wanted and the per-sample pixel count does not fill whole tiles, flat over all N * npix pixels
  const bool flat = d.flat_m != 0;
  const int mtiles = (npix + BM - 1) / BM;
  // 1-D launch: workgroups [0, sk_first) are whole tiles (tile = id); from sk_first on, the LAST tiles of the grid -- the
  // ones that would have run as an almost empty extra round of the 256 CUs -- are cut along K into sk_parts workgroups each
  // (workgroup sk_first + part * tail + i works on tile sk_first + i, K slabs [part * sk_steps, ...)): see launch_gconv_cfg.
  int tile_id = (int)blockIdx.x, sk_part = -1, sk_tile = 0;
  if (tile_id >= d.sk_first) {
    const int r = tile_id - d.sk_first;
    sk_part = r / d.sk_tail;
    sk_tile = r - sk_part * d.sk_tail;
    tile_id = d.sk_first + sk_tile;
  }
  const int bx = tile_id % d.grid_m;
  int by = tile_id / d.grid_m;
  if (d.cls_skip != 0) by = (d.n_extent + BN - 1) / BN - 1 - by;    // deepest tiles (class (1,1): 4 taps) first, the 1-tap class last
  const int n = flat ? 0 : bx / mtiles;
  const int p_base = flat ? bx * BM : (bx - n * mtiles) * BM;
  const int p_end = flat ? d.N * npix : npix;                 // rows >= p_end are padding
  const int n0 = by * BN;

  f32x16 acc[MR][NR];
#pragma unroll
  for (int i = 0; i < MR; ++i)
#pragma unroll
    for (int j = 0; j < NR; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  const int lr = lane & 31, lh = lane >> 5;
#ifdef P2PHD_PROBE
  unsigned long long pr_t1_ = 0, pr_wait_ = 0, pr_bar_ = 0, pr_comp_ = 0;
  int nsteps_ = 0;
#ifdef P2PHD_PROBE_FINE
  unsigned long long pf_a = 0, pf_b = 0;
#endif
#endif
  if constexpr (HALO != 0) {
#include "gconv_halo.inc"
  } else {
  {  // row table (the only integer divisions of the kernel: one or two per tile row), then the gather table
    // input pixel index (or -1) per (tap, tile row): each thread walks its row's taps with counters
    const int Hin = d.Hin, Win = d.Win, sh = d.sh, sw = d.sw, ntw = d.ntw, pad_mode = d.pad_mode;
    const int dh0 = d.dh0, dhs = d.dh_step, dw0 = d.dw0, dws = d.dw_step;
    const int r = tid % BM;
    int p = p_base + r, nn = -1, ho = 0, wo = 0;
    if (p < p_end) {
      nn = n;
      if (flat) { nn = p / npix; p -= nn * npix; }
      ho = p / Wg; wo = p - ho * Wg;
    }
    if (tid < BM) rinfo[r] = make_int2(nn, (ho << 16) | wo);
    constexpr int TPR = NT / BM;                              // threads per row (2)
    int ta = 0, tb = tid / BM;
    while (tb >= ntw) { tb -= ntw; ++ta; }
    const bool swap_taps = d.cls_skip != 0 && ((n0 / cls_cp) >> 1) == 1;   // 2 x 2 taps in the K order of a pi = 1 class row
    for (int t = tid / BM; t < T_taps; t += TPR) {
      if (swap_taps) { ta = t & 1; tb = t >> 1; }
      int off = -1;
      if (nn >= 0) {
        int hi = ho * sh + dh0 + ta * dhs;
        int wi = wo * sw + dw0 + tb * dws;
        if (pad_mode == 1) { hi = reflect_idx(hi, Hin); wi = reflect_idx(wi, Win); }
        if (pad_mode == 2) {
          // adjoint of ReflectionPad2d(1) in front of a 3x3 conv, on the EXACT grid: the gathered tensor is dy extended
          // by two virtual rows / columns holding dy[0] + dy[2] and dy[H-3] + dy[H-1] (reflect_expand_kernel); output
          // row 1 reads the first through its tap -1 (where plain zero padding reads dy[2]), row H-2 the second
          // through its tap +1 (instead of dy[H-3]); everything else is the zero-padded transposed conv
          const int Hr = Hin - 2, Wr = Win - 2;
          if (ho == 1 && hi == 2) hi = Hr; else if (ho == Hr - 2 && hi == Hr - 3) hi = Hr + 1; else if (hi >= Hr) hi = -1;
          if (wo == 1 && wi == 2) wi = Wr; else if (wo == Wr - 2 && wi == Wr - 3) wi = Wr + 1; else if (wi >= Wr) wi = -1;
        }
        if (pad_mode == 3) {
          // the same adjoint with dy left as the PLAIN [N, Hin, Win] tensor: the pair-sum rows / columns sit in an extras block
          // behind it (written by the InstanceNorm backward that produced dy, norm.hip): rx_base + n * EX + entry
          const int Hr = Hin, Wr = Win;
          if (ho == 1 && hi == 2) hi = Hr; else if (ho == Hr - 2 && hi == Hr - 3) hi = Hr + 1; else if (hi >= Hr) hi = -1;
          if (wo == 1 && wi == 2) wi = Wr; else if (wo == Wr - 2 && wi == Wr - 3) wi = Wr + 1; else if (wi >= Wr) wi = -1;
          if (hi >= 0 && wi >= 0) {
            if (hi < Hr && wi < Wr) off = (nn * Hr + hi) * Wr + wi;
            else off = d.rx_base + nn * (2 * (Wr + 2) + 2 * Hr) + (hi >= Hr ? (hi - Hr) * (Wr + 2) + wi : 2 * (Wr + 2) + (wi - Wr) * Hr + hi);
          }
        } else if (hi >= 0 && hi < Hin && wi >= 0 && wi < Win) off = (nn * Hin + hi) * Win + wi;
      }
      tab[t * BM + r] = off;
      tb += TPR;
      while (tb >= ntw) { tb -= ntw; ++ta; }
    }
  }
  __syncthreads();
#ifdef P2PHD_PROBE_FINE
  pf_a = __builtin_readcyclecounter();
#endif

  // Direct global -> LDS staging (buffer_load_dwordx4 ... lds): one wave instruction fills 8 consecutive 128-byte
  // tile rows linearly (lane l -> row l>>3, slot l&7).  The bank-conflict swizzle therefore sits on the SOURCE:
  // the lane that owns slot s of row r fetches logical chunk s ^ ((r>>1)&7), and fragment reads undo it.
  // Buffer addressing keeps the per-piece address a 32-bit offset (one VALU add per piece and K step) and gives
  // zero padding for free: an out-of-image piece uses an offset beyond num_records, which loads zeros.
  constexpr unsigned kOOB = 0xFFFFFFF0u;
  constexpr int SZ = (int)sizeof(T);
  const int rbase = tid >> 3;                                 // rows rbase + RS i
  const int kchunk = (tid & 7) ^ ((rbase >> 1) & 7);          // logical 16-byte chunk of the K slab
  const int CpB = Cp * SZ;
  int nsteps_all = KK / BK;
  if (d.cls_skip != 0) {                                       // stop behind the taps of the tile's highest class
    const int cls_hi = min(3, (n0 + BN - 1) / cls_cp);
    const int ktaps = ((cls_hi >> 1) + 1) * ((cls_hi & 1) + 1);
    nsteps_all = min(nsteps_all, (ktaps * Cp + BK - 1) / BK);
  }
  const int s_begin = sk_part < 0 ? 0 : sk_part * d.sk_steps;     // first K slab of this workgroup
  const int nsteps = sk_part < 0 ? nsteps_all : min(d.sk_steps, nsteps_all - s_begin);
  int a_t, a_cB;
  {
    const long kb = (long)kchunk * EPP * SZ + (long)s_begin * kRowBytes;   // byte position of this thread's chunk in the K row
    a_t = (int)(kb / CpB);
    a_cB = (int)(kb - (long)a_t * CpB);
  }
  int cur_t = -1;
  unsigned aoffb[NA], va[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i) aoffb[i] = kOOB;
  unsigned boffb[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i) boffb[i] = (unsigned)(((size_t)(n0 + rbase + RS * i) * KK + kchunk * EPP) * SZ);
  const unsigned tab_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) int*)tab;
  typedef __attribute__((address_space(3))) void* lds_ptr;
  const auto rsA = __builtin_amdgcn_make_buffer_rsrc((void*)in, 0, (int)d.in_bytes, 0x00020000);
  const auto rsB = __builtin_amdgcn_make_buffer_rsrc((void*)wp, 0, (int)d.w_bytes, 0x00020000);

  // byte offsets of this thread's four A pieces for the next K slab
  auto prepare = [&]() {
    if (a_t != cur_t) {
      cur_t = a_t;
      if (a_t < T_taps) {
        // asm: a C++ LDS read here would make hipcc drain the LDS-DMA queue (see compute)
        int ro[NA];
        const unsigned ta = tab_base + (unsigned)(a_t * BM + rbase) * 4u;
#pragma unroll
        for (int i = 0; i < NA; ++i) asm volatile("ds_read_b32 %0, %1" : "=v"(ro[i]) : "v"(ta + (unsigned)(RS * i * 4)));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < NA; ++i) aoffb[i] = ro[i] >= 0 ? (unsigned)ro[i] * (unsigned)CpB : kOOB;
      } else {
#pragma unroll
        for (int i = 0; i < NA; ++i) aoffb[i] = kOOB;
      }
    }
#pragma unroll
    for (int i = 0; i < NA; ++i) va[i] = aoffb[i] == kOOB ? kOOB : aoffb[i] + (unsigned)a_cB;
    a_cB += kRowBytes;
    while (a_cB >= CpB) { a_cB -= CpB; ++a_t; }
  };
  // piece j of a tile: 0..NA-1 = A rows rbase + RS j, NA.. = B rows; tile = K-slab index (scalar offset of B)
  P2PHD_CW_DECL;
  auto issue_piece = [&](int slot, int tile, int j) {
    char* A = stages + slot * STAGE + (8 * wave) * kRowBytes;
    P2PHD_CW_ISSUE(slot);                                      // (check build: tag = the ring slot the piece fills)
    if (j < NA) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr)(A + RS * j * kRowBytes), 16, (int)va[j], 0, 0, 0);
    } else {
      char* B = A + BM * kRowBytes;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_ptr)(B + RS * (j - NA) * kRowBytes), 16, (int)boffb[j - NA],
                                               tile * kRowBytes, 0, 0);
    }
  };

  // Fragment reads are inline-asm ds_read_b128: hipcc cannot prove a C++ LDS read independent of the LDS-DMA still in
  // flight and would drain it (s_waitcnt vmcnt(0)) in front of every K step; the waits here are counted by hand.
  // byte offset inside a stage of the fragment of k-step ks: row * 128 + (((2 ks + lh) ^ ((row >> 1) & 7)) << 4)
  //   = (offset of k-step 0) ^ (ks << 5): one address register per fragment row, the k-step is an XOR at the use
  unsigned fa[MR], fb[NR];
  {
#pragma unroll
    for (int i = 0; i < MR; ++i) {
      const int row = wm * (MR * 32) + i * 32 + lr;
      fa[i] = row * kRowBytes + ((lh ^ ((row >> 1) & 7)) << 4);
    }
#pragma unroll
    for (int j = 0; j < NR; ++j) {
      const int row = wn * (NR * 32) + j * 32 + lr;
      fb[j] = BM * kRowBytes + row * kRowBytes + ((lh ^ ((row >> 1) & 7)) << 4);
    }
  }
  const unsigned frag_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)stages;

  // ---- main loop -------------------------------------------------------------------------------------------
  // NSTAGE-slot LDS ring; tile t lives in slot t % NSTAGE.  ONE workgroup barrier per K slab, placed in front of the
  // slab's LAST MFMA cluster (k-step 3), after the wave has (a) every fragment of the slab in registers
  // (lgkmcnt(0): its LDS reads of the slot are complete) and (b) its own LDS-DMA pieces of the NEXT tile landed
  // (counted vmcnt).  Past that barrier
  //   * the next tile is readable: its first fragments are fetched while the last cluster of this slab runs, so the
  //     matrix pipe never waits for a barrier + LDS round trip at a slab boundary;
  //   * this slab's slot is free: tile s + NSTAGE is issued into it at once (half now, half one k-step later), giving
  //     the DMA more than a full slab of MFMA work to land, even on the 2-slot ring of the 256-wide tiles.
  // The later tiles stay in flight ACROSS the barrier (raw s_barrier; __syncthreads() would drain them).
  // fragment buffers: two (ping-pong over the k-steps), or one per k-step for the e4m3 operands, whose block-scaled MFMA
  // consumes the fragments of TWO k-steps at once (see mfma_one)
  constexpr int NFB = sizeof(T) == 1 ? 4 : 2;
  uint4 af[NFB][MR], bfr[NFB][NR];
  auto read_frags = [&](unsigned so_, int ks, int buf) {
    const unsigned so = so_ + frag_base;
#pragma unroll
    for (int i = 0; i < MR; ++i) asm volatile("ds_read_b128 %0, %1" : "=v"(af[buf][i]) : "v"((fa[i] ^ (unsigned)(ks << 5)) + so));
#pragma unroll
    for (int j = 0; j < NR; ++j) asm volatile("ds_read_b128 %0, %1" : "=v"(bfr[buf][j]) : "v"((fb[j] ^ (unsigned)(ks << 5)) + so));
  };
  // One MFMA cluster (MR x NR tiles, one k-step); `h0` / `h1` are issued in the shadow of its first / second MFMA
  // (fragment reads, LDS-DMA issue), so the matrix pipe already has work when the wave turns to them.
  auto mfma_one = [&](int buf, int i, int j) {
    if constexpr (sizeof(T) == 1) {
      // (block-scaled form: see mfma8 below)
      (void)buf; (void)i; (void)j;
    } else if constexpr (sizeof(T) == 2) {
      acc[i][j] = p2phd_mfma_32x32x16(*reinterpret_cast<bf16x8*>(&af[buf][i]),
                                                          *reinterpret_cast<bf16x8*>(&bfr[buf][j]), acc[i][j]);
    } else {
      // exact f32 MFMA; any k permutation is fine as long as A and B share it
      const f32x4 a4 = *reinterpret_cast<f32x4*>(&af[buf][i]);
      const f32x4 b4 = *reinterpret_cast<f32x4*>(&bfr[buf][j]);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[e], b4[e], acc[i][j], 0, 0, 0);
    }
  };
  // Block-scaled MFMA of the e4m3 operands (round 3): v_mfma_scale_f32_32x32x64_f8f6f4 runs at TWICE the bf16 rate (the
  // non-scaled 32x32x16_fp8_fp8 runs AT the bf16 rate).  It takes 32 bytes of K per lane: the 16-byte fragments of two
  // consecutive k-steps side by side (any K permutation is fine as long as A and B share it).  Scales: e8m0 = 127 (1.0)
  // for every 32-element block -- the layer's scale is applied once in the epilogue, as before, so the numbers are those
  // of the non-scaled form.  A pair of k-steps (2q, 2q+1) is complete at the odd k-step, where its MR x NR MFMAs go out
  // (H1 = all of them; spreading them over both k-steps of a pair measured slower).  Position p of k-step ks:
  constexpr int NT8 = MR * NR, H1 = NT8;
  auto mfma8 = [&](int ks, int p) {
    if constexpr (sizeof(T) == 1) {
      const int idx = (ks & 1) ? p : H1 + p;
      if (p < H1 && idx < NT8) {
        const int lo = (ks & 1) ? ks - 1 : ((ks + 2) & 3), hi = lo + 1;
        const int i = idx / NR, j = idx % NR;
        typedef __attribute__((ext_vector_type(8))) int i32x8;
        const uint4 a0 = af[lo][i], a1 = af[hi][i], b0 = bfr[lo][j], b1 = bfr[hi][j];
        const i32x8 av = {(int)a0.x, (int)a0.y, (int)a0.z, (int)a0.w, (int)a1.x, (int)a1.y, (int)a1.z, (int)a1.w};
        const i32x8 bv = {(int)b0.x, (int)b0.y, (int)b0.z, (int)b0.w, (int)b1.x, (int)b1.y, (int)b1.z, (int)b1.w};
        acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, acc[i][j], 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
      }
    }
  };
  constexpr bool kScaled = sizeof(T) == 1;
  if constexpr (kScaled) {                                     // (the first slab's k-step 0 multiplies zeros)
#pragma unroll
    for (int i = 0; i < MR; ++i) af[2][i] = af[3][i] = make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int j = 0; j < NR; ++j) bfr[2][j] = bfr[3][j] = make_uint4(0, 0, 0, 0);
  }
  // the MFMAs of a cluster after its first `skip`
  auto mfma_rest = [&](int buf, int skip) {
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
      for (int j = 0; j < NR; ++j)
        if (i * NR + j >= skip) mfma_one(buf, i, j);
  };

#pragma unroll
  for (int t = 0; t < NSTAGE; ++t) {
    if (t < nsteps) {
      prepare();
#pragma unroll
      for (int j = 0; j < NLOADS; ++j) issue_piece(t, s_begin + t, j);
    }
  }
#ifdef P2PHD_PROBE_FINE
  pf_b = __builtin_readcyclecounter();
#endif
  if (nsteps >= NSTAGE) {
    P2PHD_CW_WAIT(CW_GCONV, (NSTAGE - 1) * NLOADS, 1u << 0);   // slot 0 is read behind the barrier
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"((NSTAGE - 1) * NLOADS) : "memory");
  } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  read_frags(0u, 0, 0);

  int cur = 0;
  bool pend = false;                              // second half of a tile's pieces still to be issued (at k-step 0)
  int pend_slot = 0, pend_tile = 0;
#ifdef P2PHD_PROBE
  unsigned long long pr_wait = 0, pr_bar = 0, pr_comp = 0;
  const unsigned long long pr_t1 = __builtin_readcyclecounter();
#endif
  for (int s = 0; s < nsteps; ++s) {
    const unsigned so = (unsigned)(cur * STAGE);
    const int nslot = cur == NSTAGE - 1 ? 0 : cur + 1;
#ifdef P2PHD_PROBE
    const unsigned long long pt0 = __builtin_readcyclecounter();
#endif
    // Each k-step: its fragments were fetched behind the previous cluster and have had that cluster's time to land.
    // The first MFMA goes out at once; the next fragment reads and the LDS-DMA issue follow in its shadow (written out
    // inline: a closure that captures the unrolled `ks` turns the fragment-address arrays into scratch).
    const bool has_next = s + 1 < nsteps;
    const bool issue_new = s + NSTAGE < nsteps;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int buf = sizeof(T) == 1 ? ks : (ks & 1);
      const int nbuf = sizeof(T) == 1 ? ((ks + 1) & 3) : (buf ^ 1);   // where the next k-step's fragments go
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (ks == 3 && has_next) {
        // slab boundary: every LDS read of this slot is complete; own pieces of the next tile must have landed
#ifdef P2PHD_PROBE
        const unsigned long long pt1 = __builtin_readcyclecounter();
#endif
        if (NSTAGE > 2 && s + NSTAGE - 1 < nsteps) {
          P2PHD_CW_WAIT(CW_GCONV, (NSTAGE - 2) * NLOADS, 1u << nslot);   // the next slab's slot is read behind the barrier
          asm volatile("s_waitcnt vmcnt(%0)" :: "n"((NSTAGE - 2) * NLOADS) : "memory");
        } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef P2PHD_PROBE
        const unsigned long long pt2 = __builtin_readcyclecounter();
        pr_wait += pt2 - pt1;
#endif
        __builtin_amdgcn_s_barrier();
#ifdef P2PHD_PROBE
        pr_bar += __builtin_readcyclecounter() - pt2;
#endif
      }
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (kScaled) mfma8(ks, 0); else mfma_one(buf, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (ks < 3) read_frags(so, ks + 1, nbuf);
      else if (has_next) read_frags((unsigned)(nslot * STAGE), 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (kScaled) mfma8(ks, 1); else if constexpr (MR * NR > 1) mfma_one(buf, 1 / NR, 1 % NR);
      __builtin_amdgcn_sched_barrier(0);
      if (ks == 0 && pend) {
#pragma unroll
        for (int j = 1; j < NLOADS; j += 2) issue_piece(pend_slot, pend_tile, j);
        pend = false;
      }
      if (ks == 3 && issue_new) {
        prepare();
#pragma unroll
        for (int j = 0; j < NLOADS; j += 2) issue_piece(cur, s_begin + s + NSTAGE, j);
        pend = true; pend_slot = cur; pend_tile = s_begin + s + NSTAGE;
      }
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (kScaled) {
#pragma unroll
        for (int p = 2; p < H1; ++p) mfma8(ks, p);
      } else {
        mfma_rest(buf, MR * NR > 1 ? 2 : 1);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#ifdef P2PHD_PROBE
    pr_comp += __builtin_readcyclecounter() - pt0;
#endif
    cur = nslot;
  }
  if constexpr (kScaled) {                                     // second half of the last slab's pair (2,3)
#pragma unroll
    for (int p = 0; p < H1; ++p) mfma8(0, p);
  }
#ifdef P2PHD_PROBE
  pr_t1_ = pr_t1; pr_wait_ = pr_wait; pr_bar_ = pr_bar; pr_comp_ = pr_comp; nsteps_ = nsteps;
#endif
  P2PHD_CW_DONE();
  }  // !HALO
#ifdef P2PHD_PROBE
  const unsigned long long pr_t2 = __builtin_readcyclecounter();
#endif
  __syncthreads();

  if (sk_part >= 0) {
    // Split tile: every part stores its raw accumulators (float4 pieces, lane-interleaved: coalesced), takes a ticket of
    // the tile, and the part that arrives LAST adds all parts in index order -- a fixed summation order whatever the
    // timing -- and carries on into the normal epilogue.  sc1 stores / loads: the parts run on different XCDs.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // partials travel as 16-byte pieces, lane-interleaved (coalesced), through buffer instructions with the sc0 sc1 cache
    // policy on BOTH sides: write-through stores, and loads that are served from memory, not from this XCD's L2 -- an
    // agent-scope acquire followed by plain loads read stale partials of the previous launch here (measured: wrong sums in
    // the tail tiles), the per-XCD L2s are not coherent with each other.  Compiler-visible builtins: the waits are its.
    constexpr int NQ = MR * NR * 4;                            // 16-byte pieces per thread
    constexpr int kSc = 0x11;                                  // aux: bit 0 = sc0, bit 4 = sc1
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
    const size_t part_bytes = (size_t)BM * BN * 4;
    const auto rsP = __builtin_amdgcn_make_buffer_rsrc((void*)(d.sk_part + (size_t)sk_tile * d.sk_parts * (size_t)(BM * BN)), 0,
                                                       (int)(d.sk_parts * part_bytes), 0x00020000);
    const int lane_off = tid * 16;
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
      for (int j = 0; j < NR; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const u32x4 v = {__float_as_uint(acc[i][j][4 * q]), __float_as_uint(acc[i][j][4 * q + 1]),
                           __float_as_uint(acc[i][j][4 * q + 2]), __float_as_uint(acc[i][j][4 * q + 3])};
          __builtin_amdgcn_raw_buffer_store_b128(v, rsP, (int)(sk_part * part_bytes) + ((i * NR + j) * 4 + q) * NT * 16 + lane_off, 0, kSc);
        }
    if (!p2phd::fold_arrive_last(d.sk_ticket + sk_tile, (unsigned)d.sk_parts)) return;
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
      for (int j = 0; j < NR; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    // pieces in flight per batch: all of a part where the registers allow, 8 for the 128-accumulator tile (it spills otherwise)
    constexpr int CH = MR * NR > 6 ? 8 : NQ;
    static_assert(NQ % CH == 0, "piece batches");
    for (int pp = 0; pp < d.sk_parts; ++pp) {
#pragma unroll
      for (int q0 = 0; q0 < NQ; q0 += CH) {
        u32x4 v[CH];
#pragma unroll
        for (int q = 0; q < CH; ++q) v[q] = __builtin_amdgcn_raw_buffer_load_b128(rsP, (int)(pp * part_bytes) + (q0 + q) * NT * 16 + lane_off, 0, kSc);
#pragma unroll
        for (int q = 0; q < CH; ++q) {
          const int ij = (q0 + q) >> 2, qq = (q0 + q) & 3;
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[ij / NR][ij % NR][4 * qq + e] += __uint_as_float(v[q][e]);
        }
        __builtin_amdgcn_sched_barrier(0);                      // keep the batches apart (hoisting every load at once spills)
      }
    }
    __syncthreads();
  }

  // ---- epilogue: bias, InstanceNorm partial sums, activation, LDS-staged coalesced store ----
  constexpr int CROW = BN * (int)sizeof(TO) + 16;           // padded C-tile row
  float oscale = 1.f;                                        // fp8: de-quantisation factor of the packed weights
  if constexpr (sizeof(T) == 1) oscale = *d.out_scale;
  char* ct = stages;
  // The epilogue is VALU-bound (64-192 accumulators per lane, two waves per SIMD), so its per-element work is chosen
  // ONCE per tile: ACT = tanh | slope family (ReLU / LeakyReLU as one select) | identity (every layer that wants
  // statistics: its activation runs after the normalisation), and FULL = every tile row is a pixel of the sample (no
  // row masks in the sums; all tiles but a sample's last).  A per-element switch costs a dozen scalar branches per
  // value and keeps the tanh expansion in every element's path.
  const float neg_slope = act == P2PHD_ACT_RELU ? 0.f : (act == P2PHD_ACT_LRELU ? 0.2f : 1.f);
  constexpr int ACT_IDENT = -1;
  auto stage_tile = [&](auto act_tag, auto full_tag) {
    constexpr int ACT = decltype(act_tag)::value;
    constexpr bool FULL = decltype(full_tag)::value;
#pragma unroll
    for (int j = 0; j < NR; ++j) {
      const int col = wn * (NR * 32) + j * 32 + lr;
      int k = n0 + col, kcls = 0;
      if (cls_cp > 0) {                                        // merged sub-pixel classes share bias / statistics of channel k
        if (k >= n_extent) k = Kout;
        else { kcls = (k >= cls_cp) + (k >= 2 * cls_cp) + (k >= 3 * cls_cp); k -= kcls * cls_cp; }
      }
      const float bv = (bias != nullptr && k < Kout) ? bias[k] : 0.f;
      float s1 = 0.f;
#pragma unroll
      for (int i = 0; i < MR; ++i) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int row = wm * (MR * 32) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
          if constexpr (sizeof(T) == 1) acc[i][j][e] *= oscale;
          float v = acc[i][j][e] + bv;
          if (FULL || p_base + row < p_end) s1 += v;
          if constexpr (ACT == P2PHD_ACT_TANH) v = tanhf(v);
          else if constexpr (ACT != ACT_IDENT) v = v > 0.f ? v : neg_slope * v;   // none / ReLU / LeakyReLU(0.2) as one select
          *reinterpret_cast<TO*>(ct + row * CROW + col * (int)sizeof(TO)) = from_f<TO>(v);
        }
      }
      if (stats != nullptr) {
        // InstanceNorm partial of this wave's MR*32 rows: (sum, sum of squared deviations from the wave's OWN mean),
        // stored plainly in the wave's slot of a [N][slots][classes][Cp][2] table that a small kernel merges with Chan's
        // update.  No float atomics (bit-reproducible), and no E[x^2] - E[x]^2 cancellation: a dB spectrogram puts
        // |mean| / sigma up to 25 in front of the first InstanceNorm, which costs that formula 3 digits in fp32.
        s1 += __shfl_xor(s1, 32);
        const int first = p_base + wm * (MR * 32);
        const int cnt = FULL ? MR * 32 : min(max(p_end - first, 0), MR * 32);
        const float mean_w = cnt > 0 ? s1 / (float)cnt : 0.f;
        float m2 = 0.f;
#pragma unroll
        for (int i = 0; i < MR; ++i) {
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int row = wm * (MR * 32) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
            const float dlt = acc[i][j][e] + bv - mean_w;
            if (FULL || p_base + row < p_end) m2 += dlt * dlt;
          }
        }
        m2 += __shfl_xor(m2, 32);
        if (lh == 0 && k < Kout && cnt > 0) {                  // waves past the sample's last row own no slot
          const int slot = first / (MR * 32);
          const int ncls = cls_cp > 0 ? 4 : 1;
          float* sp = stats + 2 * ((((size_t)n * stats_slots + slot) * ncls + kcls) * Cp_out + k);
          sp[0] = s1;
          sp[1] = m2;
        }
      }
    }
  };
  {
    typedef std::true_type Y;
    typedef std::false_type N_;
    typedef std::integral_constant<int, P2PHD_ACT_TANH> Tanh;
    typedef std::integral_constant<int, P2PHD_ACT_RELU> Slope;
    typedef std::integral_constant<int, ACT_IDENT> Ident;
    if constexpr (MR * NR <= 6) {
      const bool full = p_base + BM <= p_end;
      if (act == P2PHD_ACT_TANH) stage_tile(Tanh{}, N_{});
      else if (act == P2PHD_ACT_NONE) { if (full) stage_tile(Ident{}, Y{}); else stage_tile(Ident{}, N_{}); }
      else { if (full) stage_tile(Slope{}, Y{}); else stage_tile(Slope{}, N_{}); }
    } else {
      // the 128-accumulator tile keeps two instances: more straight-line copies cost it registers (it spills)
      if (act == P2PHD_ACT_TANH) stage_tile(Tanh{}, N_{});
      else stage_tile(Slope{}, N_{});
    }
  }
#ifdef P2PHD_PROBE_FINE
  const unsigned long long pf_c = __builtin_readcyclecounter();
#endif
  __syncthreads();
#ifdef P2PHD_PROBE_FINE
  const unsigned long long pf_d = __builtin_readcyclecounter();
#endif
  constexpr int CPR = BN / EPPO;                             // 16-byte pieces per C-tile row
  const int Hout = d.Hout, Wout = d.Wout, ohm = d.oh_mul, oho = d.oh_off, owm = d.ow_mul, owo = d.ow_off;
  if constexpr (MR * NR <= 6 && sizeof(T) != 1) {
    if (d.bs_out != nullptr || d.as_x != nullptr) {
      // Store loop with the consumer's InstanceNorm-backward sums riding on it (GDesc::bs_out) -- or, for a producer
      // without normalisation, just its activation derivative applied to the stored gradient (GDesc::as_x).  A thread keeps ONE
      // piece column (8 / 4 channels) for all its rows, so the channel constants are loaded once and the sums stay in
      // registers; they are folded over the threads of a column through LDS in a fixed order (no atomics) and leave as
      // this tile's row of the partial table.
      constexpr int RG = NT / CPR;                           // threads per piece column (the last NT % CPR threads idle)
      const int pcb = tid % CPR, rgb = tid / CPR;
      const int kb = n0 + pcb * EPPO;
      int kch = kb, clsb = 0;
      if (cls_cp > 0) { clsb = (kb >= cls_cp) + (kb >= 2 * cls_cp) + (kb >= 3 * cls_cp); kch = kb - clsb * cls_cp; }
      const bool col_ok = kb < n_extent && rgb < RG;
      float a1[EPPO], a2[EPPO], mean_b[EPPO], rstd_b[EPPO];
#pragma unroll
      for (int e = 0; e < EPPO; ++e) {
        a1[e] = a2[e] = 0.f;
        const bool ch_ok = col_ok && kch + e < Kout && d.bs_out != nullptr;
        const float2 ms = ch_ok ? *reinterpret_cast<const float2*>(d.bs_stats + 2 * ((size_t)n * Cp_out + kch + e)) : make_float2(0.f, 0.f);
        mean_b[e] = ms.x;
        rstd_b[e] = ch_ok ? rsqrtf(fmaxf(ms.y * d.bs_inv_hw, 0.f) + d.bs_eps) : 0.f;
      }
      const bool act_only = d.bs_out == nullptr;
      const TO* bsy = reinterpret_cast<const TO*>(act_only ? d.as_x : d.bs_y);
      const float slope_b = d.bs_slope;
      if (col_ok) {
        // U rows at a time: their pre-normalisation pieces (and addends) are all requested before the first one is used --
        // a load consumed in the iteration that issues it costs a memory round trip per row
        constexpr int U = 4;
        for (int row0 = rgb; row0 < BM; row0 += RG * U) {
          uint4 yv[U], av[U];
          size_t opx[U];
          bool ok[U];
          int rowu[U];
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const int row = row0 + u * RG;
            rowu[u] = min(row, BM - 1);
            const int2 ri = rinfo[rowu[u]];
            int ho = ri.y >> 16, wo = ri.y & 0xFFFF;
            ok[u] = row < BM && ri.x >= 0;
            if (cls_cp > 0) {
              ho = 2 * ho + (clsb >> 1); wo = 2 * wo + (clsb & 1);
              ok[u] = ok[u] && ho < Hout && wo < Wout;
            }
            opx[u] = ok[u] ? ((size_t)ri.x * Hout + (ho * ohm + oho)) * Wout + (wo * owm + owo) : 0;   // clamped: always loadable
            yv[u] = *reinterpret_cast<const uint4*>(bsy + opx[u] * Cp_out + kch);
            if (addend != nullptr) av[u] = *reinterpret_cast<const uint4*>(addend + opx[u] * Cp_out + kch);
          }
#pragma unroll
          for (int u = 0; u < U; ++u) {
            if (!ok[u]) continue;
            uint4 v = *reinterpret_cast<const uint4*>(ct + rowu[u] * CROW + pcb * 16);
            if (addend != nullptr) {
              TO* vv = reinterpret_cast<TO*>(&v);
              const TO* aa = reinterpret_cast<const TO*>(&av[u]);
#pragma unroll
              for (int e = 0; e < EPPO; ++e) vv[e] = from_f<TO>(to_f(vv[e]) + to_f(aa[e]));
            }
            const TO* yy = reinterpret_cast<const TO*>(&yv[u]);
            if (act_only) {
              TO* vv = reinterpret_cast<TO*>(&v);
#pragma unroll
              for (int e = 0; e < EPPO; ++e) vv[e] = from_f<TO>(to_f(vv[e]) * (to_f(yy[e]) > 0.f ? 1.f : slope_b));
              *reinterpret_cast<uint4*>(out + opx[u] * Cp_out + kch) = v;
              continue;
            }
            *reinterpret_cast<uint4*>(out + opx[u] * Cp_out + kch) = v;
            const TO* gg = reinterpret_cast<const TO*>(&v);      // the ROUNDED gradient: what the apply pass will read
#pragma unroll
            for (int e = 0; e < EPPO; ++e) {
              const float yh = (to_f(yy[e]) - mean_b[e]) * rstd_b[e];
              const float gp = to_f(gg[e]) * (yh > 0.f ? 1.f : slope_b);
              a1[e] += gp; a2[e] += gp * yh;
            }
          }
        }
      }
      if (act_only) return;
      __syncthreads();                                        // every thread is done with the C tile
      float* red = reinterpret_cast<float*>(ct);              // [NT][2 * EPPO]
#pragma unroll
      for (int e = 0; e < EPPO; ++e) { red[tid * (2 * EPPO) + e] = a1[e]; red[tid * (2 * EPPO) + EPPO + e] = a2[e]; }
      __syncthreads();
      const int tile_in_sample = bx - n * mtiles;
      for (int t = tid; t < 2 * BN; t += NT) {
        const int col = t >> 1, which = t & 1, pc = col / EPPO, e = col - pc * EPPO;
        float sum = 0.f;
        for (int rg = 0; rg < RG; ++rg) sum += red[(rg * CPR + pc) * (2 * EPPO) + which * EPPO + e];
        if (n0 + col < n_extent)
          d.bs_out[(((size_t)n * mtiles + tile_in_sample) * n_extent + n0 + col) * 2 + which] = sum;
      }
      return;
    }
  }
  // U pieces per thread at a time: their row records, staged pieces (and addends) are all requested before the first one
  // is used -- one piece per iteration costs two LDS round trips and a branch per 16 bytes stored (2 waves per SIMD: nobody
  // to hide them behind)
  {
    constexpr int TOTAL = BM * CPR, U = 4;
    for (int q0 = tid; q0 < TOTAL; q0 += NT * U) {
      int2 ri[U];
      int rowu[U], pcu[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int q = min(q0 + u * NT, TOTAL - 1);
        rowu[u] = q / CPR; pcu[u] = q - rowu[u] * CPR;
        ri[u] = rinfo[rowu[u]];
      }
      uint4 v[U], av[U];
      size_t off[U];
      bool ok[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        int k = n0 + pcu[u] * EPPO;
        ok[u] = q0 + u * NT < TOTAL && ri[u].x >= 0 && k < n_extent;
        int ho = ri[u].y >> 16, wo = ri[u].y & 0xFFFF;
        if (cls_cp > 0) {                                        // class (pi,pj) -> output pixel (2 ho + pi, 2 wo + pj)
          const int cls = (k >= cls_cp) + (k >= 2 * cls_cp) + (k >= 3 * cls_cp);
          k -= cls * cls_cp;
          ho = 2 * ho + (cls >> 1); wo = 2 * wo + (cls & 1);
          ok[u] = ok[u] && ho < Hout && wo < Wout;
        }
        off[u] = ok[u] ? (((size_t)ri[u].x * Hout + (ho * ohm + oho)) * Wout + (wo * owm + owo)) * Cp_out + k : 0;   // clamped: always loadable
        v[u] = *reinterpret_cast<const uint4*>(ct + rowu[u] * CROW + pcu[u] * 16);
        if (addend != nullptr) av[u] = *reinterpret_cast<const uint4*>(addend + off[u]);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (addend != nullptr) {
          TO* vv = reinterpret_cast<TO*>(&v[u]);
          const TO* aa = reinterpret_cast<const TO*>(&av[u]);
#pragma unroll
          for (int e = 0; e < EPPO; ++e) vv[e] = from_f<TO>(to_f(vv[e]) + to_f(aa[e]));
        }
        if (ok[u]) *reinterpret_cast<uint4*>(out + off[u]) = v[u];
      }
    }
  }
#ifdef P2PHD_PROBE
#ifdef P2PHD_PROBE_DRAIN
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // include the completion of this tile's stores
#endif
  const unsigned long long pr_t3 = __builtin_readcyclecounter();
  if (tid == 0) {
    const unsigned wg = (unsigned)blockIdx.x % kProbeSlots;
    unsigned long long* r = g_probe + (size_t)wg * 8;
#ifdef P2PHD_PROBE_FINE
    // prologue: table build | descriptor + fragment addresses + DMA issue | first wait + barrier + first fragments;
    // epilogue: statistics + LDS staging | barrier | store loop (the barrier after the K loop is in the first)
    r[0] += pf_a - pr_t0; r[1] += pf_b - pf_a; r[2] += pr_t1_ - pf_b; r[3] += pr_t2 - pr_t1_;
    r[4] += pf_c - pr_t2; r[5] += pf_d - pf_c; r[6] += 1ull; r[7] += pr_t3 - pf_d;
#else
    r[0] += pr_wait_; r[1] += pr_bar_; r[2] += pr_comp_; r[3] += (unsigned long long)nsteps_;
    r[4] += pr_t1_ - pr_t0; r[5] += pr_t3 - pr_t2; r[6] += 1ull; r[7] += pr_t3 - pr_t0;
#endif
  }
#endif
}

// ------------------------------------------------------------------------------------------------------
// weight gradient:  dWp[split][m][t*Cg + c] = sum_{p in split} rows[p][m] * gather[pix(p,t)][c]
//   rows   : [N*Hg*Wg][Cp_r]   the tensor on the pixel grid (dy for Conv2d, x for ConvTranspose2d)
//   gather : [N,Hin,Win,Cp_in] the tensor reached through the taps
// Tile TM x 256 (TM = 128, or 32 for folded 2-channel layers), 64 (bf16) / 32 (f32) pixels per K step, 8 waves,
// 3-slot LDS ring fed by buffer_load ... lds with the next-but-one tile's pieces issued between MFMA clusters.
// The pixel reduction is split over blockIdx.z; every split writes its own slab (plain stores) and the unpack
// kernel adds the slabs in a fixed order: no float atomics, bit-reproducible gradients.
// ------------------------------------------------------------------------------------------------------
template <typename T, int TM>
__global__ __launch_bounds__(512) void wgrad_kernel(const GDesc d, const T* __restrict__ rows, const T* __restrict__ gat,
                                                    float* __restrict__ dwp, int Cp_r, int steps_per_split, long slab_elems,
                                                    unsigned rows_bytes, int grid_nx, int grid_my, int grid_sp, int xcd_order) {
  constexpr int EPP = Elem<T>::EPP;
  constexpr int SZ = (int)sizeof(T);
  constexpr int NT = 512;
  constexpr int BKP = SZ == 2 ? 64 : 32;                    // pixels per K-step
  constexpr int TN = 256;
  static_assert(TM == 256 || TM == 128 || TM == 32, "row tile");
  constexpr int WAVES_M = TM == 32 ? 1 : 2, WAVES_N = 8 / WAVES_M;
  constexpr int MI = TM / WAVES_M / 32, NI = TN / WAVES_N / 32;
  // LDS image: both operands are stored as PANELS of [BKP pixel rows][128 bytes] (64 bf16 / 32 f32 columns), the same
  // shape as the gconv tiles: a wave instruction of the direct-to-LDS load fills 8 rows of one panel linearly and
  // every thread owns ONE pixel row (all its pieces are that pixel at different column panels), so the gather
  // coordinates are advanced once per thread and K step.
  constexpr int PANEL = BKP * 128;
  constexpr int CPP = 128 / SZ;                             // columns per panel
  constexpr int GROUPS = NT / (8 * BKP);                    // 1 (bf16) / 2 (f32) thread groups per row set
  constexpr int NPG = TN / CPP;                             // gather panels: 4 / 8
  constexpr int PPT = NPG / GROUPS;                         // gather pieces per thread
  static_assert(PPT == 4, "four gather pieces per thread");
  constexpr bool kNarrowA = TM == 32 && SZ == 2;            // rows tile [64][64 B]: half-panel rows, own mapping
  constexpr int NPA = kNarrowA ? 1 : (TM / CPP);            // rows-operand panels
  constexpr int PPTA = kNarrowA ? 1 : (NPA >= GROUPS ? NPA / GROUPS : 1);
  constexpr int TILEA = kNarrowA ? BKP * 64 : NPA * PANEL;
  constexpr int TILE = NPG * PANEL;
  constexpr int STAGE = TILE + TILEA;
  constexpr int NSTAGE = TM == 256 ? 2 : 3;
  constexpr int NLOADS = PPT + PPTA;
  constexpr unsigned kOOB = 0xFFFFFFF0u;

  extern __shared__ float4 smem_raw[];
  char* smem = reinterpret_cast<char*>(smem_raw);
  // loop-resident descriptor fields in registers (see gconv_kernel)
  const int Hg = d.Hg, Wg = d.Wg, Hin = d.Hin, Win = d.Win, Cpi = d.Cp_in, sh = d.sh, sw = d.sw, pad_mode = d.pad_mode, KK = d.KK;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  // 1-D launch of grid_nx (column tiles) x grid_my (row tiles) x grid_sp (pixel splits) workgroups.  Which tile a workgroup
  // takes decides what shares an XCD's L2: workgroups are dealt round-robin over the 8 XCDs (speed only, never
  // correctness), so physical id -> logical index L puts a CONTIGUOUS run of L on each XCD (bijective chunk remap), and L
  // orders the tiles so that neighbours stream the same bytes at the same time: same pixel split first, then the same
  // input-channel slice (column tiles jx = tap * slices + slice read the same pixels of the gathered tensor through
  // different taps), then tap, then row tile (same rows-operand panel).  The 243 workgroups of a trunk layer then read
  // each activation panel from memory about twice instead of 8 times (measured: DESIGN section 6).
  int bx, by, bz;
  {
    const int W = grid_nx * grid_my * grid_sp;
    int L = (int)blockIdx.x;
    if (xcd_order) {
      const int q = W >> 3, r = W & 7, xcd = L & 7, k = L >> 3;
      L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
      by = L % grid_my;
      const int u = L / grid_my;
      const int slices = (d.Cp_in * SZ) % (TN * SZ) == 0 ? d.Cp_in / TN : 0;
      if (slices > 0 && grid_nx % slices == 0 && grid_nx * TN == d.KK) {
        const int taps = grid_nx / slices;
        const int t = u % taps, v = u / taps;
        bx = t * slices + v % slices;
        bz = v / slices;
      } else {
        bx = u % grid_nx;
        bz = u / grid_nx;
      }
    } else {
      bx = L % grid_nx;
      by = (L / grid_nx) % grid_my;
      bz = L / (grid_nx * grid_my);
    }
  }
  const int j0 = bx * TN;                                   // first kk column
  const int m0 = by * TM;                                   // first output row
  const int npix = Hg * Wg;
  const long P = (long)d.N * npix;
  const int total_steps = (int)((P + BKP - 1) / BKP);
  const int s_begin = bz * steps_per_split;
  int s_end = s_begin + steps_per_split;
  if (s_end > total_steps) s_end = total_steps;
  const int nsteps = s_end - s_begin;                       // >= 1 by construction of the grid

  // this thread's pixel row, slot and panel group; bf16 tiles are read back with the transposing ds_read_b64_tr_b16,
  // whose 32-lane half touches 4 pixel rows x 64 B at a 128-byte row pitch: rows 2,3 (mod 4) are moved to the other
  // half of the row by XORing the 16-byte chunk index with 4 (applied to the SOURCE column, the LDS side is linear)
  const int row = (tid >> 3) & (BKP - 1), slot = tid & 7, grp = tid / (8 * BKP);
  const int chunk = SZ == 2 ? (slot ^ (((row >> 1) & 1) << 2)) : slot;
  const int wrow8 = 8 * (wave % (BKP / 8));                 // first tile row of this wave's instruction
  typedef __attribute__((address_space(3))) void* lds_ptr;
  const auto rsG = __builtin_amdgcn_make_buffer_rsrc((void*)gat, 0, (int)d.in_bytes, 0x00020000);
  const auto rsA = __builtin_amdgcn_make_buffer_rsrc((void*)rows, 0, (int)rows_bytes, 0x00020000);

  // gather pieces: column -> (tap, channel), fixed per thread
  const int T_taps = d.nth * d.ntw;
  int g_dh[PPT], g_dw[PPT];
  unsigned g_cB[PPT];
  bool g_ok[PPT];
#pragma unroll
  for (int j = 0; j < PPT; ++j) {
    const int kk = j0 + (grp * PPT + j) * CPP + chunk * EPP;
    const int t = kk / Cpi;
    g_ok[j] = t < T_taps;
    const int ta = t / d.ntw, tb = t - ta * d.ntw;
    g_dh[j] = d.dh0 + ta * d.dh_step; g_dw[j] = d.dw0 + tb * d.dw_step;
    g_cB[j] = (unsigned)((kk - t * Cpi) * SZ);
  }
  unsigned g_offB[PPT];                                       // (dh * Win + dw) * bytes per pixel + channel offset, mod 2^32
#pragma unroll
  for (int j = 0; j < PPT; ++j) g_offB[j] = (unsigned)(g_dh[j] * Win + g_dw[j]) * (unsigned)(Cpi * SZ) + g_cB[j];
  const bool same_tap = g_ok[0] && g_ok[PPT - 1] && g_dh[0] == g_dh[PPT - 1] && g_dw[0] == g_dw[PPT - 1] &&
                        (j0 + (grp * PPT) * CPP + chunk * EPP) / Cpi == (j0 + (grp * PPT + PPT - 1) * CPP + chunk * EPP) / Cpi;
  // rows-operand pieces
  unsigned a_cB[PPTA];
  bool a_ok[PPTA];
  int rowA = row;
  int waveA8 = wrow8;
  if constexpr (kNarrowA) {
    const int tidA = tid & 255;
    rowA = tidA >> 2;
    const int mcol = m0 + (tidA & 3) * EPP;
    a_ok[0] = mcol < Cp_r; a_cB[0] = (unsigned)(mcol * SZ);
    waveA8 = 16 * (wave & 3);
  } else {
#pragma unroll
    for (int j = 0; j < PPTA; ++j) {
      const int panel = NPA >= GROUPS ? grp * PPTA + j : 0;
      const int mcol = m0 + panel * CPP + chunk * EPP;
      a_ok[j] = mcol < Cp_r; a_cB[j] = (unsigned)(mcol * SZ);
    }
  }
  const unsigned CpiB = (unsigned)(Cpi * SZ), CprB = (unsigned)(Cp_r * SZ);

  // pixel of this thread's row, advanced by BKP per K step
  long pcur = (long)s_begin * BKP + row;
  int pn, ph, pw;
  {
    const long nn = pcur / npix;
    const int rem = (int)(pcur - nn * npix);
    pn = (int)nn; ph = rem / Wg; pw = rem - ph * Wg;
  }
  long pA = (long)s_begin * BKP + rowA;
  unsigned aB = (unsigned)pA * CprB;                          // byte offset of this thread's rows-operand pixel, advanced per step
  unsigned vG[PPT], vA[PPTA];
  // one K step moves the pixel by BKP: as (samples, rows, columns) so the walk is three adds with carries, no loops
  const int adv_n = BKP / npix, adv_rem = BKP - adv_n * npix;
  const int adv_h = adv_rem / Wg, adv_w = adv_rem - adv_h * Wg;
  const bool reflect = pad_mode == 1;
  // 24-bit multiplies are full rate (v_mad_u32_u24); the host guarantees N * Hin * Win < 2^31 and Hin, Win < 2^24
  auto pix_off = [&](int dh, int dw) -> unsigned {
    int hi = __mul24(ph, sh) + dh, wi = __mul24(pw, sw) + dw;
    if (reflect) {                                              // branch-free |.| and mirror at the far edge
      hi = hi < 0 ? -hi : hi; hi = hi >= Hin ? 2 * (Hin - 1) - hi : hi;
      wi = wi < 0 ? -wi : wi; wi = wi >= Win ? 2 * (Win - 1) - wi : wi;
    }
    const bool ok = (unsigned)hi < (unsigned)Hin && (unsigned)wi < (unsigned)Win;
    const unsigned pix = (unsigned)(__mul24(pn, Hin) + hi) * (unsigned)Win + (unsigned)wi;
    return ok ? pix * CpiB : kOOB;
  };
  auto prepare = [&]() {
    if (pcur < P) {
      if (same_tap) {
        const unsigned o = pix_off(g_dh[0], g_dw[0]);
#pragma unroll
        for (int j = 0; j < PPT; ++j) vG[j] = o == kOOB ? kOOB : o + g_cB[j];
      } else if (!reflect) {
        // zero padding: every tap is the un-shifted pixel plus a per-piece constant; only the bounds test is per tap
        const int hb = __mul24(ph, sh), wb = __mul24(pw, sw);
        const unsigned baseB = ((unsigned)(__mul24(pn, Hin) + hb) * (unsigned)Win + (unsigned)wb) * CpiB;
#pragma unroll
        for (int j = 0; j < PPT; ++j) {
          const bool ok = g_ok[j] && (unsigned)(hb + g_dh[j]) < (unsigned)Hin && (unsigned)(wb + g_dw[j]) < (unsigned)Win;
          vG[j] = ok ? baseB + g_offB[j] : kOOB;
        }
      } else {
#pragma unroll
        for (int j = 0; j < PPT; ++j) {
          const unsigned o = g_ok[j] ? pix_off(g_dh[j], g_dw[j]) : kOOB;
          vG[j] = o == kOOB ? kOOB : o + g_cB[j];
        }
      }
    } else {
#pragma unroll
      for (int j = 0; j < PPT; ++j) vG[j] = kOOB;
    }
#pragma unroll
    for (int j = 0; j < PPTA; ++j) vA[j] = (a_ok[j] && pA < P) ? aB + a_cB[j] : kOOB;
    pcur += BKP; pA += BKP; aB += (unsigned)BKP * CprB;
    pw += adv_w;
    const int cw = pw >= Wg ? 1 : 0;
    pw -= cw ? Wg : 0;
    ph += adv_h + cw;
    const int ch = ph >= Hg ? 1 : 0;
    ph -= ch ? Hg : 0;
    pn += adv_n + ch;
  };
  // piece j of a tile: 0..3 gather panels, 4.. rows-operand panels
  P2PHD_CW_DECL;
  auto issue_piece = [&](int slot_, int j) {
    char* A = smem + slot_ * STAGE;
    P2PHD_CW_ISSUE(slot_);
    if (j < PPT) {
      char* G = A + TILEA + (grp * PPT + j) * PANEL + wrow8 * 128;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsG, (lds_ptr)G, 16, (int)vG[j], 0, 0, 0);
    } else {
      char* Aw;
      if constexpr (kNarrowA) Aw = A + waveA8 * 64;
      else Aw = A + (NPA >= GROUPS ? grp * PPTA + (j - PPT) : 0) * PANEL + waveA8 * 128;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr)Aw, 16, (int)vA[j - PPT], 0, 0, 0);
    }
  };

  f32x16 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const unsigned sbase = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  // transposing LDS read: 16-lane group g reads a 4-pixel x 16-channel block, lane i gets channel i; lane 4q+p of the
  // group supplies row (8h + q), 8-byte column unit u = 4*(g&1) + p of the 32-column block (u>>1 = 16-B chunk)
  const int g16 = lane >> 4, i16 = lane & 15, q4 = i16 >> 2, pq = i16 & 3, hh = g16 >> 1;
  const int u8 = 4 * (g16 & 1) + pq;
  const int swzq = ((q4 >> 1) & 1) << 2;
  constexpr int RPA = kNarrowA ? 64 : 128;                  // row pitch of the rows-operand tile
  unsigned ta_off[MI], tg_off[NI];                          // byte offsets (within a stage) of the sub = 0 reads
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int mb = wm * (MI * 32) + i * 32;
    if constexpr (kNarrowA) ta_off[i] = (unsigned)((8 * hh + q4) * 64 + ((u8 >> 1) << 4) + 8 * (u8 & 1));
    else ta_off[i] = (unsigned)((mb / CPP) * PANEL + (8 * hh + q4) * 128 + (((((mb % CPP) >> 3) + (u8 >> 1)) ^ swzq) << 4) + 8 * (u8 & 1));
  }
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const int nb = wn * (NI * 32) + j * 32;
    tg_off[j] = (unsigned)(TILEA + (nb / CPP) * PANEL + (8 * hh + q4) * 128 + (((((nb % CPP) >> 3) + (u8 >> 1)) ^ swzq) << 4) + 8 * (u8 & 1));
  }

  if constexpr (SZ == 2) {
    // Same pipeline as gconv_kernel's main loop: one barrier per K step, in front of its last MFMA cluster; the next
    // tile's first fragments and the LDS-DMA of tile s + NSTAGE (into the slot just drained) go out in the MFMA shadow.
    uint2 af[2][MI][2], gf[2][NI][2];
    // the k sub-step and the second half of a fragment ride on the instruction's immediate offset: one address VGPR
    // per fragment and K step instead of one add per read (`sub` is a literal after unrolling, the switch folds away)
#define P2PHD_TR_READ(dst, addr, OFF) \
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF))
#define P2PHD_TR_PAIR(lo, hi, addr, SUB, PITCH)                                                        \
  do {                                                                                                 \
    P2PHD_TR_READ(lo, addr, 16 * (SUB) * (PITCH));                                                     \
    P2PHD_TR_READ(hi, addr, 16 * (SUB) * (PITCH) + 4 * (PITCH));                                       \
  } while (0)
    auto read_frags = [&](unsigned so, int sub, int buf) {
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const unsigned ad = so + ta_off[i];
        switch (sub) {
          case 0: P2PHD_TR_PAIR(af[buf][i][0], af[buf][i][1], ad, 0, RPA); break;
          case 1: P2PHD_TR_PAIR(af[buf][i][0], af[buf][i][1], ad, 1, RPA); break;
          case 2: P2PHD_TR_PAIR(af[buf][i][0], af[buf][i][1], ad, 2, RPA); break;
          default: P2PHD_TR_PAIR(af[buf][i][0], af[buf][i][1], ad, 3, RPA); break;
        }
      }
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const unsigned ad = so + tg_off[j];
        switch (sub) {
          case 0: P2PHD_TR_PAIR(gf[buf][j][0], gf[buf][j][1], ad, 0, 128); break;
          case 1: P2PHD_TR_PAIR(gf[buf][j][0], gf[buf][j][1], ad, 1, 128); break;
          case 2: P2PHD_TR_PAIR(gf[buf][j][0], gf[buf][j][1], ad, 2, 128); break;
          default: P2PHD_TR_PAIR(gf[buf][j][0], gf[buf][j][1], ad, 3, 128); break;
        }
      }
    };
    static_assert(BKP / 16 <= 4, "sub-step switch covers 4 k sub-steps");
    auto mfma_one = [&](int buf, int i, int j) {
      bf16x8 a8, g8;
      uint2* ap = reinterpret_cast<uint2*>(&a8);
      uint2* gp = reinterpret_cast<uint2*>(&g8);
      ap[0] = af[buf][i][0]; ap[1] = af[buf][i][1];
      gp[0] = gf[buf][j][0]; gp[1] = gf[buf][j][1];
      acc[i][j] = p2phd_mfma_32x32x16(a8, g8, acc[i][j]);
    };
    constexpr int NSUB = BKP / 16;
#ifdef P2PHD_PROBE
    const unsigned long long pr_t0 = __builtin_readcyclecounter();
    unsigned long long pr_wait = 0, pr_bar = 0;
#endif
#pragma unroll
    for (int t = 0; t < NSTAGE; ++t) {
      if (t < nsteps) {
        prepare();
#pragma unroll
        for (int j = 0; j < NLOADS; ++j) issue_piece(t, j);
      }
    }
    if (nsteps >= NSTAGE) {
      P2PHD_CW_WAIT(CW_WGRAD, (NSTAGE - 1) * NLOADS, 1u << 0);
      asm volatile("s_waitcnt vmcnt(%0)" :: "n"((NSTAGE - 1) * NLOADS) : "memory");
    } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    read_frags(sbase, 0, 0);
    int cur = 0;
    bool pend = false;
    int pend_slot = 0;
    for (int s = 0; s < nsteps; ++s) {
      const unsigned so = sbase + (unsigned)(cur * STAGE);
      const int nslot = cur == NSTAGE - 1 ? 0 : cur + 1;
      const bool has_next = s + 1 < nsteps;
      const bool issue_new = s + NSTAGE < nsteps;
#pragma unroll
      for (int sub = 0; sub < NSUB; ++sub) {
        const int buf = sub & 1;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (sub == NSUB - 1 && has_next) {
#ifdef P2PHD_PROBE
          const unsigned long long q0 = __builtin_readcyclecounter();
#endif
          if (NSTAGE > 2 && s + NSTAGE - 1 < nsteps) {
            P2PHD_CW_WAIT(CW_WGRAD, (NSTAGE - 2) * NLOADS, 1u << nslot);
            asm volatile("s_waitcnt vmcnt(%0)" :: "n"((NSTAGE - 2) * NLOADS) : "memory");
          } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef P2PHD_PROBE
          const unsigned long long q1 = __builtin_readcyclecounter();
#endif
          __builtin_amdgcn_s_barrier();
#ifdef P2PHD_PROBE
          pr_wait += q1 - q0; pr_bar += __builtin_readcyclecounter() - q1;
#endif
        }
        __builtin_amdgcn_sched_barrier(0);
        mfma_one(buf, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (sub < NSUB - 1) read_frags(so, sub + 1, buf ^ 1);
        else if (has_next) read_frags(sbase + (unsigned)(nslot * STAGE), 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (MI * NI > 1) mfma_one(buf, 1 / NI, 1 % NI);
        __builtin_amdgcn_sched_barrier(0);
        if (sub == 0 && pend) {
#pragma unroll
          for (int j = 1; j < NLOADS; j += 2) issue_piece(pend_slot, j);
          pend = false;
        }
        if (sub == NSUB - 1 && issue_new) {
          prepare();
#pragma unroll
          for (int j = 0; j < NLOADS; j += 2) issue_piece(cur, j);
          pend = true; pend_slot = cur;
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NI; ++j)
            if (i * NI + j >= (MI * NI > 1 ? 2 : 1)) mfma_one(buf, i, j);
        __builtin_amdgcn_sched_barrier(0);
      }
      cur = nslot;
    }
#ifdef P2PHD_PROBE
    if (tid == 0) {
      const unsigned wg = (unsigned)blockIdx.x % kProbeSlots;
      unsigned long long* r = g_probe + (size_t)wg * 8;
      r[0] += pr_wait; r[1] += pr_bar; r[2] += __builtin_readcyclecounter() - pr_t0; r[3] += (unsigned long long)nsteps;
      r[6] += 1ull;
    }
#endif
  } else {
    // f32 (parity runs): plain LDS reads; hipcc drains the DMA queue in front of them, which is correct, just slower
    auto compute = [&](int slot_, bool pf, int pf_slot) {
      if (pf) {
        prepare();
#pragma unroll
        for (int j = 0; j < NLOADS; ++j) issue_piece(pf_slot, j);
      }
      const char* A = smem + slot_ * STAGE;
      const char* G = A + TILEA;
      const int lr = lane & 31, lh = lane >> 5;
#pragma unroll 4
      for (int s2 = 0; s2 < BKP / 2; ++s2) {
        const int prow = 2 * s2 + lh;
        float af[MI], gf[NI];
#pragma unroll
        for (int i = 0; i < MI; ++i) {
          const int col = wm * (MI * 32) + i * 32 + lr;
          af[i] = *reinterpret_cast<const float*>(A + (col / CPP) * PANEL + prow * 128 + (col % CPP) * 4);
        }
#pragma unroll
        for (int j = 0; j < NI; ++j) {
          const int col = wn * (NI * 32) + j * 32 + lr;
          gf[j] = *reinterpret_cast<const float*>(G + (col / CPP) * PANEL + prow * 128 + (col % CPP) * 4);
        }
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NI; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], gf[j], acc[i][j], 0, 0, 0);
      }
    };
    constexpr int D = NSTAGE - 1;
#pragma unroll
    for (int t = 0; t < D; ++t) {
      if (t < nsteps) {
        prepare();
#pragma unroll
        for (int j = 0; j < NLOADS; ++j) issue_piece(t, j);
      }
    }
    int cur = 0, nxt = D;
    for (int s = 0; s < nsteps; ++s) {
      if (D >= 2 && s + 1 < nsteps) {
        P2PHD_CW_WAIT(CW_WGRAD_F32, NLOADS, 1u << cur);          // this step's slot is read behind the barrier
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NLOADS) : "memory");
      } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      compute(cur, s + D < nsteps, nxt);
      cur = cur == NSTAGE - 1 ? 0 : cur + 1;
      nxt = nxt == NSTAGE - 1 ? 0 : nxt + 1;
    }
  }

  P2PHD_CW_DONE();
  float* slab = dwp + (size_t)bz * slab_elems;
  const int lr = lane & 31, lh = lane >> 5;
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int col = j0 + wn * (NI * 32) + j * 32 + lr;
      if (col >= KK) continue;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row_o = m0 + wm * (MI * 32) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
        slab[(size_t)row_o * KK + col] = acc[i][j][e];
      }
    }
}

// ------------------------------------------------------------------------------------------------------
// weight packing: master f32 tensor (generic strides) -> Wp[rows_pad][KK] of T, zero padded
// and the inverse for gradients (packed f32 -> master layout, overwrite)
// ------------------------------------------------------------------------------------------------------
// One thread owns one (packed row, channel) pair and walks the taps with counters: its master-tensor reads are the
// contiguous R*S block of that pair (consecutive lanes = consecutive channels, so a wave covers one contiguous span), its
// packed writes are channel-contiguous per tap.  No integer division per element; block = 64 channels x 4 rows.
template <typename T>
__global__ __launch_bounds__(256) void pack_kernel(GDesc d, p2phd::WMap m, const float* __restrict__ w, T* __restrict__ wp,
                                                   int rows_pad) {
  const int row = blockIdx.y * 4 + threadIdx.y;
  if (row >= rows_pad) return;
  const int T_taps = d.nth * d.ntw, Cp = d.Cp_in, KK = d.KK, ntw = d.ntw;
  T* orow = wp + (size_t)row * KK;
  const bool row_ok = row < m.rows;
  const long roff = row_ok ? (long)(row % m.row_mod) * m.s_row + (long)(row / m.row_mod) * m.s_rowq : 0;
  for (int c = blockIdx.x * 64 + threadIdx.x; c < Cp; c += gridDim.x * 64) {
    const bool ok = row_ok && c < m.inner;
    const float* src = w + roff + (ok ? (long)(c % m.c_mod) * m.s_inner + (long)(c / m.c_mod) * m.s_innerq : 0);
    int ta = 0, tb = 0;
    for (int t0 = 0; t0 < T_taps; t0 += 16) {                   // 16 independent (clamped, unconditional) loads in flight
      float v[16];
      int ta2 = ta, tb2 = tb;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        v[i] = src[((d.wr0 + ta2 * d.wr_step) * m.S + d.ws0 + tb2 * d.ws_step) * m.s_tap];
        if (t0 + i + 1 < T_taps && ++tb2 == ntw) { tb2 = 0; ++ta2; }
      }
#pragma unroll
      for (int i = 0; i < 16; ++i)
        if (t0 + i < T_taps) orow[(t0 + i) * Cp + c] = from_f<T>(ok ? v[i] : 0.f);
      ta = ta2; tb = tb2;                                       // = tap t0 + 16 when there is another batch
    }
  }
  // zero tail of the padded K extent
  for (int kk = T_taps * Cp + blockIdx.x * 64 + threadIdx.x; kk < KK; kk += gridDim.x * 64) orow[kk] = from_f<T>(0.f);
}

// Dense variants for the common case "every tap of a plain [rows][inner][R][S] master tensor, in order" (all stride-1
// forward packs and weight gradients, i.e. almost all of the parameter bytes): the R*S values of a (row, channel) pair
// and of its 63 neighbours form ONE contiguous run of the master tensor, which is moved with coalesced accesses and
// re-ordered to / from the tap-major packed layout through a small LDS tile (T_taps is coprime to the bank count or small,
// so the strided side of the tile costs at most a few-way conflict on 16 KiB).
template <typename T>
__global__ __launch_bounds__(256) void pack_dense_kernel(GDesc d, p2phd::WMap m, const float* __restrict__ w, T* __restrict__ wp,
                                                         int rows_pad) {
  __shared__ float tile[4][64 * 16];
  const int T_taps = d.nth * d.ntw, Cp = d.Cp_in, KK = d.KK;
  const int tx = threadIdx.x, ty = threadIdx.y;
  const int row = blockIdx.y * 4 + ty, c0 = blockIdx.x * 64;
  const bool row_ok = row < m.rows;
  const int ncols = max(0, min(64, m.inner - c0));
  if (row_ok) {
    const float* src = w + (long)row * m.s_row + (long)c0 * T_taps;
    for (int i = tx; i < ncols * T_taps; i += 64) tile[ty][i] = src[i];
  }
  __syncthreads();
  if (row < rows_pad) {
    T* orow = wp + (size_t)row * KK;
    const int c = c0 + tx;
    if (c < Cp) {
      const bool ok = row_ok && tx < ncols;
      for (int t = 0; t < T_taps; ++t) orow[t * Cp + c] = from_f<T>(ok ? tile[ty][tx * T_taps + t] : 0.f);
    }
    if (blockIdx.x == 0)
      for (int kk = T_taps * Cp + tx; kk < KK; kk += 64) orow[kk] = from_f<T>(0.f);
  }
}

// TT = compile-time tap count (9: 3x3, 16: 4x4) so that exactly TT loads per slab are issued; 0 = any count <= 16
// (loads clamped to the last tap: up to 16 issued)
template <int TT>
__global__ __launch_bounds__(256) void unpack_dense_kernel(GDesc d, p2phd::WMap m, const float* __restrict__ dwp,
                                                           float* __restrict__ dw, int splits, long slab_elems, int accumulate) {
  __shared__ float tile[4][64 * 16];
  const int T_taps = TT > 0 ? TT : d.nth * d.ntw, Cp = d.Cp_in;
  constexpr int NV = TT > 0 ? TT : 16;
  const int tx = threadIdx.x, ty = threadIdx.y;
  const int row = blockIdx.y * 4 + ty, c0 = blockIdx.x * 64;
  const bool row_ok = row < m.rows;
  const int ncols = max(0, min(64, m.inner - c0));
  if (row_ok && tx < ncols) {
    float v[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = 0.f;
    const float* srow = dwp + (size_t)row * d.KK + c0 + tx;
    for (int z = 0; z < splits; ++z) {                           // fixed order: reproducible
      const float* src = srow + (size_t)z * slab_elems;
#pragma unroll
      for (int i = 0; i < NV; ++i) v[i] += src[(size_t)min(i, T_taps - 1) * Cp];
    }
#pragma unroll
    for (int i = 0; i < NV; ++i)
      if (i < T_taps) tile[ty][tx * T_taps + i] = v[i];
  }
  __syncthreads();
  if (row_ok) {
    float* dst = dw + (long)row * m.s_row + (long)c0 * T_taps;
    for (int i = tx; i < ncols * T_taps; i += 64) dst[i] = accumulate ? dst[i] + tile[ty][i] : tile[ty][i];
  }
}

// Transposing variant for master tensors laid out [inner][rows][R*S] (the input-gradient pack of a Conv2d: packed rows =
// input channels, packed inner = output channels): a block moves a 64 (inner) x 16 (rows) x R*S brick through LDS, so
// both the master reads (R*S * 16 contiguous floats per inner index) and the packed writes (64 consecutive inner
// indices) are coalesced; the generic kernel reads this case with one cache line per lane.
template <typename T>
__global__ __launch_bounds__(256) void pack_transposed_kernel(GDesc d, p2phd::WMap m, const float* __restrict__ w, T* __restrict__ wp,
                                                              int rows_pad) {
  constexpr int RB = 16;
  __shared__ float tile[64][RB * 16 + 1];
  __shared__ int tapidx[16];
  const int T_taps = d.nth * d.ntw, Cp = d.Cp_in, KK = d.KK, RS = (int)m.s_row;
  const int tx = threadIdx.x, ty = threadIdx.y;
  const int k0 = blockIdx.x * 64, row0 = blockIdx.y * RB;
  if (ty == 0 && tx < T_taps) {
    const int ta = tx / d.ntw, tb = tx - ta * d.ntw;
    tapidx[tx] = (d.wr0 + ta * d.wr_step) * m.S + d.ws0 + tb * d.ws_step;
  }
  const int nrows = max(0, min(RB, m.rows - row0));
  // 4 x 4 unconditional (clamped) loads in flight per thread and pass: a load inside a data-dependent branch is
  // serialised by its own s_waitcnt
  const int run = nrows * RS;                                    // contiguous floats per inner index (<= 256)
#pragma unroll
  for (int kb = 0; kb < (run > 0 ? 64 : 0); kb += 16) {          // run == 0: a block of padding rows reads nothing
    float v[4][4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int k = kb + 4 * kk + ty;
      const float* src = w + (long)min(k0 + k, m.inner - 1) * m.s_inner + (long)row0 * RS;
#pragma unroll
      for (int q = 0; q < 4; ++q) v[kk][q] = src[min(tx + 64 * q, max(run - 1, 0))];
    }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int k = kb + 4 * kk + ty;
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (tx + 64 * q < run) tile[k][tx + 64 * q] = v[kk][q];
    }
  }
  __syncthreads();
  const bool k_ok = k0 + tx < m.inner;
  if (k0 + tx < Cp) {
    for (int r = ty; r < RB; r += 4) {
      const int row = row0 + r;
      if (row >= rows_pad) break;
      T* orow = wp + (size_t)row * KK + k0 + tx;
      const bool ok = k_ok && r < nrows;
      for (int t = 0; t < T_taps; ++t) orow[(size_t)t * Cp] = from_f<T>(ok ? tile[tx][r * RS + tapidx[t]] : 0.f);
    }
  }
  if (blockIdx.x == 0) {                                          // zero tail of the padded K extent
    for (int r = ty; r < RB; r += 4) {
      const int row = row0 + r;
      if (row >= rows_pad) break;
      for (int kk = T_taps * Cp + tx; kk < KK; kk += 64) wp[(size_t)row * KK + kk] = from_f<T>(0.f);
    }
  }
}

inline bool transposed_map(const GDesc& d, const p2phd::WMap& m) {
  const int T_taps = d.nth * d.ntw;
  if (!(T_taps <= 16 && m.s_tap == 1 && m.s_row >= 1 && m.s_row <= 16 && m.c_mod >= m.inner && m.row_mod >= m.rows && m.inner > 0 && m.rows > 0 &&
        m.s_inner >= (long)m.rows * m.s_row))
    return false;
  for (int t = 0; t < T_taps; ++t) {                             // every tap must address inside the R*S block
    const int ta = t / d.ntw, tb = t - ta * d.ntw;
    const int idx = (d.wr0 + ta * d.wr_step) * m.S + d.ws0 + tb * d.ws_step;
    if (idx < 0 || idx >= m.s_row) return false;
  }
  return true;
}

inline bool dense_map(const GDesc& d, const p2phd::WMap& m) {
  const int T_taps = d.nth * d.ntw;
  return T_taps <= 16 && m.s_tap == 1 && m.c_mod >= m.inner && m.row_mod >= m.rows && m.s_inner == T_taps && d.wr0 == 0 && d.wr_step == 1 &&
         d.ws0 == 0 && d.ws_step == 1 && d.ntw == m.S && m.inner > 0 && m.rows > 0;
}

// ---- K-major master weights [rows = K][tap][inner = C] (p2phd_conv_desc::w_layout = 1) ------------------------------------------
// forward pack / weight gradient: the packed row [tap][Cp] is the master row (Cp == C, taps in order)
inline bool kmajor_dense_map(const GDesc& d, const p2phd::WMap& m) {
  const int T_taps = d.nth * d.ntw;
  return m.s_inner == 1 && m.s_tap == m.inner && m.s_row == (long)T_taps * m.inner && d.Cp_in == m.inner && m.c_mod >= m.inner &&
         m.row_mod >= m.rows && d.wr0 == 0 && d.wr_step == 1 && d.ws0 == 0 && d.ws_step == 1 && d.ntw == m.S && m.inner > 0 && m.rows > 0;
}
// input-gradient pack: packed rows = C (master inner index), packed inner = K (master rows): a transpose per tap
inline bool kmajor_transposed_map(const GDesc& d, const p2phd::WMap& m) {
  const int T_taps = d.nth * d.ntw;
  return T_taps <= 16 && m.s_row == 1 && m.s_tap == m.rows && m.s_inner == (long)T_taps * m.rows && m.c_mod >= m.inner &&
         m.row_mod >= m.rows && d.ntw * d.nth == T_taps && m.inner > 0 && m.rows > 0;
}

// wp[row][kk] = T(w[row][kk]) for kk < T_taps * C, zero in the K tail and in the padding rows: a cast, float4 in / 8 or 16 bytes out
template <typename T>
__global__ __launch_bounds__(256) void pack_kmajor_dense_kernel(const float* __restrict__ w, T* __restrict__ wp, int rows, int rows_pad,
                                                                int row_len, int KK) {
  const long total4 = (long)rows_pad * (KK / 4);
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total4; e += (long)gridDim.x * 256) {
    const int row = (int)(e / (KK / 4)), k4 = (int)(e - (long)row * (KK / 4)) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row < rows && k4 < row_len) v = *reinterpret_cast<const float4*>(w + (size_t)row * row_len + k4);    // row_len % 8 == 0
    T* o = wp + (size_t)row * KK + k4;
    if constexpr (sizeof(T) == 2) {
      bf16x4 b = {(bf16_t)v.x, (bf16_t)v.y, (bf16_t)v.z, (bf16_t)v.w};
      *reinterpret_cast<bf16x4*>(o) = b;
    } else {
      *reinterpret_cast<float4*>(o) = v;
    }
  }
}

// wp[c][t'][k] = T(w[k][tap(t')][c]): per packed tap a 64 (k) x 64 (c) tile through LDS.  Round 5: 16-byte accesses on both sides --
// the tile's rows are read as float4 runs along c (one 256-byte master row per 16 lanes), and every thread writes whole 16-byte
// pieces of eight consecutive k of one packed row (the first version moved 4 bytes in and 2 bytes out per lane and instruction:
// 13.5 us for 32 MB on the trunk layer).  Callers guarantee K-major shapes (kmajor_transposed_map: C and K multiples of 64 here,
// so a tile is whole unless it hangs over rows_pad / Cp, which the scalar tail below handles).
template <typename T>
__global__ __launch_bounds__(256) void pack_kmajor_transposed_kernel(GDesc d, p2phd::WMap m, const float* __restrict__ w, T* __restrict__ wp,
                                                                     int rows_pad) {
  __shared__ float tile[64][65];
  const int T_taps = d.nth * d.ntw, Cp = d.Cp_in, KK = d.KK;
  const int tid = threadIdx.y * 64 + threadIdx.x;                // (64, 4) threads
  const int k0 = blockIdx.x * 64, c0 = blockIdx.y * 64, tp = blockIdx.z;
  const int ta = tp / d.ntw, tb = tp - ta * d.ntw;
  const long tap = ((long)(d.wr0 + ta * d.wr_step) * m.S + d.ws0 + tb * d.ws_step) * m.s_tap;   // master offset of this packed tap
  // master element (k, c): w[k * s_inner + tap + c]   (m.rows = C, m.inner = K)
  const bool whole = k0 + 64 <= m.inner && c0 + 64 <= m.rows && c0 + 64 <= rows_pad && k0 + 64 <= Cp &&
                     ((m.s_inner | tap | (long)c0) & 3) == 0 && ((uintptr_t)w & 15) == 0;
  if (whole) {
    const int c4 = (tid & 15) * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int kr = (tid >> 4) + 16 * i;
      const float4 v = *reinterpret_cast<const float4*>(w + (size_t)(k0 + kr) * m.s_inner + tap + c0 + c4);
      tile[kr][c4] = v.x; tile[kr][c4 + 1] = v.y; tile[kr][c4 + 2] = v.z; tile[kr][c4 + 3] = v.w;
    }
  } else {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int kr = threadIdx.y + 4 * i, k = k0 + kr, c = c0 + (int)threadIdx.x;
      tile[kr][threadIdx.x] = (k < m.inner && c < m.rows) ? w[(size_t)k * m.s_inner + tap + c] : 0.f;
    }
  }
  __syncthreads();
  constexpr int EP = 16 / (int)sizeof(T);                        // elements per 16-byte piece (8 for the 16-bit types, 4 for f32)
  constexpr int PPR = 64 / EP;                                   // pieces per packed row of the tile
  if (whole) {
    for (int q = tid; q < 64 * PPR; q += 256) {
      const int cr = q / PPR, kp = (q - cr * PPR) * EP;           // packed row c0 + cr, k = k0 + kp .. + EP - 1
      T v[EP];
#pragma unroll
      for (int e = 0; e < EP; ++e) v[e] = from_f<T>(tile[kp + e][cr]);
      *reinterpret_cast<uint4*>(wp + (size_t)(c0 + cr) * KK + (size_t)tp * Cp + k0 + kp) = *reinterpret_cast<const uint4*>(v);
    }
  } else {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int c = c0 + threadIdx.y + 4 * i, k = k0 + (int)threadIdx.x;
      if (c < rows_pad && k < Cp) wp[(size_t)c * KK + (size_t)tp * Cp + k] = from_f<T>(tile[threadIdx.x][threadIdx.y + 4 * i]);
    }
  }
  if (blockIdx.x == 0 && tp == 0) {                                // zero tail of the padded K extent of these 64 rows
    for (int r = threadIdx.y; r < 64; r += 4) {
      const int c = c0 + r;
      if (c >= rows_pad) break;
      for (int kk = T_taps * Cp + threadIdx.x; kk < KK; kk += 64) wp[(size_t)c * KK + kk] = from_f<T>(0.f);
    }
  }
}

// dw[row][kk] (+)= sum_z slab[z][row][kk], kk < T_taps * C: the weight gradient of a K-major layer lands with whole rows
__global__ __launch_bounds__(256) void unpack_kmajor_kernel(const float* __restrict__ dwp, float* __restrict__ dw, int rows, int row_len,
                                                            int KK, int splits, long slab_elems, int accumulate) {
  const int r4 = row_len / 4;
  const long total4 = (long)rows * r4;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total4; e += (long)gridDim.x * 256) {
    const int row = (int)(e / r4), k4 = (int)(e - (long)row * r4) * 4;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int z = 0; z < splits; ++z) {                           // fixed order: reproducible
      const float4 v = *reinterpret_cast<const float4*>(dwp + (size_t)z * slab_elems + (size_t)row * KK + k4);
      a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    float4* o = reinterpret_cast<float4*>(dw + (size_t)row * row_len + k4);
    if (accumulate) { const float4 p = *o; a.x += p.x; a.y += p.y; a.z += p.z; a.w += p.w; }
    *o = a;
  }
}

// Packed weights of a merged sub-pixel launch (stride 2, transposed form):
//   Wp[(cls, k)][(dh, dw)][c] = w(k, c, r, s)  with  r = pi + pad - 2 dh,  s = pj + pad - 2 dw  (0 when outside the kernel)
template <typename T>
__global__ void pack_merged_kernel(GDesc d, const float* __restrict__ w, T* __restrict__ wp, int rows_pad, int K, int C, int R, int S,
                                   int pad, long s_k, long s_c) {
  const long total = (long)rows_pad * d.KK;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int row = (int)(e / d.KK);
    const int kk = (int)(e - (long)row * d.KK);
    const int t = kk / d.Cp_in, c = kk - t * d.Cp_in;
    float v = 0.f;
    if (row < 4 * d.cls_cp && t < d.nth * d.ntw && c < C) {
      const int cls = row / d.cls_cp, k = row - cls * d.cls_cp;
      const int pi = cls >> 1, pj = cls & 1;
      const int tt = d.cls_skip && pi == 1 ? ((t & 1) << 1 | (t >> 1)) : t;    // K position -> tap (GDesc::cls_skip)
      const int ta = tt / d.ntw, tb = tt - ta * d.ntw;
      const int dh = d.dh0 + ta * d.dh_step, dw = d.dw0 + tb * d.dw_step;
      const int r = pi + pad - 2 * dh, s2 = pj + pad - 2 * dw;
      if (k < K && r >= 0 && r < R && s2 >= 0 && s2 < S) v = w[k * s_k + c * s_c + r * S + s2];
    }
    wp[e] = from_f<T>(v);
  }
}

__global__ __launch_bounds__(256) void unpack_grad_kernel(GDesc d, p2phd::WMap m, const float* __restrict__ dwp,
                                                          float* __restrict__ dw, int splits, long slab_elems, int accumulate) {
  // same ownership as pack_kernel: thread = (row, channel), taps walked with counters; the split slabs are summed in
  // a fixed order (reproducible), reads are channel-contiguous, the R*S results of a pair land in one contiguous block
  const int row = blockIdx.y * 4 + threadIdx.y;
  if (row >= m.rows) return;
  const int T_taps = d.nth * d.ntw, Cp = d.Cp_in, ntw = d.ntw;
  const float* srow = dwp + (size_t)row * d.KK;
  const long roff = (long)(row % m.row_mod) * m.s_row + (long)(row / m.row_mod) * m.s_rowq;
  for (int c = blockIdx.x * 64 + threadIdx.x; c < m.inner; c += gridDim.x * 64) {
    float* dst = dw + roff + (long)(c % m.c_mod) * m.s_inner + (long)(c / m.c_mod) * m.s_innerq;
    int ta = 0, tb = 0;
    for (int t0 = 0; t0 < T_taps; t0 += 16) {                   // 16 taps at a time: that many independent loads in flight
      float v[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) v[i] = 0.f;
      for (int z = 0; z < splits; ++z) {
        const float* src = srow + (size_t)z * slab_elems + c;
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] += src[(size_t)min(t0 + i, T_taps - 1) * Cp];      // unconditional, clamped
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        if (t0 + i < T_taps) {
          float* o = dst + ((d.wr0 + ta * d.wr_step) * m.S + d.ws0 + tb * d.ws_step) * m.s_tap;
          *o = accumulate ? *o + v[i] : v[i];
          if (++tb == ntw) { tb = 0; ++ta; }
        }
      }
    }
  }
}

// ---- fp8 (OCP e4m3) weight pack of a dense direct plan: wp8[row][tap][Cp] = e4m3(w * 448 / amax), scale = amax / 448 ----
__global__ __launch_bounds__(256) void amax_kernel(const float* __restrict__ w, long n, unsigned* __restrict__ amax_bits) {
  // float4 pieces, four in flight per thread (n is a multiple of 4 and w 16-byte aligned: flat Adam buffer slices)
  float m = 0.f;
  const long n4 = n >> 2, stride = (long)gridDim.x * 256;
  const float4* w4 = reinterpret_cast<const float4*>(w);
  for (long e0 = (long)blockIdx.x * 256 + threadIdx.x; e0 < n4; e0 += 4 * stride) {
    float4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = w4[min(e0 + u * stride, n4 - 1)];
#pragma unroll
    for (int u = 0; u < 4; ++u) m = fmaxf(m, fmaxf(fmaxf(fabsf(v[u].x), fabsf(v[u].y)), fmaxf(fabsf(v[u].z), fabsf(v[u].w))));
  }
  for (long e = (n4 << 2) + (long)blockIdx.x * 256 + threadIdx.x; e < n; e += stride) m = fmaxf(m, fabsf(w[e]));
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  __shared__ float red[4];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0)
    atomicMax(amax_bits, __float_as_uint(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]))));   // non-negative floats order like their bits
}
__global__ void fp8_scale_kernel(const unsigned* __restrict__ amax_bits, float* __restrict__ scale2) {
  const float a = fmaxf(__uint_as_float(*amax_bits), 1e-30f);
  scale2[0] = a / 448.f;                                                      // de-quantisation factor (read by the conv epilogue)
  scale2[1] = 448.f / a;
}
__global__ __launch_bounds__(256) void pack_fp8_kernel(GDesc d, p2phd::WMap m, const float* __restrict__ w, unsigned char* __restrict__ wp,
                                                       int rows_pad, const float* __restrict__ scale2) {
  // thread = (row, 4 consecutive channels): reads their 4 x T_taps master values (one contiguous run, consecutive threads
  // continue it), writes one packed dword per tap (consecutive threads -> consecutive dwords of the [tap][channel] row)
  const int T_taps = d.nth * d.ntw, Cp = d.Cp_in, KK = d.KK;
  const float q = scale2[1];
  const int c4n = Cp / 4;
  const long total = (long)rows_pad * c4n;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int row = (int)(e / c4n), c = (int)(e - (long)row * c4n) * 4;
    const bool row_ok = row < m.rows;
    // element (row, channel cc, tap t) of the master tensor: PyTorch layout s_inner = T_taps, s_tap = 1; K-major s_inner = 1, s_tap = C
    const float* src = w + (row_ok ? (long)row * m.s_row : 0) + (long)min(c, max(m.inner - 4, 0)) * m.s_inner;
    for (int t0 = 0; t0 < T_taps; t0 += 4) {
      float v[4][4];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int u = 0; u < 4; ++u) v[i][u] = src[i * m.s_inner + (long)min(t0 + u, T_taps - 1) * m.s_tap];   // unconditional, clamped
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (t0 + u >= T_taps) break;
        float f[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) f[i] = (row_ok && c + i < m.inner) ? v[i][u] * q : 0.f;
        int pk = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], 0, false);
        pk = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], pk, true);
        *reinterpret_cast<int*>(wp + (size_t)row * KK + (size_t)(t0 + u) * Cp + c) = pk;
      }
    }
  }
}

// E[n, r', c', :] for the pad_mode 2 gather (see gconv_kernel): rows r' < H are dy's, r' = H holds dy[0] + dy[2],
// r' = H + 1 holds dy[H-3] + dy[H-1]; the same along W (corners: sums of sums).  H, W >= 3.
template <typename T>
__global__ void reflect_expand_kernel(const T* __restrict__ dy, T* __restrict__ e_out, int N, int H, int W, int Cp) {
  constexpr int EPP = Elem<T>::EPP;
  const int cpr = Cp / EPP;
  const int He = H + 2, We = W + 2;
  const long total = (long)N * He * We * cpr;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int pc = (int)(e % cpr);
    long r = e / cpr;
    const int j = (int)(r % We); r /= We;
    const int i = (int)(r % He);
    const int n = (int)(r / He);
    int hs[2], ws[2], nh = 1, nw = 1;
    hs[0] = i; ws[0] = j;
    if (i == H) { hs[0] = 0; hs[nh++] = 2; } else if (i == H + 1) { hs[0] = H - 3; hs[nh++] = H - 1; }
    if (j == W) { ws[0] = 0; ws[nw++] = 2; } else if (j == W + 1) { ws[0] = W - 3; ws[nw++] = W - 1; }
    uint4 ov;
    if (nh == 1 && nw == 1) {
      ov = *reinterpret_cast<const uint4*>(dy + (((size_t)n * H + hs[0]) * W + ws[0]) * Cp + pc * EPP);
    } else {
      float acc[EPP];
#pragma unroll
      for (int k = 0; k < EPP; ++k) acc[k] = 0.f;
      for (int a = 0; a < nh; ++a)
        for (int b = 0; b < nw; ++b) {
          const uint4 v = *reinterpret_cast<const uint4*>(dy + (((size_t)n * H + hs[a]) * W + ws[b]) * Cp + pc * EPP);
          const T* vv = reinterpret_cast<const T*>(&v);
#pragma unroll
          for (int k = 0; k < EPP; ++k) acc[k] += to_f(vv[k]);
        }
      T* oo = reinterpret_cast<T*>(&ov);
#pragma unroll
      for (int k = 0; k < EPP; ++k) oo[k] = from_f<T>(acc[k]);
    }
    *reinterpret_cast<uint4*>(e_out + (size_t)e * EPP) = ov;
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
    int hs[3], ws[3], nh = 1, nw = 1;                            // a row within P of BOTH borders (H <= 2 P + 1) has two mirrors
    hs[0] = i + P; ws[0] = j + P;
    if (i >= 1 && i <= P) hs[nh++] = P - i;
    if (i >= H - 1 - P && i <= H - 2) hs[nh++] = 2 * (H - 1) - i + P;
    if (j >= 1 && j <= P) ws[nw++] = P - j;
    if (j >= W - 1 - P && j <= W - 2) ws[nw++] = 2 * (W - 1) - j + P;
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

// column sums of a [P][Cp] matrix (bias gradient): db[c] (+)= sum_p x[p][c].
// Block = cpg channel pieces x R pixel rows; the rows of a block meet in LDS and are added in row order, the blocks of a
// column group store their partial row and the LAST of them (fold_arrive_last) adds the rows in block order: no float
// atomics, the same bits on every run (these are the biases with a real gradient: no InstanceNorm behind the conv).
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ x, long P, int Cp, int K, float* __restrict__ db, int cpg,
                                                     int accumulate, float* __restrict__ part, unsigned* __restrict__ tickets) {
  constexpr int EPP = Elem<T>::EPP;
  __shared__ float red[256 * 8];
  const int cpr = Cp / EPP;
  const int pl = threadIdx.x % cpg, rl = threadIdx.x / cpg, R = 256 / cpg;
  const int pc = blockIdx.y * cpg + pl;
  float acc[EPP];
#pragma unroll
  for (int k = 0; k < EPP; ++k) acc[k] = 0.f;
  if (pc < cpr) {
    // four rows per trip, all requested before the first is added (one 16-byte load in flight per thread left this pass at
    // 2.7 TB/s); rows past the end re-read the last one and are masked in the sum (no branch around a load)
    constexpr int U = 4;
    const long stride = (long)gridDim.x * R;
    for (long p0 = (long)blockIdx.x * R + rl; p0 < P; p0 += stride * U) {
      uint4 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const long p = p0 + stride * u;
        v[u] = *reinterpret_cast<const uint4*>(x + (size_t)(p < P ? p : P - 1) * Cp + pc * EPP);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const float live = p0 + stride * u < P ? 1.f : 0.f;
        const T* vv = reinterpret_cast<const T*>(&v[u]);
#pragma unroll
        for (int k = 0; k < EPP; ++k) acc[k] += to_f(vv[k]) * live;
      }
    }
  }
#pragma unroll
  for (int k = 0; k < EPP; ++k) red[threadIdx.x * 8 + k] = acc[k];
  __syncthreads();
  const int width = cpg * EPP;                                   // channels of this column group
  float* rows = part + (size_t)blockIdx.y * gridDim.x * width;
  if (rl == 0) {
#pragma unroll
    for (int k = 0; k < EPP; ++k) {
      float t = 0.f;
      for (int r = 0; r < R; ++r) t += red[(r * cpg + pl) * 8 + k];
      p2phd::fold_store(rows + (size_t)blockIdx.x * width + pl * EPP + k, t);
    }
  }
  if (!p2phd::fold_arrive_last(tickets + blockIdx.y, gridDim.x)) return;
  const int nb = (int)gridDim.x;
  for (int j = threadIdx.x; j < width; j += 256) {
    const int c = blockIdx.y * width + j;
    if (c >= K) continue;
    float s = 0.f;
    for (int b = 0; b < nb; b += 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = p2phd::fold_load(rows + (size_t)min(b + u, nb - 1) * width + j);
#pragma unroll
      for (int u = 0; u < 8; ++u) s += b + u < nb ? v[u] : 0.f;
    }
    db[c] = accumulate ? db[c] + s : s;
  }
}

// ------------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------------
// measurement hook (p2phd_probe_gconv): events around matching launches, on the launch stream
struct GconvProbe {
  bool on = false;
  int cp = 0, kk = 0, hg = 0, wg = 0;
  int pad_mode = -1, esize = 0;                  // -1 / 0 = any: tells the forward (reflect gather) from the input gradient (pad_mode 2)
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;
};
GconvProbe g_probe_cfg;

template <typename T, int BM, int BN, int MR, int NR, int NSTAGE, int HALO = 0>
int launch_gconv_cfg(const GDesc& d_in, const void* in, const void* wp, const float* bias, const void* addend, void* out,
                     float* stats, hipStream_t st, int* slot_rows) {
  GDesc d = d_in;
  // InstanceNorm partials: one slot per wave row block (MR * 32 rows) of a sample, see the epilogue
  d.stats_slots = (d.Hg * d.Wg + MR * 32 - 1) / (MR * 32);
  if (slot_rows) *slot_rows = d.bs_out != nullptr ? BM : MR * 32;   // (fused backward sums: one partial per TILE)
  constexpr int STAGE = (BM + BN) * kRowBytes;
  constexpr int CT = BM * (BN * (int)sizeof(typename OutOf<T>::type) + 16);
  const int tab = (HALO ? 0 : ((d.nth * d.ntw * BM * 4 + 15) & ~15)) + BM * 8;        // gather table + row table
  constexpr int RING = HALO ? 2 * (20 * 20 * kRowBytes) + 2 * BN * kRowBytes : NSTAGE * STAGE;   // (HALO: two halo grids + two weight slabs, gconv_halo.inc)
  const size_t lds = tab + (size_t)(RING > CT ? RING : CT);
  auto kern = gconv_kernel<T, BM, BN, MR, NR, NSTAGE, HALO>;
  if (lds > 64 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const int npix = d.Hg * d.Wg;
  const int mtiles = d.flat_m ? (int)(((long)d.N * npix + BM - 1) / BM) : ((npix + BM - 1) / BM) * d.N;
  const int ntiles = (d.n_extent + BN - 1) / BN;
  // 1-D grid over tiles (tile = n_tile * mtiles + m_tile).  Split-K tail ("stream-K" for the last round only): with one
  // workgroup per CU a grid of T tiles runs in ceil(T / CUs) rounds, and the last round of e.g. 561 or 269 tiles keeps
  // 49 / 13 CUs busy for a whole tile time.  Those tail tiles are cut along K into P = floor(CUs / tail) parts that fill
  // the round; the part that finishes last adds the partials in a fixed order (gconv_kernel).  Partials live in the
  // library's reduction scratch (common.h), so no entry point needs a bigger workspace.
  const int TT = mtiles * ntiles;
  d.grid_m = mtiles;
  d.sk_first = TT; d.sk_tail = 1; d.sk_parts = 1; d.sk_steps = 0; d.sk_part = nullptr; d.sk_ticket = nullptr;
  int wgs = TT;
  // CUs this launch can occupy: the device's count (cached per device), or what the caller states with
  // p2phd_set_option("cus", n) when the step runs on a CU-masked stream (opt.comm_cus leaves some to the RCCL kernels)
  const int cus = p2phd::g_opt_cus > 0 ? p2phd::g_opt_cus : p2phd::device_cus();
  if (p2phd::g_opt_splitk_tail != 0 && d.cls_skip == 0 && HALO == 0) {        // (tap-skipping tiles differ in depth: their order balances the rounds)
    const int nsteps = d.KK / (8 * Elem<T>::EPP);
    const int full = TT / cus * cus, tail = TT - full;
    // Cost model in microseconds (layer tables of profiles/r03_*): a K slab of a BM x BN tile at the rate one CU sustains in
    // this loop, a fixed prologue + epilogue, one partial store per part and one partial load per part by the finisher
    // (sc1 traffic of 4 BM BN bytes each).  The finisher term grows with P, so the best P is about sqrt(K time / load time):
    // deep reductions (the 256 -> 512 layers) split 4-7 ways, short ones not at all.
    const double area = (double)BM * BN / 65536.0;
    const double t_slab = (double)BM * BN * 128.0 / 6.0e6, c0 = 8.0 + 16.0 * area, c_io = 4.0 * area;
    const double t_tile = c0 + nsteps * t_slab;
    // (a sparse last round runs faster per tile than a full one -- fewer CUs on the memory system, higher clock --, which
    // is why the measured gain of filling it is smaller than a whole tile time)
    const double now = (double)(full / cus) * t_tile + (tail > 0 ? 0.75 * t_tile : 0.0);
    int bestP = 1;
    double best = now;
    const int pmax = tail > 0 ? std::min(cus / tail, nsteps / 4) : 1;
    for (int P = 2; P <= pmax; ++P) {
      const int steps = (nsteps + P - 1) / P;
      const int Pe = (nsteps + steps - 1) / steps;               // no empty parts
      const double t = (double)(full / cus) * t_tile + c0 + steps * t_slab + c_io + Pe * c_io;
      if (t < best) { best = t; bestP = Pe; }
    }
    if (p2phd::g_opt_splitk_tail == 2 && pmax >= 2) {            // tests: split as deep as allowed whatever the model says
      const int steps = (nsteps + pmax - 1) / pmax;
      bestP = (nsteps + steps - 1) / steps;
      best = 0.0;
    }
    if (bestP >= 2 && best < 0.95 * now) {
      const p2phd::FoldScratch fs = p2phd::fold_scratch(p2phd::FOLD_GCONV, st);
      const int steps = (nsteps + bestP - 1) / bestP;
      if (fs.part != nullptr && (size_t)tail * bestP * BM * BN <= fs.floats && tail <= fs.tickets) {   // (more tail tiles than tickets -- chips beyond 256 CUs -- simply run unsplit)
        d.sk_first = full; d.sk_tail = tail; d.sk_parts = bestP; d.sk_steps = steps; d.sk_part = fs.part; d.sk_ticket = fs.ticket;
        wgs = full + tail * bestP;
      }
    }
  }
#ifdef P2PHD_CHECK_WAITS
  d.cw_inject = p2phd::g_opt_cw_inject;
#endif
  dim3 grid((unsigned)wgs);
  const bool probe = g_probe_cfg.on && d.Cp_in == g_probe_cfg.cp && d.KK == g_probe_cfg.kk && d.Hg == g_probe_cfg.hg &&
                     d.Wg == g_probe_cfg.wg && (g_probe_cfg.pad_mode < 0 || d.pad_mode == g_probe_cfg.pad_mode) &&
                     (g_probe_cfg.esize == 0 || (int)sizeof(T) == g_probe_cfg.esize) && g_probe_cfg.ev.size() < 4096;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (probe) { (void)hipEventCreate(&e0); (void)hipEventCreate(&e1); (void)hipEventRecord(e0, st); }
  typedef typename OutOf<T>::type TO;
  hipLaunchKernelGGL(kern, grid, dim3((BM / (MR * 32)) * (BN / (NR * 32)) * 64), lds, st, d, (const T*)in, (const T*)wp, bias, (const TO*)addend, (TO*)out, stats);
  ++p2phd::g_launch_count[p2phd::LC_GCONV];
  if (HALO != 0) ++p2phd::g_launch_count[p2phd::LC_HALO];
  if (d.cls_skip != 0) ++p2phd::g_launch_count[p2phd::LC_CLS_SKIP];
  if (d.sk_parts > 1) ++p2phd::g_launch_count[p2phd::LC_SPLITK];
  if (BM == 256 && BN == 256) ++p2phd::g_launch_count[p2phd::LC_TILE256];
  if (BM == 128 && BN == 192) ++p2phd::g_launch_count[p2phd::LC_TILE128X192];
  if (probe) { (void)hipEventRecord(e1, st); g_probe_cfg.ev.emplace_back(e0, e1); }
  return p2phd::check_launch("gconv");
}

// The HALO main loop of the 256 x 192 tile (gconv_halo.inc): 3 x 3 taps within one pixel of the centre on a 16-wide plane whose
// height is a multiple of 16, gathered tensor of the same size, 64-channel chunks, full K rows (no padding tail)
bool gconv_halo_ok(const GDesc& d) {
  const bool taps = d.nth == 3 && d.ntw == 3 && d.sh == 1 && d.sw == 1 &&
                    ((d.dh0 == -1 && d.dh_step == 1) || (d.dh0 == 1 && d.dh_step == -1)) &&
                    ((d.dw0 == -1 && d.dw_step == 1) || (d.dw0 == 1 && d.dw_step == -1));
  return p2phd::g_opt_gconv_halo != 0 && taps && d.cls_cp == 0 && d.Wg == 16 && d.Hg % 16 == 0 && d.Hg >= 16 && d.Hin == d.Hg && d.Win == d.Wg &&
         d.Cp_in % 64 == 0 && d.KK == 9 * d.Cp_in && (d.pad_mode == 0 || d.pad_mode == 1 || d.pad_mode == 3) && d.flat_m == 0 &&
         d.oh_mul == 1 && d.ow_mul == 1 && d.oh_off == 0 && d.ow_off == 0;
}

template <typename T>
int launch_gconv_t(GDesc d, const void* in, const void* wp, const float* bias, const void* addend, void* out,
                   float* stats, hipStream_t st, int* slot_rows) {
  // N tile: the 128-wide tile has the best MFMA density (64x64 per wave) and reads the gathered A operand once;
  // narrower tiles only for layers that would leave most of it empty
  if (d.n_extent == 0) d.n_extent = d.Cp_out;
  const int k = d.n_extent;
  if constexpr (sizeof(T) == 2) {
    if (d.cls_skip != 0) {                                     // (planned for this tile: merged_plan)
      d.flat_m = 0;
      return launch_gconv_cfg<T, 256, 192, 2, 3, 2>(d, in, wp, bias, addend, out, stats, st, slot_rows);
    }
  } else if (d.cls_skip != 0) {
    p2phd::set_error("gconv: a tap-skipping merged plan reached a non-16-bit launch");
    return P2PHD_EINVAL;
  }
  const int bn = k > 64 ? 128 : (k > 32 ? 64 : 32);
  const int npix = d.Hg * d.Wg;
  const int taps = d.nth * d.ntw;
  // 256-row tiles (8 waves, 3-slot ring) halve the weight traffic per FLOP: used when a sample has enough pixels to
  // fill them, the grid still covers the chip, and ring + gather table fit the 160 KiB of LDS
  d.flat_m = stats == nullptr && d.bs_out == nullptr && (npix % 256 != 0);   // no per-sample sums wanted: tiles may straddle samples
  const long mt256 = d.flat_m ? ((long)d.N * npix + 255) / 256 : (long)((npix + 255) / 256) * d.N;
  const long tabb = (long)taps * 256 * 4 + 16 + 256 * 8;
  const long kLds = 160 * 1024;
  const bool enough_px = d.flat_m ? (long)d.N * npix >= 2048 : (npix >= 256 && (npix % 256 == 0 || npix >= 2048));
  // 256 x 256 tiles (8 waves of 128 x 64, 2-slot ring): twice the MFMA work per LDS-DMA piece; for wide layers whose
  // grid still fills most of the chip.  256 x {128,64}: 3-slot ring when it fits beside the gather table, else 2-slot.
  const bool fits_huge = sizeof(T) == 2 && 2 * 512 * kRowBytes + tabb <= kLds && 256 * (256 * 2 + 16) + tabb <= kLds;
  bool huge = fits_huge && enough_px && k >= 256 && (k % 256 == 0 || k >= 1024) && mt256 * ((k + 255) / 256) >= 160;
  if (d.bs_out != nullptr || d.as_x != nullptr) huge = false;   // (the 128-accumulator tile has no fused store loop)
  const bool fits3 = bn >= 64 && 3 * (256 + bn) * kRowBytes + tabb <= kLds && p2phd::g_opt_gconv_bm != 258;   // (258: experiments, 256 rows on the 2-slot ring)
  const bool fits2 = bn >= 64 && 2 * (256 + bn) * kRowBytes + tabb <= kLds && 256 * (bn * (long)sizeof(typename OutOf<T>::type) + 16) + tabb <= kLds;
  // short reductions (<= 4 K steps: the folded 2-channel layers, the 4-channel D input) are all prologue and epilogue:
  // keep the light 128-row kernel there, several of which fit on a CU and overlap each other's fixed costs
  const bool short_k = d.KK <= 4 * 8 * Elem<T>::EPP;
  bool big = (fits3 || fits2) && enough_px && !short_k && mt256 * ((k + bn - 1) / bn) >= 192;
  const int force = p2phd::g_opt_gconv_bm;
  if (force == 128) { big = false; huge = false; }
  if (force == 256 || force == 258) { big = fits3 || fits2; huge = false; }
  if (force == 512) { huge = fits_huge && k > 128 && d.bs_out == nullptr && d.as_x == nullptr; }
  // 256 x 192 (8 waves of 64 x 96): when the 256 x 256 grid would leave CUs idle that a 192-wide N tile fills
  // (the residual trunk: 768 = 4 x 192 -> 64 x 4 = 256 workgroups instead of 64 x 3 = 192)
  if (force == 192 && sizeof(T) <= 2 && k % 192 == 0 && 2 * 448 * kRowBytes + tabb <= kLds)
    return launch_gconv_cfg<T, 256, 192, 2, 3, 2>(d, in, wp, bias, addend, out, stats, st, slot_rows);
  // (fp8 operands: no 256 x 256 instantiation -- it spills -- so the 192-wide tile is taken whenever it divides the output)
  const bool wide = enough_px && k >= 256 && (k % 256 == 0 || k >= 1024) && mt256 * ((k + 255) / 256) >= 160;
  if ((huge || (sizeof(T) == 1 && wide)) && force == 0 && k % 192 == 0) {
    const long wg256 = mt256 * ((k + 255) / 256), wg192 = mt256 * (k / 192);
    const double c256 = std::ceil(wg256 / 256.0) * 256.0 * 256.0, c192 = std::ceil(wg192 / 256.0) * 256.0 * 192.0 / 0.95;
    if ((sizeof(T) == 1 || c192 < c256) && 2 * 448 * kRowBytes + tabb <= kLds) {
      if constexpr (sizeof(T) == 2) {
        if (gconv_halo_ok(d)) return launch_gconv_cfg<T, 256, 192, 2, 3, 2, 1>(d, in, wp, bias, addend, out, stats, st, slot_rows);
      }
      return launch_gconv_cfg<T, 256, 192, 2, 3, 2>(d, in, wp, bias, addend, out, stats, st, slot_rows);
    }
  }
  // ... and for 192- / 384-wide outputs (the 96-channel layers and the merged sub-pixel launches of the up path),
  // where 128-wide tiles would gather the A operand once more and pad the last tile
  if (!huge && force == 0 && sizeof(T) <= 2 && enough_px && !short_k && k % 192 == 0 && (k <= 384 || d.bs_out != nullptr || d.as_x != nullptr) &&
      2 * 448 * kRowBytes + tabb <= kLds && mt256 * (k / 192) >= 192)
    return launch_gconv_cfg<T, 256, 192, 2, 3, 2>(d, in, wp, bias, addend, out, stats, st, slot_rows);
  if constexpr (sizeof(T) == 2) {                              // (the 256 x 256 tile spills with the two-MFMA fp8 fragments; f32 never takes it)
    if (huge) return launch_gconv_cfg<T, 256, 256, 4, 2, 2>(d, in, wp, bias, addend, out, stats, st, slot_rows);
  }
  if (big && bn == 128) {
    if constexpr (sizeof(T) == 2) {                            // the HALO loop on 128-wide tiles: 16-wide planes whose 192- / 256-wide grids leave CUs idle
      if (force == 0 && k % 128 == 0 && !short_k && gconv_halo_ok(d))   // (configs[4]'s 2048-channel trunk at B = 8; the 768-channel trunk below B = 27)
        return launch_gconv_cfg<T, 256, 128, 2, 2, 2, 1>(d, in, wp, bias, addend, out, stats, st, slot_rows);
    }
    if (fits3) return launch_gconv_cfg<T, 256, 128, 2, 2, 3>(d, in, wp, bias, addend, out, stats, st, slot_rows);
    return launch_gconv_cfg<T, 256, 128, 2, 2, 2>(d, in, wp, bias, addend, out, stats, st, slot_rows);
  }
  if (big && bn == 64) {
    if (fits3) return launch_gconv_cfg<T, 256, 64, 2, 1, 3>(d, in, wp, bias, addend, out, stats, st, slot_rows);
    return launch_gconv_cfg<T, 256, 64, 2, 1, 2>(d, in, wp, bias, addend, out, stats, st, slot_rows);
  }
  // 128 x 192 (4 waves of 64 x 96): planes too small for 256-row tiles whose 128 x 128 grid would run a second, half-empty
  // round (the 1536-channel trunk of the two-scale generator at 16 x 8: 32 x 12 = 384 tiles -> 32 x 8 = 256)
  if constexpr (sizeof(T) == 2) {
    const long mt128 = (long)((npix + 127) / 128) * d.N;
    const long wg128 = mt128 * ((k + 127) / 128), wg192 = mt128 * (k / 192);
    if (force == 0 && !d.flat_m && !short_k && k % 192 == 0 && k >= 384 && p2phd::g_opt_tile128x192 != 0 &&
        std::ceil(wg192 / 256.0) * 192.0 < std::ceil(wg128 / 512.0) * 2.0 * 128.0 && wg192 >= 192)
      return launch_gconv_cfg<T, 128, 192, 2, 3, 2>(d, in, wp, bias, addend, out, stats, st, slot_rows);
  }
  if (bn == 128) return launch_gconv_cfg<T, 128, 128, 2, 2, 2>(d, in, wp, bias, addend, out, stats, st, slot_rows);
  if (bn == 64) return launch_gconv_cfg<T, 128, 64, 2, 1, 2>(d, in, wp, bias, addend, out, stats, st, slot_rows);
  return launch_gconv_cfg<T, 128, 32, 1, 1, 2>(d, in, wp, bias, addend, out, stats, st, slot_rows);
}

}  // namespace

namespace p2phd {

// launch_gconv_t's tile choice for a bf16 launch WITHOUT statistics / fused sums: does it take the 256 x 256 tile?  That
// tile has no fused store loop, so an input gradient carrying the producer's InstanceNorm-backward sums falls back to
// 256 x 128 tiles and loses more (D 128->256 <- 256->512 at B = 64: 735 + 40 us against 539 + 89 us for plain input
// gradient + two-pass backward) than the saved pass is worth: p2phd_conv_dgrad_bsum_pays says no for such a layer.
bool gconv_plain_launch_takes_256x256(const GDesc& d_in, int dtype) {
  if (dtype != P2PHD_BF16 || d_in.cls_skip != 0) return false;
  const int k = d_in.n_extent ? d_in.n_extent : d_in.Cp_out;
  const int npix = d_in.Hg * d_in.Wg, taps = d_in.nth * d_in.ntw;
  const bool flat = npix % 256 != 0;
  const long mt256 = flat ? ((long)d_in.N * npix + 255) / 256 : (long)((npix + 255) / 256) * d_in.N;
  const long tabb = (long)taps * 256 * 4 + 16 + 256 * 8, kLds = 160 * 1024;
  const bool enough_px = flat ? (long)d_in.N * npix >= 2048 : (npix >= 256 && (npix % 256 == 0 || npix >= 2048));
  const bool fits_huge = 2 * 512 * kRowBytes + tabb <= kLds && 256 * (256 * 2 + 16) + tabb <= kLds;
  bool huge = fits_huge && enough_px && k >= 256 && (k % 256 == 0 || k >= 1024) && mt256 * ((k + 255) / 256) >= 160;
  const int force = g_opt_gconv_bm;
  if (force == 512) huge = fits_huge && k > 128;
  else if (force != 0) huge = false;
  if (huge && force == 0 && k % 192 == 0) {
    const long wg256 = mt256 * ((k + 255) / 256), wg192 = mt256 * (k / 192);
    const double c256 = std::ceil(wg256 / 256.0) * 256.0 * 256.0, c192 = std::ceil(wg192 / 256.0) * 256.0 * 192.0 / 0.95;
    if (c192 < c256 && 2 * 448 * kRowBytes + tabb <= kLds) huge = false;
  }
  return huge;
}

}  // namespace p2phd

namespace {

// bstats[n][c] = sum over tiles (and sub-pixel classes) of the partials one input-gradient launch left (GDesc::bs_out):
// one wavefront per (sample, channel), lanes over tiles, fixed shuffle tree -> the result does not depend on timing.
// Pad channels [C, Cp) are written as zeros here (they used to cost a memset node per launch).
__global__ __launch_bounds__(256) void bsum_merge_kernel(const float* __restrict__ part, float* __restrict__ bstats, int tiles,
                                                         int n_extent, int cls_cp, int Cp, int C) {
  const int lane = threadIdx.x & 63;
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6), n = blockIdx.y;
  if (c >= Cp) return;
  float s1 = 0.f, s2 = 0.f;
  if (c < C) {
    const int ncls = cls_cp > 0 ? 4 : 1;
    for (int t = lane; t < tiles; t += 64)
      for (int q = 0; q < ncls; ++q) {
        const float2 v = *reinterpret_cast<const float2*>(part + (((size_t)n * tiles + t) * n_extent + q * cls_cp + c) * 2);
        s1 += v.x; s2 += v.y;
      }
    for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
  }
  if (lane == 0) *reinterpret_cast<float2*>(bstats + 2 * ((size_t)n * Cp + c)) = make_float2(s1, s2);
}

}  // namespace

namespace p2phd {

int launch_bsum_merge(const float* table, float* bstats, int N, long npix, int tile_rows, int n_extent, int cls_cp, int Cp, int C,
                      hipStream_t st) {
  const int tiles = (int)((npix + tile_rows - 1) / tile_rows);
  hipLaunchKernelGGL(bsum_merge_kernel, dim3((unsigned)((Cp + 3) / 4), (unsigned)N), dim3(256), 0, st, table, bstats, tiles, n_extent,
                     cls_cp, Cp, C);
  return check_launch("bsum_merge");
}

int launch_gconv(const GDesc& d_in, int dtype, const void* in, const void* wp, const float* bias, const void* addend,
                 void* out, float* stats, hipStream_t st, int* slot_rows) {
  if (d_in.N == 0 || d_in.Hg * d_in.Wg == 0) return P2PHD_OK;
  GDesc d = d_in;
  {
    const size_t esz = dtype == P2PHD_FP8_INTERNAL ? 1 : (dtype == P2PHD_BF16 ? 2 : 4);
    size_t ib = (size_t)d.N * d.Hin * d.Win * d.Cp_in * esz;
    if (d.pad_mode == 3) ib += (size_t)d.N * (2 * (d.Win + 2) + 2 * d.Hin) * d.Cp_in * esz;   // + the reflection extras behind the tensor
    const size_t wb = (size_t)round_up(d.cls_cp > 0 ? 4 * d.cls_cp : d.Cp_out, 128) * d.KK * esz;   // packed rows are padded to 128
    P2PHD_REQUIRE(ib < 0xFFFFFFF0ull && wb < 0xFFFFFFF0ull, "gconv: tensor larger than 4 GiB");
    d.in_bytes = (unsigned)ib;
    d.w_bytes = (unsigned)wb;
  }
  P2PHD_REQUIRE(d.Cp_in % 8 == 0 && d.Cp_out % 8 == 0, "gconv: channel pitch must be a multiple of 8");
  P2PHD_REQUIRE((long)d.N * d.Hin * d.Win < (1l << 31) && (long)d.N * d.Hout * d.Wout < (1l << 31), "gconv: too many pixels");
  if (dtype == P2PHD_FP8_INTERNAL) {
    P2PHD_REQUIRE(d.Cp_in % 16 == 0 && d.KK % 128 == 0 && d.out_scale != nullptr, "gconv(fp8): channel pitch %% 16, GEMM-K %% 128 and a scale are required");
    return launch_gconv_t<fp8_t>(d, in, wp, bias, addend, out, stats, st, slot_rows);
  }
  if (dtype == P2PHD_BF16) return launch_gconv_t<bf16_t>(d, in, wp, bias, addend, out, stats, st, slot_rows);
  if (dtype == P2PHD_F32) return launch_gconv_t<float>(d, in, wp, bias, addend, out, stats, st, slot_rows);
  set_error("gconv: unsupported dtype %d", dtype);
  return P2PHD_EUNSUPPORTED;
}

template <typename T, int TM>
void launch_wgrad_cfg(const GDesc& d, const void* rows, const void* gat, float* dwp, int Cp_r, int mrows, int sps, int splits,
                      long slab_elems, unsigned rows_bytes, hipStream_t st) {
  constexpr int bkp = sizeof(T) == 2 ? 64 : 32;
  constexpr int tilea = (TM == 32 && sizeof(T) == 2) ? bkp * 64 : (TM * (int)sizeof(T) / 128 > 0 ? TM * (int)sizeof(T) / 128 : 1) * bkp * 128;
  constexpr int lds = (TM == 256 ? 2 : 3) * (256 * (int)sizeof(T) / 128 * bkp * 128 + tilea);
  auto kern = wgrad_kernel<T, TM>;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  const int nx = (d.KK + 255) / 256, my = mrows / TM;
  ++p2phd::g_launch_count[p2phd::LC_WGRAD];
  hipLaunchKernelGGL(kern, dim3((unsigned)(nx * my * splits)), dim3(512), lds, st, d, (const T*)rows, (const T*)gat, dwp, Cp_r, sps, slab_elems,
                     rows_bytes, nx, my, splits, p2phd::g_opt_wgrad_xcd);
}

// Split plan of the pixel reduction: shared by the workspace query and the launch.
void wgrad_split_plan(const GDesc& d, int dtype, int M_rows, int M_rows_pad, int* tm, int* mrows, int* splits, int* sps) {
  const long P = (long)d.N * d.Hg * d.Wg;
  const int bkp = dtype == P2PHD_BF16 ? 64 : 32;
  const int total_steps = (int)std::max<long>(1, (P + bkp - 1) / bkp);
  // row tile: 32 for folded 2-channel layers, 256 (wave tile 128 x 64: twice the MFMA work per LDS-DMA piece) when
  // the output rows fill it, else 128
  *tm = M_rows <= 32 ? 32 : ((M_rows % 256 == 0 || M_rows >= 1024) ? 256 : 128);
  if (g_opt_wgrad_tm == 128 && *tm == 256) *tm = 128;
  *mrows = *tm == 32 ? 32 : round_up(M_rows_pad, *tm);
  const int tiles = (*mrows / *tm) * ((d.KK + 255) / 256);
  // Split of the pixel reduction over blockIdx.z: one 8-wave workgroup per CU, so the grid runs in
  // ceil(tiles * sp / 256) rounds of ceil(total_steps / sp) K steps; every extra split costs one more slab to write
  // and to sum.  Pick the cheapest under a 256 MiB workspace.
  const double t_step = *tm == 256 ? 0.9e-6 : (*tm == 128 ? 0.5e-6 : 0.25e-6);
  const double slab_bytes = (double)*mrows * d.KK * sizeof(float);
  int best = 1;
  double best_cost = 1e30;
  for (int sp = 1; sp <= 512; ++sp) {
    if (sp > 1 && (total_steps / sp < 4 || slab_bytes * sp > (double)(256u << 20))) break;
    const double rounds = std::ceil((double)tiles * sp / 256.0);
    const double cost = rounds * std::ceil((double)total_steps / sp) * t_step + sp * slab_bytes * 2.0 / 4.0e12 + 2e-6;
    if (cost < best_cost) { best_cost = cost; best = sp; }
  }
  *sps = (total_steps + best - 1) / best;
  *splits = (total_steps + *sps - 1) / *sps;
}

size_t wgrad_workspace_floats(const GDesc& d, int dtype, int M_rows, int M_rows_pad) {
  int tm, mrows, splits, sps;
  wgrad_split_plan(d, dtype, M_rows, M_rows_pad, &tm, &mrows, &splits, &sps);
  return (size_t)splits * mrows * d.KK;
}

int launch_wgrad(const GDesc& d_in, const WMap& m, int dtype, const void* rows, int Cp_r, int M_rows, int M_rows_pad,
                 const void* gat, float* dwp, float* dw, int accumulate, hipStream_t st) {
  // dwp: wgrad_workspace_floats() floats; dw: master-layout gradient (overwritten)
  GDesc d = d_in;
  const long P = (long)d.N * d.Hg * d.Wg;
  const size_t esz = dtype == P2PHD_BF16 ? 2 : 4;
  const size_t gb = (size_t)d.N * d.Hin * d.Win * d.Cp_in * esz, rbytes = (size_t)P * Cp_r * esz;
  P2PHD_REQUIRE(gb < 0xFFFFFFF0ull && rbytes < 0xFFFFFFF0ull, "wgrad: tensor larger than 4 GiB");
  d.in_bytes = (unsigned)gb;
  int tm, mrows, splits, sps;
  wgrad_split_plan(d, dtype, M_rows, M_rows_pad, &tm, &mrows, &splits, &sps);
  const long slab = (long)mrows * d.KK;
  if (P == 0) {
    (void)hipMemsetAsync(dwp, 0, sizeof(float) * (size_t)slab, st);
    splits = 1;
  } else if (dtype == P2PHD_BF16) {
    if (tm == 32) launch_wgrad_cfg<bf16_t, 32>(d, rows, gat, dwp, Cp_r, mrows, sps, splits, slab, (unsigned)rbytes, st);
    else if (tm == 256) launch_wgrad_cfg<bf16_t, 256>(d, rows, gat, dwp, Cp_r, mrows, sps, splits, slab, (unsigned)rbytes, st);
    else launch_wgrad_cfg<bf16_t, 128>(d, rows, gat, dwp, Cp_r, mrows, sps, splits, slab, (unsigned)rbytes, st);
  } else if (dtype == P2PHD_F32) {
    if (tm == 32) launch_wgrad_cfg<float, 32>(d, rows, gat, dwp, Cp_r, mrows, sps, splits, slab, (unsigned)rbytes, st);
    else if (tm == 256) launch_wgrad_cfg<float, 256>(d, rows, gat, dwp, Cp_r, mrows, sps, splits, slab, (unsigned)rbytes, st);
    else launch_wgrad_cfg<float, 128>(d, rows, gat, dwp, Cp_r, mrows, sps, splits, slab, (unsigned)rbytes, st);
  } else {
    set_error("wgrad: unsupported dtype %d", dtype);
    return P2PHD_EUNSUPPORTED;
  }
  if (int rc = check_launch("wgrad")) return rc;
  if (m.rows > 0 && m.inner > 0) {
    if (kmajor_dense_map(d, m)) {
      const int row_len = d.nth * d.ntw * m.inner;
      const long total4 = (long)m.rows * (row_len / 4);
      hipLaunchKernelGGL(unpack_kmajor_kernel, dim3((unsigned)std::min<long>((total4 + 255) / 256, 8192)), dim3(256), 0, st, dwp, dw, m.rows,
                         row_len, d.KK, splits, slab, accumulate);
    } else if (dense_map(d, m)) {
      const dim3 grid((unsigned)((m.inner + 63) / 64), (unsigned)((m.rows + 3) / 4));
      const int tt = d.nth * d.ntw;
      if (tt == 9) hipLaunchKernelGGL(unpack_dense_kernel<9>, grid, dim3(64, 4), 0, st, d, m, dwp, dw, splits, slab, accumulate);
      else if (tt == 16) hipLaunchKernelGGL(unpack_dense_kernel<16>, grid, dim3(64, 4), 0, st, d, m, dwp, dw, splits, slab, accumulate);
      else hipLaunchKernelGGL(unpack_dense_kernel<0>, grid, dim3(64, 4), 0, st, d, m, dwp, dw, splits, slab, accumulate);
    } else {
      const dim3 grid((unsigned)std::min((m.inner + 63) / 64, 64), (unsigned)((m.rows + 3) / 4));
      hipLaunchKernelGGL(unpack_grad_kernel, grid, dim3(64, 4), 0, st, d, m, dwp, dw, splits, slab, accumulate);
    }
  }
  return check_launch("unpack_grad");
}

int launch_pack_merged(const GDesc& d, int dtype, const float* w, void* wp, int rows_pad, int K, int C, int R, int S, int pad,
                       long s_k, long s_c, hipStream_t st) {
  const long total = (long)rows_pad * d.KK;
  const int blocks = (int)std::min<long>((total + 255) / 256, 4096);
  if (dtype == P2PHD_BF16)
    hipLaunchKernelGGL(pack_merged_kernel<bf16_t>, dim3(blocks), dim3(256), 0, st, d, w, (bf16_t*)wp, rows_pad, K, C, R, S, pad, s_k, s_c);
  else
    hipLaunchKernelGGL(pack_merged_kernel<float>, dim3(blocks), dim3(256), 0, st, d, w, (float*)wp, rows_pad, K, C, R, S, pad, s_k, s_c);
  return check_launch("pack_weights(merged)");
}

int launch_pack(const GDesc& d, const WMap& m, int dtype, const float* w, void* wp, int rows_pad, hipStream_t st) {
  if (rows_pad <= 0) return P2PHD_OK;
  if (kmajor_dense_map(d, m) && d.KK % 4 == 0) {
    const int row_len = d.nth * d.ntw * m.inner;
    const long total4 = (long)rows_pad * (d.KK / 4);
    const dim3 grid((unsigned)std::min<long>((total4 + 255) / 256, 8192));
    if (dtype == P2PHD_BF16)
      hipLaunchKernelGGL(pack_kmajor_dense_kernel<bf16_t>, grid, dim3(256), 0, st, w, (bf16_t*)wp, m.rows, rows_pad, row_len, d.KK);
    else
      hipLaunchKernelGGL(pack_kmajor_dense_kernel<float>, grid, dim3(256), 0, st, w, (float*)wp, m.rows, rows_pad, row_len, d.KK);
    return check_launch("pack_weights(k-major)");
  }
  if (kmajor_transposed_map(d, m)) {
    const dim3 grid((unsigned)((std::max(m.inner, d.Cp_in) + 63) / 64), (unsigned)((rows_pad + 63) / 64), (unsigned)(d.nth * d.ntw));
    if (dtype == P2PHD_BF16)
      hipLaunchKernelGGL(pack_kmajor_transposed_kernel<bf16_t>, grid, dim3(64, 4), 0, st, d, m, w, (bf16_t*)wp, rows_pad);
    else
      hipLaunchKernelGGL(pack_kmajor_transposed_kernel<float>, grid, dim3(64, 4), 0, st, d, m, w, (float*)wp, rows_pad);
    return check_launch("pack_weights(k-major transposed)");
  }
  if (dense_map(d, m)) {
    const dim3 dgrid((unsigned)((d.Cp_in + 63) / 64), (unsigned)((rows_pad + 3) / 4));
    if (dtype == P2PHD_BF16)
      hipLaunchKernelGGL(pack_dense_kernel<bf16_t>, dgrid, dim3(64, 4), 0, st, d, m, w, (bf16_t*)wp, rows_pad);
    else
      hipLaunchKernelGGL(pack_dense_kernel<float>, dgrid, dim3(64, 4), 0, st, d, m, w, (float*)wp, rows_pad);
    return check_launch("pack_weights(dense)");
  }
  if (transposed_map(d, m)) {
    const dim3 tgrid((unsigned)((d.Cp_in + 63) / 64), (unsigned)((rows_pad + 15) / 16));
    if (dtype == P2PHD_BF16)
      hipLaunchKernelGGL(pack_transposed_kernel<bf16_t>, tgrid, dim3(64, 4), 0, st, d, m, w, (bf16_t*)wp, rows_pad);
    else
      hipLaunchKernelGGL(pack_transposed_kernel<float>, tgrid, dim3(64, 4), 0, st, d, m, w, (float*)wp, rows_pad);
    return check_launch("pack_weights(transposed)");
  }
  const dim3 grid((unsigned)std::min((d.Cp_in + 63) / 64, 64), (unsigned)((rows_pad + 3) / 4));
  if (dtype == P2PHD_BF16)
    hipLaunchKernelGGL(pack_kernel<bf16_t>, grid, dim3(64, 4), 0, st, d, m, w, (bf16_t*)wp, rows_pad);
  else
    hipLaunchKernelGGL(pack_kernel<float>, grid, dim3(64, 4), 0, st, d, m, w, (float*)wp, rows_pad);
  return check_launch("pack_weights");
}

int launch_pack_fp8(const GDesc& d, const WMap& m, const float* w, void* wp8, int rows_pad, float* scale2, unsigned* amax_bits,
                    hipStream_t st) {
  (void)hipMemsetAsync(amax_bits, 0, sizeof(unsigned), st);
  const long n = (long)m.rows * m.s_row;
  hipLaunchKernelGGL(amax_kernel, dim3((unsigned)std::max<long>(1, std::min<long>((n / 4 + 1023) / 1024, 2048))), dim3(256), 0, st, w, n, amax_bits);
  hipLaunchKernelGGL(fp8_scale_kernel, dim3(1), dim3(1), 0, st, amax_bits, scale2);
  const long total = (long)rows_pad * (d.Cp_in / 4);
  hipLaunchKernelGGL(pack_fp8_kernel, dim3((unsigned)std::min<long>((total + 255) / 256, 8192)), dim3(256), 0, st, d, m, w,
                     (unsigned char*)wp8, rows_pad, scale2);
  return check_launch("pack_weights(fp8)");
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

int launch_reflect_expand(int dtype, const void* dy, void* e_out, int N, int H, int W, int Cp, hipStream_t st) {
  const int epp = dtype == P2PHD_BF16 ? 8 : 4;
  const long total = (long)N * (H + 2) * (W + 2) * (Cp / epp);
  const int blocks = (int)std::min<long>((total + 255) / 256, 8192);
  if (dtype == P2PHD_BF16)
    hipLaunchKernelGGL(reflect_expand_kernel<bf16_t>, dim3(blocks), dim3(256), 0, st, (const bf16_t*)dy, (bf16_t*)e_out, N, H, W, Cp);
  else
    hipLaunchKernelGGL(reflect_expand_kernel<float>, dim3(blocks), dim3(256), 0, st, (const float*)dy, (float*)e_out, N, H, W, Cp);
  return check_launch("reflect_expand");
}

int launch_colsum(int dtype, const void* x, long P, int Cp, int K, float* db, int accumulate, hipStream_t st) {
  if (P == 0) {
    if (!accumulate) (void)hipMemsetAsync(db, 0, sizeof(float) * (size_t)K, st);
    return P2PHD_OK;
  }
  const int epp = dtype == P2PHD_BF16 ? 8 : 4;
  const int cpr = Cp / epp;
  int cpg = 1;
  while (cpg * 2 <= cpr && cpg * 2 <= 64) cpg *= 2;
  const int R = 256 / cpg;
  const int ygroups = (cpr + cpg - 1) / cpg;
  const FoldScratch fs = fold_scratch(FOLD_COLSUM, st);
  if (fs.part == nullptr) return P2PHD_EINVAL;                   // (refused: error text set by fold_scratch)
  P2PHD_REQUIRE(ygroups <= fs.tickets, "colsum: too many channels (%d)", Cp);
  const long rows_max = (long)(fs.floats / ((size_t)ygroups * cpg * epp));   // partial rows per column group
  P2PHD_REQUIRE(rows_max >= 1, "colsum: too many channels for the reduction scratch");
  const int xblocks = (int)std::max<long>(1, std::min<long>(std::min<long>((P + R * 32 - 1) / (R * 32), 512), rows_max));
  dim3 grid(xblocks, ygroups);
  if (dtype == P2PHD_BF16)
    hipLaunchKernelGGL(colsum_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)x, P, Cp, K, db, cpg, accumulate, fs.part, fs.ticket);
  else
    hipLaunchKernelGGL(colsum_kernel<float>, grid, dim3(256), 0, st, (const float*)x, P, Cp, K, db, cpg, accumulate, fs.part, fs.ticket);
  return check_launch("colsum");
}

}  // namespace p2phd

// Wait checker (-DP2PHD_CHECK_WAITS build, libp2phd_hip_chk.so): out[0] = bit mask of the kernel families (1 gather-GEMM generic
// loop, 2 HALO loop, 4 weight gradient, 8 its f32 form) in which a relaxed s_waitcnt vmcnt(n) left a piece in flight that
// targets a buffer read behind the following barrier, out[1] = relaxed waits checked, out[2] = first offender
// (family << 16 | n << 8 | buffer tag), out[3] = LDS-DMA pieces logged.  Synchronises the device.  Returns P2PHD_EUNSUPPORTED in
// the product build (which carries no instrumentation).
extern "C" int p2phd_wait_check(unsigned* out4, int reset) {
#ifdef P2PHD_CHECK_WAITS
  P2PHD_REQUIRE(out4 != nullptr, "wait_check: null pointer");
  if (hipDeviceSynchronize() != hipSuccess || hipMemcpyFromSymbol(out4, HIP_SYMBOL(g_cw_flag), sizeof(unsigned) * 4) != hipSuccess) {
    p2phd::set_error("wait_check: cannot read the device flag");
    return P2PHD_ELAUNCH;
  }
  if (reset) {
    const unsigned z[4] = {0, 0, 0, 0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_cw_flag), z, sizeof(z)) != hipSuccess) { p2phd::set_error("wait_check: reset failed"); return P2PHD_ELAUNCH; }
  }
  return P2PHD_OK;
#else
  (void)out4; (void)reset;
  p2phd::set_error("wait_check: this library was built without -DP2PHD_CHECK_WAITS (load libp2phd_hip_chk.so)");
  return P2PHD_EUNSUPPORTED;
#endif
}

extern "C" int p2phd_probe_gconv_ex(int enable, int cin_pitch, int kk, int hg, int wg, int pad_mode, int elem_bytes) {
  for (auto& e : g_probe_cfg.ev) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
  g_probe_cfg.ev.clear();
  g_probe_cfg.on = enable != 0;
  g_probe_cfg.cp = cin_pitch; g_probe_cfg.kk = kk; g_probe_cfg.hg = hg; g_probe_cfg.wg = wg;
  g_probe_cfg.pad_mode = pad_mode; g_probe_cfg.esize = elem_bytes;
  return P2PHD_OK;
}

extern "C" int p2phd_probe_gconv(int enable, int cin_pitch, int kk, int hg, int wg) {
  return p2phd_probe_gconv_ex(enable, cin_pitch, kk, hg, wg, -1, 0);
}

extern "C" int p2phd_probe_read(float* ms_out, int cap) {
  int n = 0;
  for (auto& e : g_probe_cfg.ev) {
    if (n >= cap) break;
    if (hipEventSynchronize(e.second) != hipSuccess) break;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, e.first, e.second) != hipSuccess) break;
    if (ms_out) ms_out[n] = ms;
    ++n;
  }
  return n;
}

#ifdef P2PHD_PROBE
extern "C" int p2phd_debug_probe(unsigned long long* out8, int reset) {
  static unsigned long long host[kProbeSlots * 8];
  if (hipMemcpyFromSymbol(host, HIP_SYMBOL(g_probe), sizeof(host)) != hipSuccess) return -1;
  for (int k = 0; k < 8; ++k) out8[k] = 0;
  for (int i = 0; i < kProbeSlots; ++i)
    for (int k = 0; k < 8; ++k) out8[k] += host[(size_t)i * 8 + k];
  if (reset) {
    void* dp = nullptr;
    if (hipGetSymbolAddress(&dp, HIP_SYMBOL(g_probe)) != hipSuccess || hipMemset(dp, 0, sizeof(host)) != hipSuccess) return -1;
  }
  return 0;
}
#endif
