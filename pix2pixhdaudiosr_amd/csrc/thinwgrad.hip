// Weight gradient of convolutions with a THIN side (<= 4 channels in or out) at full resolution, bf16:
//   * Conv2d(2, ngf, 7) behind ReflectionPad2d(3)  (models/networks.py:190)    thin = x,  wide = dy
//   * Conv2d(ngf, 2, 7) behind ReflectionPad2d(3)  (:207)                      thin = dy, wide = x (reflect-gathered)
//   * Conv2d(4, ndf, 4, stride 2, padding 2)        (:343, discriminator input) thin = x,  wide = dy
// dW[wide ch][tap][thin ch] = sum over pixels q of wide[q][ch] * thin[q * stride + tap - pad][c]: a GEMM with M = wide
// channels, N = taps x thin channels (98 / 64 columns), reduction over 0.5-4 M pixels, 39 GFLOP and 0.47 GB per launch:
// HBM-bound.  The generic panel kernel (conv.hip) reached these layers through a materialised W-fold image at 1 TB/s.
//
// Here a workgroup marches over tiles of 8 grid rows: the thin tensor's halo of the tile is read once into LDS in
// compact form; per step of 64 pixels (one row segment) the wide rows are copied global -> LDS (full 16-byte pieces, the
// only real HBM stream), the im2col block [64 pixels][N columns] of the thin side is BUILT IN LDS from the halo, and the
// four waves run v_mfma_f32_32x32x16_bf16 with both operands fetched by the transposing ds_read_b64_tr_b16 (reduction
// index = LDS row), exactly the fragment scheme of wgrad_kernel.  Accumulators stay in registers over the whole march;
// every workgroup writes one [M][N] slab, a second kernel adds the slabs in a fixed order into the master layout.
#include "convplan.h"

namespace {

typedef p2phd_h16 bf16_t;                 // the library's 16-bit storage type: bf16, or fp16 in the -DP2PHD_F16 build (common.h)
typedef __attribute__((ext_vector_type(8))) bf16_t bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int RT = 8;                      // grid rows per tile
constexpr int SEG = 64;                    // pixels per step
constexpr int PANEL = SEG * 128;           // one LDS panel: 64 pixel rows x 128 B (64 bf16 columns)

struct ThinDesc {
  int N, Hg, Wg;                 // pixel grid of the wide tensor (rows of the reduction)
  int Hw, Ww, Cpw;               // wide tensor storage [N][Hw][Ww][Cpw]; grid pixel (h, w) reads storage pixel (map(h - woff), map(w - woff))
  int woff, wide_reflect;        // woff = 3 and reflect for the 48 -> 2 layer (grid = padded domain), else 0
  int Ht, Wt;                    // thin tensor storage [N][Ht][Wt][8]
  int stride, pad, thin_reflect; // thin position = grid * stride + tap - pad; reflect or zero outside
  int M;                         // wide channels (<= 128)
};

__device__ __forceinline__ int reflect_idx(int i, int n) {
  if (i < 0) i = -i;
  if (i >= n) i = 2 * (n - 1) - i;
  return i;
}

// byte offset of element (row, col) inside a panel image [64 rows][128 B] whose 16-byte chunks are XOR-swizzled for the
// transposing reads (rows 2,3 mod 4 use the other half of the row; see wgrad_kernel)
__device__ __forceinline__ int panel_off(int row, int col) {
  const int chunk = (col & 63) >> 3;
  return (col >> 6) * PANEL + row * 128 + ((chunk ^ (((row >> 1) & 1) << 2)) << 4) + (col & 7) * 2;
}

#define THIN_TR_READ(dst, addr, OFF) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF))

// KS x KS taps, C2 thin channels (2 or 4), MB row blocks of 32 wide channels, NBLK column blocks of 32 (N = KS*KS*C2 <= 32*NBLK)
template <int KS, int C2, int MB, int NBLK>
__global__ __launch_bounds__(256) void thin_wgrad_kernel(const ThinDesc d, const bf16_t* __restrict__ wide,
                                                         const bf16_t* __restrict__ thin, float* __restrict__ slabs) {
  constexpr int NCOL = KS * KS * C2;
  constexpr int NPANEL_B = (32 * NBLK + 63) / 64;
  constexpr int NPANEL_A = (32 * MB + 63) / 64;
  static_assert(NCOL <= 32 * NBLK && (NBLK == 2 || NBLK == 4), "column blocks");
  constexpr int CW = NBLK >= 4 ? NBLK / 4 : 1;                 // column blocks per wave (4 waves side by side)
  constexpr int TPX = C2 * 2;                                  // bytes per thin pixel in the compact halo
  extern __shared__ float4 smem_raw[];
  char* smem = reinterpret_cast<char*>(smem_raw);
  const int halo_h = (RT - 1) * d.stride + KS, halo_w = (SEG - 1) * d.stride + KS;
  const int halo_pitch = (halo_w * TPX + 15) & ~15;            // bytes
  char* s_a = smem;                                            // wide tile: NPANEL_A panels
  char* s_b = s_a + NPANEL_A * PANEL;                          // im2col tile: NPANEL_B panels
  char* s_h = s_b + NPANEL_B * PANEL;                          // thin halo: halo_h x halo_pitch

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int segs = (d.Wg + SEG - 1) / SEG, rtiles = (d.Hg + RT - 1) / RT;
  const long ntiles = (long)d.N * rtiles * segs;

  f32x16 acc[MB][CW];
#pragma unroll
  for (int i = 0; i < MB; ++i)
#pragma unroll
    for (int j = 0; j < CW; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // zero the im2col and wide panels once: columns / channels beyond the real extent must read as zeros forever
  for (int e = tid; e < (NPANEL_A + NPANEL_B) * PANEL / 16; e += 256) reinterpret_cast<float4*>(s_a)[e] = make_float4(0.f, 0.f, 0.f, 0.f);

  // fragment addresses (see wgrad_kernel): 16-lane group g16 reads a 4-pixel x 16-column block
  const int g16 = lane >> 4, i16 = lane & 15, q4 = i16 >> 2, pq = i16 & 3, hh = g16 >> 1;
  const int u8 = 4 * (g16 & 1) + pq;
  const int swzq = ((q4 >> 1) & 1) << 2;
  const unsigned sa_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)s_a;
  const unsigned sb_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)s_b;
  unsigned ta_off[MB], tb_off[CW];
#pragma unroll
  for (int i = 0; i < MB; ++i) {
    const int cb = 32 * i;
    ta_off[i] = sa_base + (unsigned)((cb >> 6) * PANEL + (8 * hh + q4) * 128 + (((((cb & 63) >> 3) + (u8 >> 1)) ^ swzq) << 4) + 8 * (u8 & 1));
  }
#pragma unroll
  for (int j = 0; j < CW; ++j) {
    const int cb = 32 * (wave * CW + j);
    tb_off[j] = sb_base + (unsigned)((cb >> 6) * PANEL + (8 * hh + q4) * 128 + (((((cb & 63) >> 3) + (u8 >> 1)) ^ swzq) << 4) + 8 * (u8 & 1));
  }
  const bool wave_live = NBLK >= 4 || wave < NBLK;             // NBLK = 2: waves 2,3 only help with the copies

  // The march is a flat sequence of steps L = (tile k of this workgroup, row rr of the tile).  The wide rows of a step are
  // the only real HBM stream; they are fetched TWO steps ahead into registers (MB 16-byte pieces per thread and step)
  // so that with 4 workgroups per CU ~50 KB are in flight per CU, which covers the memory latency at the HBM rate.
  const long my_tiles = blockIdx.x < ntiles ? (ntiles - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
  const long nsteps = my_tiles * RT;
  const int cpr = d.Cpw / 8;                                   // 16-byte pieces per wide pixel
  // this thread's pieces of a wide segment and its elements of the im2col block never change: coordinates and LDS
  // offsets are computed once (the integer divisions here cost more than a step's MFMAs when done per step)
  int wpx[MB], wpc[MB], wdst[MB];
#pragma unroll
  for (int i = 0; i < MB; ++i) {
    const int e = tid + 256 * i;
    wpx[i] = e / cpr; wpc[i] = e - wpx[i] * cpr;
    wdst[i] = (wpx[i] < SEG && wpc[i] * 8 < 64 * NPANEL_A) ? panel_off(wpx[i], wpc[i] * 8) : -1;
  }
  constexpr int NIM = (SEG * KS * KS + 255) / 256;
  int isrc[NIM], idst[NIM];
#pragma unroll
  for (int j = 0; j < NIM; ++j) {
    const int e = tid + 256 * j;
    const int px = e / (KS * KS), t = e - px * (KS * KS);
    const int dh = t / KS, dw = t - dh * KS;
    isrc[j] = dh * halo_pitch + (px * d.stride + dw) * TPX;
    idst[j] = e < SEG * KS * KS ? panel_off(px, t * C2) : -1;
  }
  auto fetch = [&](long L, uint4 (&v)[MB]) {
    const long tile = blockIdx.x + (L / RT) * gridDim.x;
    const int rr = (int)(L % RT);
    const int sg = (int)(tile % segs);
    const long t2 = tile / segs;
    const int h = (int)(t2 % rtiles) * RT + rr, n = (int)(t2 / rtiles);
    int hs = h - d.woff;
    if (d.wide_reflect) hs = reflect_idx(hs, d.Hw);
    const bool row_ok = L < nsteps && h < d.Hg;
#pragma unroll
    for (int i = 0; i < MB; ++i) {
      const int px = wpx[i], pc = wpc[i];
      const int w = sg * SEG + px;
      int ws = w - d.woff;
      if (d.wide_reflect) ws = reflect_idx(ws, d.Ww);
      const bool ok = row_ok && px < SEG && w < d.Wg && pc * 8 < d.M;
      // unconditional clamped load, masked afterwards (a load under a branch waits for itself)
      const uint4 t = *reinterpret_cast<const uint4*>(wide + (ok ? (((size_t)n * d.Hw + hs) * d.Ww + ws) * d.Cpw + pc * 8 : 0));
      v[i] = ok ? t : make_uint4(0u, 0u, 0u, 0u);
    }
  };
  auto commit = [&](const uint4 (&v)[MB]) {
#pragma unroll
    for (int i = 0; i < MB; ++i)
      if (wdst[i] >= 0) *reinterpret_cast<uint4*>(s_a + wdst[i]) = v[i];
  };
  uint4 p0[MB], p1[MB];
  fetch(0, p0);
  fetch(1, p1);
  for (long L = 0; L < nsteps; ++L) {
    const long tile = blockIdx.x + (L / RT) * gridDim.x;
    const int rr = (int)(L % RT);
    const int sg = (int)(tile % segs);
    const long t2 = tile / segs;
    const int rt = (int)(t2 % rtiles), n = (int)(t2 / rtiles);
    const int h0 = rt * RT, w0 = sg * SEG;
    __syncthreads();                                           // the previous step's MFMAs are done with the panels / halo
    if (rr == 0) {
      // ---- thin halo of the tile, compact (C2 channels), padding applied here
      const unsigned* tin = reinterpret_cast<const unsigned*>(thin) + (size_t)n * d.Ht * d.Wt * 4;   // 4 dwords per stored pixel
      const int hb = h0 * d.stride - d.pad, wb = w0 * d.stride - d.pad;
      for (int e0 = tid; e0 < halo_h * halo_w; e0 += 256 * 4) {
        unsigned lo[4], hi2[4];
        bool okv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {                          // four independent loads in flight per thread
          const int e = min(e0 + 256 * u, halo_h * halo_w - 1);
          const int r = e / halo_w, c = e - r * halo_w;
          int hi = hb + r, wi = wb + c;
          if (d.thin_reflect) { hi = reflect_idx(hi, d.Ht); wi = reflect_idx(wi, d.Wt); }
          okv[u] = hi >= 0 && hi < d.Ht && wi >= 0 && wi < d.Wt;
          const unsigned* src = tin + ((size_t)(okv[u] ? hi : 0) * d.Wt + (okv[u] ? wi : 0)) * 4;
          lo[u] = src[0];
          hi2[u] = C2 == 4 ? src[1] : 0u;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int e = e0 + 256 * u;
          if (e < halo_h * halo_w) {
            const int r = e / halo_w, c = e - r * halo_w;
            if constexpr (C2 == 2) *reinterpret_cast<unsigned*>(s_h + r * halo_pitch + c * 4) = okv[u] ? lo[u] : 0u;
            else *reinterpret_cast<uint2*>(s_h + r * halo_pitch + c * 8) = okv[u] ? make_uint2(lo[u], hi2[u]) : make_uint2(0u, 0u);
          }
        }
      }
      __syncthreads();
    }
    commit(p0);                                                // wide rows of this step (fetched two steps ago)
#pragma unroll
    for (int i = 0; i < MB; ++i) p0[i] = p1[i];
    fetch(L + 2, p1);
    // ---- im2col block of the thin side: column (dh * KS + dw) * C2 + c of pixel px = halo(rr * s + dh, px * s + dw)[c]
    {
      const char* hrow = s_h + rr * d.stride * halo_pitch;
      if constexpr (C2 == 2) {
        unsigned v[NIM];
#pragma unroll
        for (int j = 0; j < NIM; ++j) v[j] = *reinterpret_cast<const unsigned*>(hrow + (idst[j] >= 0 ? isrc[j] : 0));
#pragma unroll
        for (int j = 0; j < NIM; ++j)
          if (idst[j] >= 0) *reinterpret_cast<unsigned*>(s_b + idst[j]) = v[j];
      } else {
        uint2 v[NIM];
#pragma unroll
        for (int j = 0; j < NIM; ++j) v[j] = *reinterpret_cast<const uint2*>(hrow + (idst[j] >= 0 ? isrc[j] : 0));
#pragma unroll
        for (int j = 0; j < NIM; ++j)
          if (idst[j] >= 0) *reinterpret_cast<uint2*>(s_b + idst[j]) = v[j];
      }
    }
    __syncthreads();
    if (wave_live) {
#pragma unroll
      for (int sub = 0; sub < SEG / 16; ++sub) {
        uint2 af[MB][2], bf[CW][2];
#pragma unroll
        for (int i = 0; i < MB; ++i) {
          switch (sub) {
            case 0: THIN_TR_READ(af[i][0], ta_off[i], 0); THIN_TR_READ(af[i][1], ta_off[i], 512); break;
            case 1: THIN_TR_READ(af[i][0], ta_off[i], 2048); THIN_TR_READ(af[i][1], ta_off[i], 2560); break;
            case 2: THIN_TR_READ(af[i][0], ta_off[i], 4096); THIN_TR_READ(af[i][1], ta_off[i], 4608); break;
            default: THIN_TR_READ(af[i][0], ta_off[i], 6144); THIN_TR_READ(af[i][1], ta_off[i], 6656); break;
          }
        }
#pragma unroll
        for (int j = 0; j < CW; ++j) {
          switch (sub) {
            case 0: THIN_TR_READ(bf[j][0], tb_off[j], 0); THIN_TR_READ(bf[j][1], tb_off[j], 512); break;
            case 1: THIN_TR_READ(bf[j][0], tb_off[j], 2048); THIN_TR_READ(bf[j][1], tb_off[j], 2560); break;
            case 2: THIN_TR_READ(bf[j][0], tb_off[j], 4096); THIN_TR_READ(bf[j][1], tb_off[j], 4608); break;
            default: THIN_TR_READ(bf[j][0], tb_off[j], 6144); THIN_TR_READ(bf[j][1], tb_off[j], 6656); break;
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < MB; ++i)
#pragma unroll
          for (int j = 0; j < CW; ++j) {
            bf16x8 a8, b8;
            uint2* ap = reinterpret_cast<uint2*>(&a8);
            uint2* bp = reinterpret_cast<uint2*>(&b8);
            ap[0] = af[i][0]; ap[1] = af[i][1];
            bp[0] = bf[j][0]; bp[1] = bf[j][1];
            acc[i][j] = p2phd_mfma_32x32x16(a8, b8, acc[i][j]);
          }
      }
    }
  }

  // ---- slab of this workgroup: [32 MB rows][32 NBLK columns] f32
  if (wave_live) {
    float* slab = slabs + (size_t)blockIdx.x * (32 * MB) * (32 * NBLK);
    const int lr = lane & 31, lh = lane >> 5;
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
      for (int j = 0; j < CW; ++j) {
        const int col = 32 * (wave * CW + j) + lr;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int row = 32 * i + (e & 3) + 8 * (e >> 2) + 4 * lh;
          slab[(size_t)row * (32 * NBLK) + col] = acc[i][j][e];
        }
      }
  }
}

// dw[...] (+)= sum over slabs; (wide channel m, tap t, thin channel c) -> master offset m * s_m + c * s_c + tap index (flipped if
// asked).  Workgroup = 64 consecutive slab elements x 16 slab groups; every thread adds its group's slabs with eight loads
// in flight, the groups meet in LDS: fixed order, reproducible.
__global__ __launch_bounds__(1024) void thin_wgrad_reduce_kernel(const float* __restrict__ slabs, int nslabs, int rows, int cols,
                                                                 int M, int KS, int C2, long s_m, long s_c, int flip, int accumulate,
                                                                 float* __restrict__ dw) {
  __shared__ float red[16][64];
  const int el = blockIdx.x * 64 + (threadIdx.x & 63), grp = threadIdx.x >> 6;
  const size_t stride = (size_t)rows * cols;
  const int per = (nslabs + 15) / 16;
  const int s0 = grp * per, s1 = min(nslabs, s0 + per);
  float a[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) a[k] = 0.f;
  if (el < rows * cols) {
    const float* src = slabs + (size_t)s0 * stride + el;
    int s = s0;
    for (; s + 8 <= s1; s += 8) {
#pragma unroll
      for (int k = 0; k < 8; ++k) a[k] += src[(size_t)k * stride];
      src += 8 * stride;
    }
    for (; s < s1; ++s) { a[0] += *src; src += stride; }
  }
  red[grp][threadIdx.x & 63] = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
  __syncthreads();
  if (grp == 0 && el < rows * cols) {
    float v = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) v += red[k][threadIdx.x];
    const int m = el / cols, col = el - m * cols;
    const int ncol = KS * KS * C2;
    if (m < M && col < ncol) {
      const int t = col / C2, c = col - t * C2;
      const int tt = flip ? KS * KS - 1 - t : t;
      float* o = dw + m * s_m + c * s_c + tt;
      *o = accumulate ? *o + v : v;
    }
  }
}

constexpr int kThinWGs = 1024;

template <int KS, int C2, int MB, int NBLK>
int launch_thin(const ThinDesc& d, const void* wide, const void* thin, float* slabs, hipStream_t st) {
  const int halo_h = (RT - 1) * d.stride + KS, halo_w = (SEG - 1) * d.stride + KS;
  const int halo_pitch = (halo_w * C2 * 2 + 15) & ~15;
  const size_t lds = (size_t)(((32 * NBLK + 63) / 64) + ((32 * MB + 63) / 64)) * PANEL + (size_t)halo_h * halo_pitch;
  auto kern = thin_wgrad_kernel<KS, C2, MB, NBLK>;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(kern, dim3(kThinWGs), dim3(256), lds, st, d, (const bf16_t*)wide, (const bf16_t*)thin, slabs);
  return p2phd::check_launch("thin_wgrad");
}

}  // namespace

namespace p2phd {

// which layers take this kernel: bf16, square 7x7 s1 p3 reflect with 2 channels on one side, or 4x4 s2 p2 zero with 4 in
int thin_wgrad_kind(const p2phd_conv_desc* c) {
  if (g_opt_c7_generic || c->dtype != P2PHD_BF16 || c->transposed) return 0;
  if (c->R == 7 && c->S == 7 && c->stride == 1 && c->pad == 3 && c->pad_mode == 1) {
    if (c->C == 2 && c->K >= 8 && c->K <= 128) return 1;          // 2 -> K
    if (c->K == 2 && c->C >= 8 && c->C <= 128) return 2;          // C -> 2
  }
  if (c->R == 4 && c->S == 4 && c->stride == 2 && c->pad == 2 && c->pad_mode == 0 && c->C == 4 && c->K >= 8 && c->K <= 128) return 3;
  return 0;
}

size_t thin_wgrad_workspace_floats(const p2phd_conv_desc* c) {
  const int kind = thin_wgrad_kind(c);
  if (!kind) return 0;
  const int M = kind == 2 ? c->C : c->K;
  const int mb = (M + 31) / 32, nblk = kind == 3 ? 2 : 4;
  return (size_t)kThinWGs * 32 * mb * 32 * nblk;
}

int thin_wgrad(const p2phd_conv_desc* c, const void* x, const void* dy, float* dw, int accumulate, float* slabs, hipStream_t st) {
  const int kind = thin_wgrad_kind(c);
  int Ho = (c->H + 2 * c->pad - c->R) / c->stride + 1, Wo = (c->W + 2 * c->pad - c->S) / c->stride + 1;
  ThinDesc d{};
  d.N = c->N;
  const void *wide, *thin;
  int M, KS, C2;
  long s_m, s_c;
  int flip = 0;
  if (kind == 1 || kind == 3) {            // thin = x (taps reach into it), wide = dy on the output grid
    d.Hg = Ho; d.Wg = Wo; d.Hw = Ho; d.Ww = Wo; d.Cpw = cpitch(c->K); d.woff = 0; d.wide_reflect = 0;
    d.Ht = c->H; d.Wt = c->W; d.stride = c->stride; d.pad = c->pad; d.thin_reflect = c->pad_mode;
    wide = dy; thin = x; M = c->K; KS = c->R; C2 = c->C;
    s_m = (long)c->C * c->R * c->S; s_c = (long)c->R * c->S;          // dw[k][c][r][s]
  } else {                                 // thin = dy (zero outside), wide = reflect-padded x on the padded grid
    d.Hg = c->H + 6; d.Wg = c->W + 6; d.Hw = c->H; d.Ww = c->W; d.Cpw = cpitch(c->C); d.woff = 3; d.wide_reflect = 1;
    d.Ht = Ho; d.Wt = Wo; d.stride = 1; d.pad = 6; d.thin_reflect = 0;
    // dW[n][c][dh][dw] = sum_q xpad[q][c] dy[q - (dh, dw)][n]: thin position q + t' - 6 with t' = 6 - dh: flipped taps
    wide = x; thin = dy; M = c->C; KS = 7; C2 = 2; flip = 1;
    s_m = (long)c->R * c->S; s_c = (long)c->C * c->R * c->S;          // dw[n][c][r][s]: m = c, thin channel = n
  }
  d.M = M;
  const int mb = (M + 31) / 32;
  const int nblk = kind == 3 ? 2 : 4;
  int rc = P2PHD_EUNSUPPORTED;
#define P2PHD_THIN(KSV, C2V, MBV, NBV) rc = launch_thin<KSV, C2V, MBV, NBV>(d, wide, thin, slabs, st)
  if (kind == 3) {
    switch (mb) { case 1: P2PHD_THIN(4, 4, 1, 2); break; case 2: P2PHD_THIN(4, 4, 2, 2); break; case 3: P2PHD_THIN(4, 4, 3, 2); break; default: P2PHD_THIN(4, 4, 4, 2); }
  } else {
    switch (mb) { case 1: P2PHD_THIN(7, 2, 1, 4); break; case 2: P2PHD_THIN(7, 2, 2, 4); break; case 3: P2PHD_THIN(7, 2, 3, 4); break; default: P2PHD_THIN(7, 2, 4, 4); }
  }
#undef P2PHD_THIN
  if (rc) return rc;
  const int total = 32 * mb * 32 * nblk;
  hipLaunchKernelGGL(thin_wgrad_reduce_kernel, dim3((total + 63) / 64), dim3(1024), 0, st, slabs, kThinWGs, 32 * mb, 32 * nblk, M, KS, C2,
                     s_m, s_c, flip, accumulate, dw);
  return check_launch("thin_wgrad_reduce");
}

}  // namespace p2phd
