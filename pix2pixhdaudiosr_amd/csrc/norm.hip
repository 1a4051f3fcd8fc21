// HBM-bound companions of the conv kernels: InstanceNorm2d(affine=False) + activation (+ residual) forward
// and backward, activation backward, AvgPool2d(3, 2, 1, count_include_pad=False) forward/backward and the
// NCHW-f32 <-> NHWC layout converters at the module boundary.  All move 16-byte pieces along the channel
// (innermost NHWC) axis so every wave instruction touches whole 1 KiB / 512 B runs.
//
// Reference semantics: models/networks.py:22 (InstanceNorm2d, eps 1e-5, biased variance, no affine, no running
// stats), :188,233 (ReLU), :342-358 (LeakyReLU 0.2), :165,308 (AvgPool2d), :252 (residual add).
#include "common.h"
namespace p2phd { extern int g_opt_wgrad_xcd; }   // 1 (default): XCD-aware workgroup order (core.hip)

namespace {

using p2phd::fold_store;
using p2phd::fold_load;
typedef p2phd_h16 bf16_t;                 // the library's 16-bit storage type: bf16, or fp16 in the -DP2PHD_F16 build (common.h)
template <typename T> struct Elem;
template <> struct Elem<float> { static constexpr int EPP = 4; };
template <> struct Elem<bf16_t> { static constexpr int EPP = 8; };
__device__ __forceinline__ float to_f(float v) { return v; }
__device__ __forceinline__ float to_f(bf16_t v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f(float v);
template <> __device__ __forceinline__ float from_f<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f<bf16_t>(float v) { return (bf16_t)v; }

// The activation code is folded into ONE float per kernel (slope applied to negative pre-activations: 0 ReLU, 0.2
// LeakyReLU, 1 none), so the per-element work is a compare + select instead of a chain of uniform branches on `act`
// (2 branches per element and pass showed up as issue stalls in these otherwise load-bound kernels).
__device__ __forceinline__ float neg_slope_of(int act) {
  return act == P2PHD_ACT_RELU ? 0.f : (act == P2PHD_ACT_LRELU ? 0.2f : 1.f);
}
__device__ __forceinline__ float act_fwd(float v, float neg_slope) { return v > 0.f ? v : neg_slope * v; }
__device__ __forceinline__ float act_slope(float v, float neg_slope) { return v > 0.f ? 1.f : neg_slope; }   // at pre-activation v

constexpr int kMaxCp = 4096;   // bounds the LDS partial-sum tables of the backward kernels

// All three InstanceNorm kernels walk the [HW][Cp] plane of one sample in 16-byte pieces with a grid stride that the
// host makes a multiple of the pieces per pixel (cpr = Cp / EPP).  A thread therefore stays on ONE group of EPP
// channels for its whole life: mean / rstd (and the backward means) of those channels are computed once into
// registers straight from the statistics buffers -- no per-block table, no LDS in the streaming loop -- and the loads
// of UN pieces are issued before any of them is used.
template <typename T>
struct ChanConsts {
  static constexpr int EPP = Elem<T>::EPP;
  float mean[EPP], rstd[EPP];
  __device__ __forceinline__ void load(const float* __restrict__ stats, int n, int Cp, int C, int pc, float inv, float eps) {
    const float4* s4 = reinterpret_cast<const float4*>(stats + 2 * ((size_t)n * Cp + pc * EPP));
#pragma unroll
    for (int k = 0; k < EPP; k += 2) {
      const float4 v = s4[k >> 1];                               // (mean, sum of squared deviations) of two channels
      mean[k] = v.x; mean[k + 1] = v.z;
      rstd[k] = (pc * EPP + k < C) ? rsqrtf(fmaxf(v.y * inv, 0.f) + eps) : 0.f;
      rstd[k + 1] = (pc * EPP + k + 1 < C) ? rsqrtf(fmaxf(v.w * inv, 0.f) + eps) : 0.f;
    }
  }
};

// ---- forward: out = act((y - mean) * rstd) + residual --------------------------------------------
// out8 (optional, bf16 instantiation only): the same values as OCP e4m3 (scale 1, saturated at +-448), the operand of the
// next layer's fp8 forward conv (p2phd_conv_fwd_fp8)
template <typename T>
__global__ __launch_bounds__(256) void in_act_fwd_kernel(const T* __restrict__ y, const float* __restrict__ stats,
                                                         const T* __restrict__ residual, T* __restrict__ out, long HW,
                                                         int C, int Cp, float eps, int act, unsigned char* __restrict__ out8) {
  constexpr int EPP = Elem<T>::EPP;
  constexpr int UN = 4;
  const int n = blockIdx.y;
  const int cpr = Cp / EPP;
  const long total = HW * cpr;
  const size_t base = (size_t)n * HW * Cp;
  const int pc = (int)(((long)blockIdx.x * 256 + threadIdx.x) % cpr);
  const float nslope = neg_slope_of(act);
  ChanConsts<T> cc;
  cc.load(stats, n, Cp, C, pc, 1.f / (float)HW, eps);
  // Loads are UNCONDITIONAL (indices past the plane are clamped to its last piece and the result is dropped): a
  // predicated load compiles to a branch with its own s_waitcnt vmcnt(0), which serialises the UN loads of a thread.
  const long stride = (long)gridDim.x * 256;
  const long last = total - 1;
  for (long e0 = (long)blockIdx.x * 256 + threadIdx.x; e0 < total; e0 += stride * UN) {
    uint4 yv[UN], rv[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const long e = min(e0 + u * stride, last);
      yv[u] = *reinterpret_cast<const uint4*>(y + base + (size_t)e * EPP);
      if (residual != nullptr) rv[u] = *reinterpret_cast<const uint4*>(residual + base + (size_t)e * EPP);   // uniform
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const long e = e0 + u * stride;
      const T* vv = reinterpret_cast<const T*>(&yv[u]);
      const T* rr = reinterpret_cast<const T*>(&rv[u]);
      uint4 ov;
      T* oo = reinterpret_cast<T*>(&ov);
      float fq[EPP];
#pragma unroll
      for (int k = 0; k < EPP; ++k) {
        float f = act_fwd((to_f(vv[k]) - cc.mean[k]) * cc.rstd[k], nslope);
        if (residual != nullptr) f += to_f(rr[k]);
        f = pc * EPP + k < C ? f : 0.f;
        oo[k] = from_f<T>(f);
        fq[k] = fminf(fmaxf(f, -448.f), 448.f);
      }
      if (e < total) *reinterpret_cast<uint4*>(out + base + (size_t)e * EPP) = ov;
      if constexpr (EPP == 8) {
        if (out8 != nullptr && e < total) {                      // uniform
          int lo = __builtin_amdgcn_cvt_pk_fp8_f32(fq[0], fq[1], 0, false);
          lo = __builtin_amdgcn_cvt_pk_fp8_f32(fq[2], fq[3], lo, true);
          int hi = __builtin_amdgcn_cvt_pk_fp8_f32(fq[4], fq[5], 0, false);
          hi = __builtin_amdgcn_cvt_pk_fp8_f32(fq[6], fq[7], hi, true);
          *reinterpret_cast<int2*>(out8 + base + (size_t)e * EPP) = make_int2(lo, hi);
        }
      }
    }
  }
}

// Fixed-order fold of per-thread channel partials inside a workgroup of the channel-stationary kernels: thread t owns
// piece column (b0 + t) % cpr, so the threads of column pc are t0, t0 + cpr, ... with t0 = (pc - b0) mod cpr.  `red` is
// [256][NV] in LDS (NV values per thread); thread j < ncols * NV returns the sum of value (j % NV) of column (j / NV)'s
// threads, added in thread order -- no LDS atomics, the result does not depend on wave timing.
template <int NV>
__device__ __forceinline__ float column_fold(const float* red, int j, int cpr, int b0) {
  const int pc = j / NV, k = j - pc * NV;
  int t = pc - b0;
  if (t < 0) t += cpr;
  float s = 0.f;
  for (; t < 256; t += cpr) s += red[t * NV + k];
  return s;
}

// ---- backward pass 1: per (n,c) sums of g' and g' * yhat, g' = g * act'(yhat) ----------------------
// Fixed summation order end to end (round 3; the LDS and global float atomics of the earlier form made the gradients
// of every layer behind this pass differ in their last bits from run to run): threads fold per column in thread order,
// the workgroup stores its row of the partial table, the LAST workgroup of the sample (fold_arrive_last) adds the
// sample's rows in index order into bstats.
template <typename T>
__global__ __launch_bounds__(256) void in_act_bwd_reduce_kernel(const T* __restrict__ g, const T* __restrict__ y,
                                                                const float* __restrict__ stats, float* __restrict__ bstats,
                                                                long HW, int C, int Cp, float eps, int act,
                                                                float* __restrict__ part, unsigned* __restrict__ tickets) {
  constexpr int EPP = Elem<T>::EPP;
  constexpr int UN = 4;
  constexpr int NV = 2 * EPP;
  __shared__ float red[256 * NV];
  const int n = blockIdx.y;
  const int cpr = Cp / EPP;
  const long total = HW * cpr;
  const size_t base = (size_t)n * HW * Cp;
  const int b0 = (int)(((long)blockIdx.x * 256) % cpr);
  const int pc = (int)(((long)blockIdx.x * 256 + threadIdx.x) % cpr);
  const float nslope = neg_slope_of(act);
  ChanConsts<T> cc;
  cc.load(stats, n, Cp, C, pc, 1.f / (float)HW, eps);
  float a1[EPP], a2[EPP];
#pragma unroll
  for (int k = 0; k < EPP; ++k) a1[k] = a2[k] = 0.f;
  const long stride = (long)gridDim.x * 256;
  const long last = total - 1;
  for (long e0 = (long)blockIdx.x * 256 + threadIdx.x; e0 < total; e0 += stride * UN) {
    uint4 gv[UN], yv[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {                              // unconditional, clamped (see in_act_fwd_kernel)
      const long e = min(e0 + u * stride, last);
      gv[u] = *reinterpret_cast<const uint4*>(g + base + (size_t)e * EPP);
      yv[u] = *reinterpret_cast<const uint4*>(y + base + (size_t)e * EPP);
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const float live = e0 + u * stride < total ? 1.f : 0.f;
      const T* gg = reinterpret_cast<const T*>(&gv[u]);
      const T* yy = reinterpret_cast<const T*>(&yv[u]);
#pragma unroll
      for (int k = 0; k < EPP; ++k) {
        const float yh = (to_f(yy[k]) - cc.mean[k]) * cc.rstd[k];
        const float gp = to_f(gg[k]) * act_slope(yh, nslope) * live;
        a1[k] += gp; a2[k] += gp * yh;
      }
    }
  }
#pragma unroll
  for (int k = 0; k < EPP; ++k) { red[threadIdx.x * NV + 2 * k] = a1[k]; red[threadIdx.x * NV + 2 * k + 1] = a2[k]; }
  __syncthreads();
  // row of this workgroup: [2 * Cp] = (sum g', sum g' yhat) per channel, the layout of bstats
  float* row = part + ((size_t)n * gridDim.x + blockIdx.x) * (2 * (size_t)Cp);
  for (int j = threadIdx.x; j < 2 * Cp; j += 256) fold_store(row + j, column_fold<NV>(red, j, cpr, b0));
  if (!p2phd::fold_arrive_last(tickets + n, gridDim.x)) return;
  const float* rows = part + (size_t)n * gridDim.x * (2 * (size_t)Cp);
  const int nb = (int)gridDim.x;
  for (int j = threadIdx.x; j < 2 * Cp; j += 256) {
    float s = 0.f;
    for (int b = 0; b < nb; b += 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = fold_load(rows + (size_t)min(b + u, nb - 1) * (2 * (size_t)Cp) + j);   // eight loads in flight
#pragma unroll
      for (int u = 0; u < 8; ++u) s += b + u < nb ? v[u] : 0.f;
    }
    bstats[2 * (size_t)n * Cp + j] = s;
  }
}

// ---- backward pass 2: dy = rstd * (g' - mean(g') - yhat * mean(g' yhat)) ---------------------------
template <typename T>
__global__ __launch_bounds__(256) void in_act_bwd_apply_kernel(const T* __restrict__ g, const T* __restrict__ y,
                                                               const float* __restrict__ stats, const float* __restrict__ bstats,
                                                               T* __restrict__ dy, long HW, int C, int Cp, float eps, int act,
                                                               float* __restrict__ db) {
  constexpr int EPP = Elem<T>::EPP;
  constexpr int UN = 4;
  extern __shared__ float s_db[];                               // [Cp] bias-gradient partials (only when db != NULL)
  if (db != nullptr) {
    for (int c = threadIdx.x; c < Cp; c += 256) s_db[c] = 0.f;
    __syncthreads();
  }
  const int n = blockIdx.y;
  const float inv = 1.f / (float)HW;
  const int cpr = Cp / EPP;
  const long total = HW * cpr;
  const size_t base = (size_t)n * HW * Cp;
  const int pc = (int)(((long)blockIdx.x * 256 + threadIdx.x) % cpr);
  const float nslope = neg_slope_of(act);
  ChanConsts<T> cc;
  cc.load(stats, n, Cp, C, pc, inv, eps);
  float m1[EPP], m2[EPP], bsum[EPP];
  {
    const float4* b4 = reinterpret_cast<const float4*>(bstats + 2 * ((size_t)n * Cp + pc * EPP));
#pragma unroll
    for (int k = 0; k < EPP; k += 2) {
      const float4 v = b4[k >> 1];
      m1[k] = v.x * inv; m2[k] = v.y * inv; m1[k + 1] = v.z * inv; m2[k + 1] = v.w * inv;
      bsum[k] = bsum[k + 1] = 0.f;
    }
  }
  const long stride = (long)gridDim.x * 256;
  const long last = total - 1;
  for (long e0 = (long)blockIdx.x * 256 + threadIdx.x; e0 < total; e0 += stride * UN) {
    uint4 gv[UN], yv[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {                              // unconditional, clamped (see in_act_fwd_kernel)
      const long e = min(e0 + u * stride, last);
      gv[u] = *reinterpret_cast<const uint4*>(g + base + (size_t)e * EPP);
      yv[u] = *reinterpret_cast<const uint4*>(y + base + (size_t)e * EPP);
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const long e = e0 + u * stride;
      const float live = e < total ? 1.f : 0.f;
      const T* gg = reinterpret_cast<const T*>(&gv[u]);
      const T* yy = reinterpret_cast<const T*>(&yv[u]);
      uint4 ov;
      T* oo = reinterpret_cast<T*>(&ov);
#pragma unroll
      for (int k = 0; k < EPP; ++k) {
        const float yh = (to_f(yy[k]) - cc.mean[k]) * cc.rstd[k];
        const float gp = to_f(gg[k]) * act_slope(yh, nslope);
        oo[k] = from_f<T>(cc.rstd[k] * (gp - m1[k] - yh * m2[k]));
        bsum[k] += to_f(oo[k]) * live;                          // what the next kernels read, i.e. the rounded dy
      }
      if (e < total) *reinterpret_cast<uint4*>(dy + base + (size_t)e * EPP) = ov;
    }
  }
  if (db != nullptr) {
    // float atomics on purpose: db is the bias gradient of a conv IN FRONT of this InstanceNorm, whose exact value is 0
    // (the normalisation removes per-channel constants) -- both this sum and the reference's hold rounding noise only,
    // so a fixed summation order (fold_arrive_last: a tail of one workgroup per launch) would buy nothing here
#pragma unroll
    for (int k = 0; k < EPP; ++k) atomicAdd(&s_db[pc * EPP + k], bsum[k]);     // LDS atomics
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) atomicAdd(&db[c], s_db[c]);
  }
}

// ---- backward, small planes (HW <= 32 * ITERS pixels: the residual trunk, the coarse discriminator scale) ----------
// One launch instead of memset + reduce + apply: a workgroup owns sample n and 128 bytes of channels (8 pieces); its
// 256 threads are 8 piece columns x 32 pixel slices and keep their g and y pieces IN REGISTERS between the reduction
// and the apply step, so both tensors are read from HBM exactly once (3 tensor passes instead of 5).
template <typename T, int ITERS, int CGN>
__global__ __launch_bounds__(256) void in_act_bwd_fused_kernel(const T* __restrict__ g, const T* __restrict__ y,
                                                               const float* __restrict__ stats, T* __restrict__ dy, int HW,
                                                               int C, int Cp, float eps, int act, float* __restrict__ db,
                                                               T* __restrict__ rx, int W, int cblocks, int xcd_order) {
  constexpr int EPP = Elem<T>::EPP;
  // 1-D launch of cblocks x N workgroups.  A workgroup reads 16 * CGN bytes of every pixel -- half a 128-byte line for CGN = 4 --,
  // its neighbour in the channel direction the other half.  Workgroups are dealt round-robin over the 8 XCDs, so with the
  // plain order both halves of every line were fetched by two different L2s; the bijective chunk remap (as in wgrad_kernel,
  // conv.hip) puts a contiguous run of logical ids on each XCD, neighbours 8 dispatch slots apart.
  int bxl, nl;
  {
    const int Wg = (int)gridDim.x;
    int L = (int)blockIdx.x;
    if (xcd_order) {
      const int q = Wg >> 3, r = Wg & 7, xcd = L & 7, k = L >> 3;
      L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
    }
    bxl = L % cblocks; nl = L / cblocks;
  }
  // CGN piece columns x (256 / CGN) pixel slices per workgroup: CGN = 4 doubles the workgroups of a 64-channel block and
  // halves the registers a thread holds (the 8-column form ran 1.5 workgroups per CU on the trunk: latency-bound)
  constexpr int NSL = 256 / CGN;
  __shared__ float s_red[4][CGN][2 * EPP];
  __shared__ float s_tot[CGN][2 * EPP];
  const int tid = threadIdx.x, cg = tid & (CGN - 1), sl = tid / CGN, wave = tid >> 6;
  const int n = nl, cpr = Cp / EPP;
  const int pc = bxl * CGN + cg;
  const bool col_ok = pc < cpr;
  const size_t base = (size_t)n * HW * Cp + (size_t)(col_ok ? pc : 0) * EPP;
  const float nslope = neg_slope_of(act);
  ChanConsts<T> cc;
  cc.load(stats, n, Cp, C, col_ok ? pc : 0, 1.f / (float)HW, eps);
  // every load is unconditional (rows past the plane re-read its last pixel and are masked in the arithmetic): a
  // predicated load becomes a branch with its own s_waitcnt, which would serialise the 2 * ITERS loads of a thread
  uint4 gv[ITERS], yv[ITERS];
#pragma unroll
  for (int it = 0; it < ITERS; ++it) {
    const int p = min(sl + NSL * it, HW - 1);
    gv[it] = *reinterpret_cast<const uint4*>(g + base + (size_t)p * Cp);
    yv[it] = *reinterpret_cast<const uint4*>(y + base + (size_t)p * Cp);
  }
  float a[2 * EPP];
#pragma unroll
  for (int k = 0; k < 2 * EPP; ++k) a[k] = 0.f;
#pragma unroll
  for (int it = 0; it < ITERS; ++it) {
    const float live = (col_ok && sl + NSL * it < HW) ? 1.f : 0.f;
    const T* gg = reinterpret_cast<const T*>(&gv[it]);
    const T* yy = reinterpret_cast<const T*>(&yv[it]);
#pragma unroll
    for (int k = 0; k < EPP; ++k) {
      const float yh = (to_f(yy[k]) - cc.mean[k]) * cc.rstd[k];
      const float gp = to_f(gg[k]) * act_slope(yh, nslope) * live;
      a[k] += gp; a[EPP + k] += gp * yh;
    }
  }
  // slices of one piece column sit CGN lanes apart: reduce over the upper lane bits, then over the 4 waves through LDS
#pragma unroll
  for (int k = 0; k < 2 * EPP; ++k) {
#pragma unroll
    for (int o = CGN; o < 64; o <<= 1) a[k] += __shfl_xor(a[k], o);
  }
  if ((tid & 63) < CGN) {
#pragma unroll
    for (int k = 0; k < 2 * EPP; ++k) s_red[wave][cg][k] = a[k];
  }
  __syncthreads();
  if (tid < CGN * 2 * EPP) {
    const int c8 = tid / (2 * EPP), k = tid % (2 * EPP);
    s_tot[c8][k] = s_red[0][c8][k] + s_red[1][c8][k] + s_red[2][c8][k] + s_red[3][c8][k];
  }
  __syncthreads();
  const float inv = 1.f / (float)HW;
  float m1[EPP], m2[EPP], bsum[EPP];
#pragma unroll
  for (int k = 0; k < EPP; ++k) { m1[k] = s_tot[cg][k] * inv; m2[k] = s_tot[cg][EPP + k] * inv; bsum[k] = 0.f; }
#pragma unroll
  for (int it = 0; it < ITERS; ++it) {
    const int p = sl + NSL * it;
    if (col_ok && p < HW) {
      const T* gg = reinterpret_cast<const T*>(&gv[it]);
      const T* yy = reinterpret_cast<const T*>(&yv[it]);
      uint4 ov;
      T* oo = reinterpret_cast<T*>(&ov);
#pragma unroll
      for (int k = 0; k < EPP; ++k) {
        const float yh = (to_f(yy[k]) - cc.mean[k]) * cc.rstd[k];
        const float gp = to_f(gg[k]) * act_slope(yh, nslope);
        oo[k] = from_f<T>(cc.rstd[k] * (gp - m1[k] - yh * m2[k]));
        bsum[k] += to_f(oo[k]);
      }
      *reinterpret_cast<uint4*>(dy + base + (size_t)p * Cp) = ov;
    }
  }
  if (rx != nullptr) {
    // The conv this gradient belongs to sits behind ReflectionPad2d(1) (the residual trunk): its input-gradient GEMM reads dy
    // plus the pair-sum rows / columns of the reflection's adjoint (conv.hip, pad_mode 3).  This workgroup has just written
    // every pixel of its channels of the plane, so it appends them itself -- rx [N][2 (W + 2) + 2 H][Cp]: row H = dy[0] +
    // dy[2], row H + 1 = dy[H-3] + dy[H-1] (W + 2 columns each, the last two being the column sums of those), then columns
    // W = dy[:,0] + dy[:,2] and W + 1 = dy[:,W-3] + dy[:,W-1] for rows < H -- instead of a copy pass over the whole gradient.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // this thread's dy stores have left (same CU reads them back below)
    __syncthreads();
    const int H = HW / W, EX = 2 * (W + 2) + 2 * H;
    const T* dyn = dy + (size_t)n * HW * Cp;
    for (int task = tid; task < EX * CGN; task += 256) {
      const int e = task / CGN, pcx = bxl * CGN + (task - e * CGN);
      if (pcx >= cpr) continue;
      int hs[2], ws[2], nh = 1, nw = 1;
      int r, c;                                                  // expanded coordinates of this entry
      if (e < 2 * (W + 2)) { r = H + e / (W + 2); c = e % (W + 2); }
      else { const int q = e - 2 * (W + 2); r = q % H; c = W + q / H; }
      hs[0] = r; ws[0] = c;
      if (r == H) { hs[0] = 0; hs[nh++] = 2; } else if (r == H + 1) { hs[0] = H - 3; hs[nh++] = H - 1; }
      if (c == W) { ws[0] = 0; ws[nw++] = 2; } else if (c == W + 1) { ws[0] = W - 3; ws[nw++] = W - 1; }
      float accx[EPP];
#pragma unroll
      for (int k = 0; k < EPP; ++k) accx[k] = 0.f;
      for (int a2 = 0; a2 < nh; ++a2)
        for (int b2 = 0; b2 < nw; ++b2) {
          const uint4 v = *reinterpret_cast<const uint4*>(dyn + ((size_t)hs[a2] * W + ws[b2]) * Cp + pcx * EPP);
          const T* vv = reinterpret_cast<const T*>(&v);
#pragma unroll
          for (int k = 0; k < EPP; ++k) accx[k] += to_f(vv[k]);
        }
      uint4 ov;
      T* oo = reinterpret_cast<T*>(&ov);
#pragma unroll
      for (int k = 0; k < EPP; ++k) oo[k] = from_f<T>(accx[k]);
      *reinterpret_cast<uint4*>(rx + ((size_t)n * EX + e) * Cp + pcx * EPP) = ov;
    }
  }
  if (db != nullptr) {                                          // uniform branch: all threads take it
    __syncthreads();
#pragma unroll
    for (int k = 0; k < EPP; ++k) {
#pragma unroll
      for (int o = CGN; o < 64; o <<= 1) bsum[k] += __shfl_xor(bsum[k], o);
    }
    if ((tid & 63) < CGN) {
#pragma unroll
      for (int k = 0; k < EPP; ++k) s_red[wave][cg][k] = bsum[k];
    }
    __syncthreads();
    if (tid < CGN * EPP) {
      const int c8 = tid / EPP, k = tid % EPP;
      const int c = (bxl * CGN + c8) * EPP + k;
      if (c < C) atomicAdd(&db[c], s_red[0][c8][k] + s_red[1][c8][k] + s_red[2][c8][k] + s_red[3][c8][k]);
    }
  }
}

// ---- stand-alone InstanceNorm statistics of a plane (layers whose output does not come out of gconv_kernel) ----
// Workgroup (chunk, n) owns kPlaneChunk consecutive pixels of sample n: per channel it writes (sum, squared deviations
// from the chunk's own mean) into slot `chunk` of the partial table; launch_stats_merge combines the chunks.
constexpr int kPlaneChunk = 256;
template <typename T>
__global__ __launch_bounds__(256) void plane_stats_kernel(const T* __restrict__ y, float* __restrict__ table, long HW, int C, int Cp,
                                                          int slots) {
  extern __shared__ float s_sum[];                              // [Cp] sums, then [Cp] means
  const int n = blockIdx.y, chunk = blockIdx.x;
  const long p0 = (long)chunk * kPlaneChunk;
  const int cnt = (int)min((long)kPlaneChunk, HW - p0);
  const T* base = y + ((size_t)n * HW + p0) * Cp;
  for (int c = threadIdx.x; c < Cp; c += 256) {                 // channel-per-thread: rows are contiguous, Cp small here
    float a = 0.f;
    for (int p = 0; p < cnt; ++p) a += to_f(base[(size_t)p * Cp + c]);
    const float mean = a / (float)cnt;
    float m2 = 0.f;
    for (int p = 0; p < cnt; ++p) { const float d = to_f(base[(size_t)p * Cp + c]) - mean; m2 += d * d; }
    if (c < C) {
      float* sp = table + 2 * (((size_t)n * slots + chunk) * Cp + c);
      sp[0] = a; sp[1] = m2;
    }
  }
}

// ---- activation backward from the saved OUTPUT (conv layers whose activation is fused, no norm) ----
template <typename T>
__global__ void act_bwd_kernel(const T* __restrict__ g, const T* __restrict__ a, T* __restrict__ dx, long n_pieces, int act) {
  constexpr int EPP = Elem<T>::EPP;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n_pieces; e += (long)gridDim.x * blockDim.x) {
    const uint4 gv = *reinterpret_cast<const uint4*>(g + e * EPP);
    const uint4 av = *reinterpret_cast<const uint4*>(a + e * EPP);
    const T* gg = reinterpret_cast<const T*>(&gv);
    const T* aa = reinterpret_cast<const T*>(&av);
    uint4 ov;
    T* oo = reinterpret_cast<T*>(&ov);
#pragma unroll
    for (int k = 0; k < EPP; ++k) {
      const float o = to_f(aa[k]);
      float s = 1.f;
      if (act == P2PHD_ACT_TANH) s = 1.f - o * o;
      else if (act == P2PHD_ACT_LRELU) s = o > 0.f ? 1.f : 0.2f;
      else if (act == P2PHD_ACT_RELU) s = o > 0.f ? 1.f : 0.f;
      oo[k] = from_f<T>(to_f(gg[k]) * s);
    }
    *reinterpret_cast<uint4*>(dx + e * EPP) = ov;
  }
}

// Same, channel-stationary, with the conv bias gradient (column sums of the rounded dx) accumulated in the same pass:
// saves the separate column-sum read of dx for the discriminator layers that have a fused activation and no norm.
template <typename T>
__global__ __launch_bounds__(256) void act_bwd_db_kernel(const T* __restrict__ g, const T* __restrict__ a, T* __restrict__ dx,
                                                         long P, int C, int Cp, int act, float* __restrict__ db, int accumulate,
                                                         float* __restrict__ part, unsigned* __restrict__ ticket) {
  constexpr int EPP = Elem<T>::EPP;
  constexpr int UN = 4;
  __shared__ float red[256 * EPP];
  const int cpr = Cp / EPP;
  const long total = P * cpr, last = total - 1;
  const int b0 = (int)(((long)blockIdx.x * 256) % cpr);
  const float nslope = neg_slope_of(act);
  const bool is_tanh = act == P2PHD_ACT_TANH;
  float bsum[EPP];
#pragma unroll
  for (int k = 0; k < EPP; ++k) bsum[k] = 0.f;
  const long stride = (long)gridDim.x * 256;
  for (long e0 = (long)blockIdx.x * 256 + threadIdx.x; e0 < total; e0 += stride * UN) {
    uint4 gv[UN], av[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {                              // unconditional, clamped (see in_act_fwd_kernel)
      const long e = min(e0 + u * stride, last);
      gv[u] = *reinterpret_cast<const uint4*>(g + (size_t)e * EPP);
      av[u] = *reinterpret_cast<const uint4*>(a + (size_t)e * EPP);
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const long e = e0 + u * stride;
      const float live = e < total ? 1.f : 0.f;
      const T* gg = reinterpret_cast<const T*>(&gv[u]);
      const T* aa = reinterpret_cast<const T*>(&av[u]);
      uint4 ov;
      T* oo = reinterpret_cast<T*>(&ov);
#pragma unroll
      for (int k = 0; k < EPP; ++k) {
        const float o = to_f(aa[k]);
        const float s = is_tanh ? 1.f - o * o : (o > 0.f ? 1.f : nslope);
        oo[k] = from_f<T>(to_f(gg[k]) * s);
        bsum[k] += to_f(oo[k]) * live;
      }
      if (e < total) *reinterpret_cast<uint4*>(dx + (size_t)e * EPP) = ov;
    }
  }
  // fixed-order fold (see in_act_bwd_reduce_kernel): threads of a column in thread order, workgroups in index order
#pragma unroll
  for (int k = 0; k < EPP; ++k) red[threadIdx.x * EPP + k] = bsum[k];
  __syncthreads();
  float* row = part + (size_t)blockIdx.x * Cp;
  for (int j = threadIdx.x; j < Cp; j += 256) fold_store(row + j, column_fold<EPP>(red, j, cpr, b0));
  if (!p2phd::fold_arrive_last(ticket, gridDim.x)) return;
  const int nb = (int)gridDim.x;
  for (int c = threadIdx.x; c < C; c += 256) {
    float s = 0.f;
    for (int b = 0; b < nb; b += 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = fold_load(part + (size_t)min(b + u, nb - 1) * Cp + c);
#pragma unroll
      for (int u = 0; u < 8; ++u) s += b + u < nb ? v[u] : 0.f;
    }
    db[c] = accumulate ? db[c] + s : s;
  }
}

// ---- AvgPool2d(3, stride 2, pad 1, count_include_pad=False) -------------------------------------------
template <typename T>
__global__ void avgpool_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int N, int H, int W, int Ho, int Wo, int Cp) {
  constexpr int EPP = Elem<T>::EPP;
  const int cpr = Cp / EPP;
  const long total = (long)N * Ho * Wo * cpr;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int pc = (int)(e % cpr);
    long r = e / cpr;
    const int wo = (int)(r % Wo); r /= Wo;
    const int ho = (int)(r % Ho);
    const int n = (int)(r / Ho);
    float acc[EPP];
#pragma unroll
    for (int k = 0; k < EPP; ++k) acc[k] = 0.f;
    int cnt = 0;
    for (int a = 0; a < 3; ++a) {
      const int hi = 2 * ho - 1 + a;
      if (hi < 0 || hi >= H) continue;
      for (int b = 0; b < 3; ++b) {
        const int wi = 2 * wo - 1 + b;
        if (wi < 0 || wi >= W) continue;
        ++cnt;
        const uint4 v = *reinterpret_cast<const uint4*>(x + (((size_t)n * H + hi) * W + wi) * Cp + pc * EPP);
        const T* vv = reinterpret_cast<const T*>(&v);
#pragma unroll
        for (int k = 0; k < EPP; ++k) acc[k] += to_f(vv[k]);
      }
    }
    const float inv = 1.f / (float)cnt;
    uint4 ov;
    T* oo = reinterpret_cast<T*>(&ov);
#pragma unroll
    for (int k = 0; k < EPP; ++k) oo[k] = from_f<T>(acc[k] * inv);
    *reinterpret_cast<uint4*>(y + (size_t)e * EPP) = ov;
  }
}

__device__ __forceinline__ int pool_count(int o, int n) {   // valid taps of output index o along a dim of size n
  int c = 0;
  for (int a = 0; a < 3; ++a) { const int i = 2 * o - 1 + a; c += (i >= 0 && i < n); }
  return c;
}

template <typename T>
__global__ void avgpool_bwd_kernel(const T* __restrict__ dy, T* __restrict__ dx, int N, int H, int W, int Ho, int Wo, int Cp) {
  constexpr int EPP = Elem<T>::EPP;
  const int cpr = Cp / EPP;
  const long total = (long)N * H * W * cpr;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int pc = (int)(e % cpr);
    long r = e / cpr;
    const int j = (int)(r % W); r /= W;
    const int i = (int)(r % H);
    const int n = (int)(r / H);
    float acc[EPP];
#pragma unroll
    for (int k = 0; k < EPP; ++k) acc[k] = 0.f;
    // outputs ho whose window [2*ho-1, 2*ho+1] contains i
    for (int ho = i / 2; ho <= (i + 1) / 2; ++ho) {
      if (ho < 0 || ho >= Ho || 2 * ho - 1 > i || 2 * ho + 1 < i) continue;
      const int ch = pool_count(ho, H);
      for (int wo = j / 2; wo <= (j + 1) / 2; ++wo) {
        if (wo < 0 || wo >= Wo || 2 * wo - 1 > j || 2 * wo + 1 < j) continue;
        const float inv = 1.f / (float)(ch * pool_count(wo, W));
        const uint4 v = *reinterpret_cast<const uint4*>(dy + (((size_t)n * Ho + ho) * Wo + wo) * Cp + pc * EPP);
        const T* vv = reinterpret_cast<const T*>(&v);
#pragma unroll
        for (int k = 0; k < EPP; ++k) acc[k] += to_f(vv[k]) * inv;
      }
    }
    uint4 ov;
    T* oo = reinterpret_cast<T*>(&ov);
#pragma unroll
    for (int k = 0; k < EPP; ++k) oo[k] = from_f<T>(acc[k]);
    *reinterpret_cast<uint4*>(dx + (size_t)e * EPP) = ov;
  }
}

// ---- layout converters ---------------------------------------------------------------------------------
// src f32 [N,C,HW] -> channels [ch_off, ch_off+C) of dst [N,HW,Cp]; other channels of dst are left untouched
template <typename T>
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ src, T* __restrict__ dst, int N, int C, long HW, int Cp, int ch_off) {
  const long total = (long)N * C * HW;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const long p = e % HW;
    const long r = e / HW;
    const int c = (int)(r % C);
    const long n = r / C;
    dst[((size_t)n * HW + p) * Cp + ch_off + c] = from_f<T>(src[e]);
  }
}

// dst f32 [N,C,HW] <- channels [ch_off, ch_off+C) of src [N,HW,Cp]: a thread owns one 16-byte piece of a pixel (one load) and
// scatters its channels to their planes (consecutive threads = consecutive pixels when a pixel is one piece: coalesced)
template <typename T>
__global__ void nhwc_to_nchw_kernel(const T* __restrict__ src, float* __restrict__ dst, int N, int C, long HW, int Cp, int ch_off) {
  constexpr int EPP = Elem<T>::EPP;
  const int cpr = Cp / EPP;
  const long total = (long)N * HW * cpr;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const long p = e % HW;
    const long r = e / HW;
    const int pc = (int)(r % cpr);
    const long n = r / cpr;
    const int c_lo = pc * EPP - ch_off;                         // destination channel of the piece's first element
    if (c_lo + EPP <= 0 || c_lo >= C) continue;
    const uint4 v = *reinterpret_cast<const uint4*>(src + ((size_t)n * HW + p) * Cp + pc * EPP);
    const T* vv = reinterpret_cast<const T*>(&v);
#pragma unroll
    for (int k = 0; k < EPP; ++k) {
      const int c = c_lo + k;
      if (c >= 0 && c < C) dst[((size_t)n * C + c) * HW + p] = to_f(vv[k]);
    }
  }
}

// Up to four f32 [N,C_i,HW] tensors concatenated along channels -> dst [N,HW,Cp], pad channels written as zeros: every
// 16-byte piece of dst is written exactly once (no memset of the padded tensor, one launch for a concatenation).
struct CatSrc { const float* src[4]; int c0[5]; };               // source i holds channels [c0[i], c0[i+1])
template <typename T>
__global__ void nchw_cat_to_nhwc_kernel(CatSrc cs, int nsrc, T* __restrict__ dst, int N, long HW, int Cp) {
  constexpr int EPP = Elem<T>::EPP;
  const int cpr = Cp / EPP;
  const long total = (long)N * HW * cpr;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const long p = e % HW;
    const long r = e / HW;
    const int pc = (int)(r % cpr);
    const long n = r / cpr;
    uint4 ov;
    T* oo = reinterpret_cast<T*>(&ov);
#pragma unroll
    for (int k = 0; k < EPP; ++k) {
      const int c = pc * EPP + k;
      float v = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (i < nsrc && c >= cs.c0[i] && c < cs.c0[i + 1]) v = cs.src[i][((size_t)n * (cs.c0[i + 1] - cs.c0[i]) + (c - cs.c0[i])) * HW + p];
      oo[k] = from_f<T>(v);
    }
    *reinterpret_cast<uint4*>(dst + ((size_t)n * HW + p) * Cp + pc * EPP) = ov;
  }
}

inline int grid_for(long work, int cap = 8192) { return (int)std::max<long>(1, std::min<long>((work + 255) / 256, cap)); }

#define DISPATCH_T(dtype, CALL_BF16, CALL_F32, name)                 \
  if ((dtype) == P2PHD_BF16) { CALL_BF16; }                          \
  else if ((dtype) == P2PHD_F32) { CALL_F32; }                       \
  else { p2phd::set_error(name ": unsupported dtype %d", (dtype)); return P2PHD_EUNSUPPORTED; }

}  // namespace

namespace {
// grid.x for the channel-stationary kernels: stride gx * 256 must be a multiple of cpr; few fat blocks so that the
// per-thread constant setup and the end-of-block atomics are amortised
// smallest grid.x whose stride (grid.x * 256 pieces) is a multiple of the pieces per pixel
int stationary_unit(int cpr) {
  int gcd = cpr, b = 256;
  while (b) { const int t = gcd % b; gcd = b; b = t; }
  return cpr / gcd;
}
int stationary_grid(long HW, int cpr, int N) {
  const int unit = stationary_unit(cpr);
  long want = (HW * cpr + 256 * 8 - 1) / (256 * 8);            // >= 8 pieces per thread
  want = std::min<long>(want, std::max(1, 2048 / std::max(N, 1)));
  int gx = (int)std::max<long>(unit, want / unit * unit);
  return gx;
}
}  // namespace

namespace {
// One wavefront per (sample, channel): lane l folds slots l, l + 64, ... with Chan's pairwise update in a fixed order,
// then the 64 lane results are folded by a fixed shuffle tree -- the result does not depend on timing.
__device__ __forceinline__ void chan_merge(float& na, float& ma, float& qa, float nb, float mb, float qb) {
  const float n = na + nb;
  if (nb > 0.f) {
    const float d = mb - ma, f = nb / n;
    ma += d * f;
    qa += qb + d * d * na * f;
    na = n;
  }
}

// few slots (the residual trunk: 4-16 per sample): one THREAD per (sample, channel), consecutive threads on consecutive
// channels (coalesced float2 reads), slots folded in order
__global__ __launch_bounds__(256) void stats_merge_small_kernel(const float* __restrict__ table, float* __restrict__ stats, int slots,
                                                                int ncls, int Cp, int C, long npix, int slot_rows) {
  const int n = blockIdx.y, c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  float cn = 0.f, cm = 0.f, cq = 0.f;
  const int total = slots * ncls;                               // partial index e = s * ncls + k
  const float2* tp = reinterpret_cast<const float2*>(table) + (size_t)n * total * Cp + c;
  for (int e0 = 0; e0 < total; e0 += 8) {
    float2 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = tp[(size_t)min(e0 + u, total - 1) * Cp];   // eight independent loads in flight
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int e = e0 + u;
      const int sl = e / ncls;
      const float cnt = (float)min((long)slot_rows, max(npix - (long)sl * slot_rows, 0l));
      if (e < total && cnt > 0.f) chan_merge(cn, cm, cq, cnt, v[u].x / cnt, v[u].y);
    }
  }
  *reinterpret_cast<float2*>(stats + 2 * ((size_t)n * Cp + c)) = make_float2(cm, cq);
}

__global__ __launch_bounds__(256) void stats_merge_kernel(const float* __restrict__ table, float* __restrict__ stats, int slots,
                                                          int ncls, int Cp, int C, long npix, int slot_rows) {
  const int n = blockIdx.y, lane = threadIdx.x & 63;
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (c >= C) return;                                           // wave-uniform
  float cn = 0.f, cm = 0.f, cq = 0.f;
  for (int s = lane; s < slots; s += 64) {
    const float cnt = (float)min((long)slot_rows, max(npix - (long)s * slot_rows, 0l));
    for (int k = 0; k < ncls; ++k) {
      const float* sp = table + 2 * ((((size_t)n * slots + s) * ncls + k) * Cp + c);
      if (cnt > 0.f) chan_merge(cn, cm, cq, cnt, sp[0] / cnt, sp[1]);
    }
  }
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const float on = __shfl_xor(cn, o), om = __shfl_xor(cm, o), oq = __shfl_xor(cq, o);
    // both partners must compute the SAME merged value: order the pair by lane so the update is symmetric
    float an = (lane & o) ? on : cn, am = (lane & o) ? om : cm, aq = (lane & o) ? oq : cq;
    const float bn = (lane & o) ? cn : on, bm = (lane & o) ? cm : om, bq = (lane & o) ? cq : oq;
    chan_merge(an, am, aq, bn, bm, bq);
    cn = an; cm = am; cq = aq;
  }
  if (lane == 0) {
    float* o = stats + 2 * ((size_t)n * Cp + c);
    o[0] = cm; o[1] = cq;
  }
}
}  // namespace

namespace p2phd {
int launch_stats_merge(const float* table, float* stats, int N, int slots, int ncls, int Cp, int C, long npix, int slot_rows,
                       hipStream_t st) {
  if (N == 0 || slots == 0 || C == 0) return P2PHD_OK;
  if (slots * ncls <= 32) {
    hipLaunchKernelGGL(stats_merge_small_kernel, dim3((unsigned)((C + 255) / 256), (unsigned)N), dim3(256), 0, st, table, stats, slots,
                       ncls, Cp, C, npix, slot_rows);
    return check_launch("stats_merge");
  }
  hipLaunchKernelGGL(stats_merge_kernel, dim3((unsigned)((C + 3) / 4), (unsigned)N), dim3(256), 0, st, table, stats, slots, ncls, Cp, C,
                     npix, slot_rows);
  return check_launch("stats_merge");
}

size_t plane_stats_scratch_floats(int N, long HW, int C) {
  const int Cp = (C + 7) & ~7;
  return (size_t)N * ((HW + kPlaneChunk - 1) / kPlaneChunk) * Cp * 2;
}

int launch_plane_stats(int dtype, const void* y, float* stats, float* scratch, int N, long HW, int C, hipStream_t st) {
  const int Cp = (C + 7) & ~7;
  if (N == 0 || HW == 0) return P2PHD_OK;
  const int slots = (int)((HW + kPlaneChunk - 1) / kPlaneChunk);
  dim3 grid((unsigned)slots, (unsigned)N);
  if (dtype == P2PHD_BF16)
    hipLaunchKernelGGL(plane_stats_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)y, scratch, HW, C, Cp, slots);
  else
    hipLaunchKernelGGL(plane_stats_kernel<float>, grid, dim3(256), 0, st, (const float*)y, scratch, HW, C, Cp, slots);
  if (int rc = check_launch("plane_stats")) return rc;
  return launch_stats_merge(scratch, stats, N, slots, 1, Cp, C, HW, kPlaneChunk, st);
}
}  // namespace p2phd

static int instnorm_act_fwd_impl(int dtype, const void* y, const float* stats, const void* residual, void* out, void* out8,
                                 int N, int64_t HW, int C, float eps, int act, void* stream) {
  const int Cp = (C + 7) & ~7;
  P2PHD_REQUIRE(Cp <= kMaxCp, "instnorm: at most %d channels", kMaxCp);
  if (N == 0 || HW == 0) return P2PHD_OK;
  P2PHD_REQUIRE(y && stats && out, "instnorm_act_fwd: null pointer");
  P2PHD_REQUIRE(out8 == nullptr || dtype == P2PHD_BF16, "instnorm_act_fwd_q8: the fp8 twin goes with bf16 activations");
  const int epp = dtype == P2PHD_BF16 ? 8 : 4;
  dim3 grid(stationary_grid(HW, Cp / epp, N), N);
  hipStream_t st = (hipStream_t)stream;
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(in_act_fwd_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)y, stats, (const bf16_t*)residual, (bf16_t*)out, (long)HW, C, Cp, eps, act, (unsigned char*)out8),
             hipLaunchKernelGGL(in_act_fwd_kernel<float>, grid, dim3(256), 0, st, (const float*)y, stats, (const float*)residual, (float*)out, (long)HW, C, Cp, eps, act, (unsigned char*)nullptr),
             "instnorm_act_fwd");
  return p2phd::check_launch("instnorm_act_fwd");
}

extern "C" int p2phd_instnorm_act_fwd(int dtype, const void* y, const float* stats, const void* residual, void* out,
                                      int N, int64_t HW, int C, float eps, int act, void* stream) {
  return instnorm_act_fwd_impl(dtype, y, stats, residual, out, nullptr, N, HW, C, eps, act, stream);
}

extern "C" int p2phd_instnorm_act_fwd_q8(int dtype, const void* y, const float* stats, const void* residual, void* out, void* out8,
                                         int N, int64_t HW, int C, float eps, int act, void* stream) {
  P2PHD_REQUIRE(out8 != nullptr, "instnorm_act_fwd_q8: null fp8 output");
  return instnorm_act_fwd_impl(dtype, y, stats, residual, out, out8, N, HW, C, eps, act, stream);
}

// planes small enough (and numerous enough) for the single-launch register-resident backward
static bool bwd_single_launch(int dtype, int N, int64_t HW, int C) {
  const int Cp = (C + 7) & ~7;
  const int epp = dtype == P2PHD_BF16 ? 8 : 4;
  const int cblocks4 = (Cp / epp + 3) / 4;
  return HW <= 640 && (long)N * cblocks4 >= 128;
}

extern "C" int p2phd_instnorm_act_bwd_two_pass(int dtype, int N, int64_t HW, int C) {
  return (N > 0 && HW > 0 && !bwd_single_launch(dtype, N, HW, C)) ? 1 : 0;
}

static int instnorm_act_bwd_impl(int dtype, const void* g, const void* y, const float* stats, float* bstats, void* dy,
                                 float* db, int db_accumulate, int N, int64_t HW, int C, float eps, int act, void* stream,
                                 bool sums_given = false, void* rx = nullptr, int W = 0) {
  const int Cp = (C + 7) & ~7;
  P2PHD_REQUIRE(Cp <= kMaxCp, "instnorm: at most %d channels", kMaxCp);
  if (N == 0 || HW == 0) return P2PHD_OK;
  P2PHD_REQUIRE(g && y && stats && (bstats || rx) && dy, "instnorm_act_bwd: null pointer");
  P2PHD_REQUIRE(rx == nullptr || (bwd_single_launch(dtype, N, HW, C) && W >= 4 && HW % W == 0 && HW / W >= 4),
                "instnorm_act_bwd_rx: reflection extras need a plane that takes the single-launch backward (p2phd_conv_reflect_extras_elems)");
  hipStream_t st = (hipStream_t)stream;
  if (db != nullptr && !db_accumulate) (void)hipMemsetAsync(db, 0, sizeof(float) * (size_t)C, st);
  const int epp = dtype == P2PHD_BF16 ? 8 : 4;
  // small planes with enough (sample, channel block) pairs to fill the chip: single-launch register-resident variant;
  // 4 piece columns x 64 slices per workgroup when that still leaves <= 10 iterations (planes <= 640 pixels)
  const int cblocks4 = (Cp / epp + 3) / 4;
  P2PHD_REQUIRE(!sums_given || !bwd_single_launch(dtype, N, HW, C), "instnorm_act_bwd_apply: this plane takes the single-launch backward (p2phd_instnorm_act_bwd_two_pass)");
  if (bwd_single_launch(dtype, N, HW, C)) {
    dim3 fgrid((unsigned)cblocks4 * (unsigned)N);
#define P2PHD_FUSED_BWD(TT, IT) hipLaunchKernelGGL((in_act_bwd_fused_kernel<TT, IT, 4>), fgrid, dim3(256), 0, st, (const TT*)g, (const TT*)y, stats, (TT*)dy, (int)HW, C, Cp, eps, act, db, (TT*)rx, W, cblocks4, p2phd::g_opt_wgrad_xcd)
    if (dtype == P2PHD_BF16) { if (HW <= 512) P2PHD_FUSED_BWD(bf16_t, 8); else P2PHD_FUSED_BWD(bf16_t, 10); }
    else if (dtype == P2PHD_F32) { if (HW <= 512) P2PHD_FUSED_BWD(float, 8); else P2PHD_FUSED_BWD(float, 10); }
    else { p2phd::set_error("instnorm_act_bwd: unsupported dtype %d", dtype); return P2PHD_EUNSUPPORTED; }
#undef P2PHD_FUSED_BWD
    return p2phd::check_launch("instnorm_act_bwd(fused)");
  }
  dim3 grid(stationary_grid(HW, Cp / epp, N), N);
  if (!sums_given) {
    // partial table of the fixed-order reduction: one row of 2 * Cp floats per workgroup, one ticket per sample
    const p2phd::FoldScratch fs = p2phd::fold_scratch(p2phd::FOLD_IN_BWD, st);
    if (fs.part == nullptr) return P2PHD_EINVAL;                 // (refused: error text set by fold_scratch)
    P2PHD_REQUIRE(N <= fs.tickets, "instnorm_act_bwd: at most %d samples per call", fs.tickets);
    const int unit = stationary_unit(Cp / epp);
    const long rows_max = (long)(fs.floats / (2 * (size_t)Cp * (size_t)N));
    P2PHD_REQUIRE(rows_max >= unit, "instnorm_act_bwd: N * channels too large for the reduction scratch");
    dim3 rgrid(std::min<long>(grid.x, rows_max / unit * unit), N);
    DISPATCH_T(dtype,
               hipLaunchKernelGGL(in_act_bwd_reduce_kernel<bf16_t>, rgrid, dim3(256), 0, st, (const bf16_t*)g, (const bf16_t*)y, stats, bstats, (long)HW, C, Cp, eps, act, fs.part, fs.ticket),
               hipLaunchKernelGGL(in_act_bwd_reduce_kernel<float>, rgrid, dim3(256), 0, st, (const float*)g, (const float*)y, stats, bstats, (long)HW, C, Cp, eps, act, fs.part, fs.ticket),
               "instnorm_act_bwd");
    if (int rc = p2phd::check_launch("instnorm_act_bwd(reduce)")) return rc;
  }
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(in_act_bwd_apply_kernel<bf16_t>, grid, dim3(256), Cp * sizeof(float), st, (const bf16_t*)g, (const bf16_t*)y, stats, bstats, (bf16_t*)dy, (long)HW, C, Cp, eps, act, db),
             hipLaunchKernelGGL(in_act_bwd_apply_kernel<float>, grid, dim3(256), Cp * sizeof(float), st, (const float*)g, (const float*)y, stats, bstats, (float*)dy, (long)HW, C, Cp, eps, act, db),
             "instnorm_act_bwd");
  return p2phd::check_launch("instnorm_act_bwd(apply)");
}

extern "C" int p2phd_instnorm_act_bwd(int dtype, const void* g, const void* y, const float* stats, float* bstats, void* dy,
                                      float* db, int N, int64_t HW, int C, float eps, int act, void* stream) {
  return instnorm_act_bwd_impl(dtype, g, y, stats, bstats, dy, db, 0, N, HW, C, eps, act, stream);
}

extern "C" int p2phd_instnorm_act_bwd_acc(int dtype, const void* g, const void* y, const float* stats, float* bstats, void* dy,
                                          float* db, int N, int64_t HW, int C, float eps, int act, void* stream) {
  return instnorm_act_bwd_impl(dtype, g, y, stats, bstats, dy, db, 1, N, HW, C, eps, act, stream);
}

extern "C" int p2phd_instnorm_act_bwd_rx(int dtype, const void* g, const void* y, const float* stats, void* dy, float* db,
                                         int db_accumulate, int N, int H, int W, int C, float eps, int act, void* reflect_extras,
                                         void* stream) {
  P2PHD_REQUIRE(reflect_extras != nullptr, "instnorm_act_bwd_rx: null extras pointer");
  return instnorm_act_bwd_impl(dtype, g, y, stats, nullptr, dy, db, db_accumulate, N, (int64_t)H * W, C, eps, act, stream, false, reflect_extras, W);
}

extern "C" int p2phd_instnorm_act_bwd_apply(int dtype, const void* g, const void* y, const float* stats, const float* bstats,
                                            void* dy, float* db, int db_accumulate, int N, int64_t HW, int C, float eps, int act,
                                            void* stream) {
  return instnorm_act_bwd_impl(dtype, g, y, stats, const_cast<float*>(bstats), dy, db, db_accumulate, N, HW, C, eps, act, stream, true);
}

extern "C" int p2phd_act_bwd(int dtype, const void* g, const void* a, void* dx, int64_t n_elems, int act, void* stream) {
  if (n_elems == 0) return P2PHD_OK;
  P2PHD_REQUIRE(g && a && dx, "act_bwd: null pointer");
  const int epp = dtype == P2PHD_BF16 ? 8 : 4;
  P2PHD_REQUIRE(n_elems % epp == 0, "act_bwd: element count must be a multiple of %d", epp);
  const long np = n_elems / epp;
  hipStream_t st = (hipStream_t)stream;
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(act_bwd_kernel<bf16_t>, dim3(grid_for(np)), dim3(256), 0, st, (const bf16_t*)g, (const bf16_t*)a, (bf16_t*)dx, np, act),
             hipLaunchKernelGGL(act_bwd_kernel<float>, dim3(grid_for(np)), dim3(256), 0, st, (const float*)g, (const float*)a, (float*)dx, np, act),
             "act_bwd");
  return p2phd::check_launch("act_bwd");
}

extern "C" int p2phd_avgpool3s2_fwd(int dtype, const void* x, void* y, int N, int H, int W, int C, void* stream) {
  const int Cp = (C + 7) & ~7;
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  if (N == 0) return P2PHD_OK;
  P2PHD_REQUIRE(x && y && H >= 1 && W >= 1, "avgpool_fwd: bad arguments");
  const int epp = dtype == P2PHD_BF16 ? 8 : 4;
  const long work = (long)N * Ho * Wo * (Cp / epp);
  hipStream_t st = (hipStream_t)stream;
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(avgpool_fwd_kernel<bf16_t>, dim3(grid_for(work)), dim3(256), 0, st, (const bf16_t*)x, (bf16_t*)y, N, H, W, Ho, Wo, Cp),
             hipLaunchKernelGGL(avgpool_fwd_kernel<float>, dim3(grid_for(work)), dim3(256), 0, st, (const float*)x, (float*)y, N, H, W, Ho, Wo, Cp),
             "avgpool_fwd");
  return p2phd::check_launch("avgpool_fwd");
}

extern "C" int p2phd_avgpool3s2_bwd(int dtype, const void* dy, void* dx, int N, int H, int W, int C, void* stream) {
  const int Cp = (C + 7) & ~7;
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  if (N == 0) return P2PHD_OK;
  P2PHD_REQUIRE(dy && dx && H >= 1 && W >= 1, "avgpool_bwd: bad arguments");
  const int epp = dtype == P2PHD_BF16 ? 8 : 4;
  const long work = (long)N * H * W * (Cp / epp);
  hipStream_t st = (hipStream_t)stream;
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(avgpool_bwd_kernel<bf16_t>, dim3(grid_for(work)), dim3(256), 0, st, (const bf16_t*)dy, (bf16_t*)dx, N, H, W, Ho, Wo, Cp),
             hipLaunchKernelGGL(avgpool_bwd_kernel<float>, dim3(grid_for(work)), dim3(256), 0, st, (const float*)dy, (float*)dx, N, H, W, Ho, Wo, Cp),
             "avgpool_bwd");
  return p2phd::check_launch("avgpool_bwd");
}

extern "C" int p2phd_nchw_to_nhwc(int dtype, const float* src, void* dst, int N, int C, int64_t HW, int Cp, int ch_off, void* stream) {
  if ((long)N * C * HW == 0) return P2PHD_OK;
  P2PHD_REQUIRE(src && dst && ch_off >= 0 && ch_off + C <= Cp, "nchw_to_nhwc: bad arguments");
  const long work = (long)N * C * HW;
  hipStream_t st = (hipStream_t)stream;
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(nchw_to_nhwc_kernel<bf16_t>, dim3(grid_for(work)), dim3(256), 0, st, src, (bf16_t*)dst, N, C, (long)HW, Cp, ch_off),
             hipLaunchKernelGGL(nchw_to_nhwc_kernel<float>, dim3(grid_for(work)), dim3(256), 0, st, src, (float*)dst, N, C, (long)HW, Cp, ch_off),
             "nchw_to_nhwc");
  return p2phd::check_launch("nchw_to_nhwc");
}

extern "C" int p2phd_nchw_cat_to_nhwc(int dtype, const float* const* srcs, const int* chans, int nsrc, void* dst, int N, int64_t HW,
                                      int Cp, void* stream) {
  P2PHD_REQUIRE(srcs && chans && dst && nsrc >= 1 && nsrc <= 4 && Cp % 8 == 0, "nchw_cat_to_nhwc: bad arguments");
  CatSrc cs{};
  int c = 0;
  for (int i = 0; i < nsrc; ++i) {
    P2PHD_REQUIRE(srcs[i] != nullptr && chans[i] >= 1, "nchw_cat_to_nhwc: bad source %d", i);
    cs.src[i] = srcs[i]; cs.c0[i] = c; c += chans[i];
  }
  cs.c0[nsrc] = c;
  for (int i = nsrc; i < 4; ++i) { cs.src[i] = nullptr; cs.c0[i + 1] = c; }
  P2PHD_REQUIRE(c <= Cp, "nchw_cat_to_nhwc: %d channels do not fit a pitch of %d", c, Cp);
  if ((long)N * HW == 0) return P2PHD_OK;
  hipStream_t st = (hipStream_t)stream;
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(nchw_cat_to_nhwc_kernel<bf16_t>, dim3(grid_for((long)N * HW * (Cp / 8))), dim3(256), 0, st, cs, nsrc, (bf16_t*)dst, N, (long)HW, Cp),
             hipLaunchKernelGGL(nchw_cat_to_nhwc_kernel<float>, dim3(grid_for((long)N * HW * (Cp / 4))), dim3(256), 0, st, cs, nsrc, (float*)dst, N, (long)HW, Cp),
             "nchw_cat_to_nhwc");
  return p2phd::check_launch("nchw_cat_to_nhwc");
}

extern "C" int p2phd_nhwc_to_nchw(int dtype, const void* src, float* dst, int N, int C, int64_t HW, int Cp, int ch_off, void* stream) {
  if ((long)N * C * HW == 0) return P2PHD_OK;
  P2PHD_REQUIRE(src && dst && ch_off >= 0 && ch_off + C <= Cp && Cp % 8 == 0, "nhwc_to_nchw: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(nhwc_to_nchw_kernel<bf16_t>, dim3(grid_for((long)N * HW * (Cp / 8))), dim3(256), 0, st, (const bf16_t*)src, dst, N, C, (long)HW, Cp, ch_off),
             hipLaunchKernelGGL(nhwc_to_nchw_kernel<float>, dim3(grid_for((long)N * HW * (Cp / 4))), dim3(256), 0, st, (const float*)src, dst, N, C, (long)HW, Cp, ch_off),
             "nhwc_to_nchw");
  return p2phd::check_launch("nhwc_to_nchw");
}

extern "C" int p2phd_act_bwd_db(int dtype, const void* g, const void* a, void* dx, int64_t n_pixels, int C, int act, float* db,
                                int db_accumulate, void* stream) {
  const int Cp = (C + 7) & ~7;
  P2PHD_REQUIRE(Cp <= kMaxCp, "act_bwd_db: at most %d channels", kMaxCp);
  P2PHD_REQUIRE(n_pixels >= 0 && C >= 1, "act_bwd_db: bad geometry");
  P2PHD_REQUIRE(db != nullptr, "act_bwd_db: null bias-gradient pointer (use p2phd_act_bwd)");
  hipStream_t st = (hipStream_t)stream;
  if (n_pixels == 0) {
    if (!db_accumulate) (void)hipMemsetAsync(db, 0, sizeof(float) * (size_t)C, st);
    return P2PHD_OK;
  }
  P2PHD_REQUIRE(g && a && dx, "act_bwd_db: null pointer");
  const int epp = dtype == P2PHD_BF16 ? 8 : 4;
  // bias gradient with a REAL value (no normalisation behind the conv): summed in a fixed order, see in_act_bwd_reduce_kernel
  const p2phd::FoldScratch fs = p2phd::fold_scratch(p2phd::FOLD_ACT_DB, st);
  if (fs.part == nullptr) return P2PHD_EINVAL;                   // (refused: error text set by fold_scratch)
  const int unit = stationary_unit(Cp / epp);
  const long rows_max = (long)(fs.floats / (size_t)Cp);
  P2PHD_REQUIRE(rows_max >= unit, "act_bwd_db: too many channels for the reduction scratch");
  dim3 grid(std::min<long>(stationary_grid(n_pixels, Cp / epp, 1), rows_max / unit * unit));
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(act_bwd_db_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)g, (const bf16_t*)a, (bf16_t*)dx, (long)n_pixels, C, Cp, act, db, db_accumulate, fs.part, fs.ticket),
             hipLaunchKernelGGL(act_bwd_db_kernel<float>, grid, dim3(256), 0, st, (const float*)g, (const float*)a, (float*)dx, (long)n_pixels, C, Cp, act, db, db_accumulate, fs.part, fs.ticket),
             "act_bwd_db");
  return p2phd::check_launch("act_bwd_db");
}
