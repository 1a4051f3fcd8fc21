// Scalar losses of the GAN step and the fused Adam update.
//   LSGAN criterion   models/networks.py:68-110  : mean((pred - target)^2), target 1.0 / 0.0
//   feature matching  models/pix2pixHD_model.py:391-398 : mean(|a - b|), b detached
//   Adam              torch.optim.Adam(lr, betas=(beta1, 0.999)) as built at pix2pixHD_model.py:131,140
// Losses reduce on the device into a float accumulator (no host sync); their backward kernels read the upstream
// gradient from device memory too, so the whole step can be enqueued without the host waiting on a value.
#include "common.h"

namespace {

typedef p2phd_h16 bf16_t;                 // the library's 16-bit storage type: bf16, or fp16 in the -DP2PHD_F16 build (common.h)
__device__ __forceinline__ float to_f(float v) { return v; }
__device__ __forceinline__ float to_f(bf16_t v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f(float v);
template <> __device__ __forceinline__ float from_f<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f<bf16_t>(float v) { return (bf16_t)v; }

__device__ __forceinline__ float block_sum(float v, float* red) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float t = 0.f;
  if (threadIdx.x == 0) for (unsigned i = 0; i < (blockDim.x + 63) / 64; ++i) t += red[i];
  __syncthreads();
  return t;   // valid in thread 0
}

template <typename T> struct Elem;
template <> struct Elem<float> { static constexpr int EPP = 4; };
template <> struct Elem<bf16_t> { static constexpr int EPP = 8; };

// kind 0: (a - target)^2 ; kind 1: |a - b|.  a,b are [P][Cp] with C valid channels.  out += coeff * sum / (P*C)
// 16-byte pieces, two in flight per thread; pad channels are masked
template <typename T>
__global__ __launch_bounds__(256) void loss_fwd_kernel(int kind, const T* __restrict__ a, const T* __restrict__ b, float target,
                                                       long P, int C, int Cp, float coeff, float* __restrict__ out,
                                                       float* __restrict__ part, unsigned* __restrict__ ticket) {
  constexpr int EPP = Elem<T>::EPP;
  __shared__ float red[4];
  const int cpr = Cp / EPP;
  const long pieces = P * cpr;
  float acc = 0.f;
  // U pieces of each operand are requested before the first is used (round 5: one piece per iteration kept 32 KB in flight per
  // CU -- 2.7 TB/s); the channel of a piece is walked with a counter (the 64-bit e % cpr was a division per piece); layers
  // without pad channels (every feature-matching term) skip the channel mask.  The per-thread order of the additions is
  // unchanged: same bits as before.
  constexpr int U = 4;
  const long stride = (long)gridDim.x * 256;
  const bool masked = C < Cp;
  long e0 = (long)blockIdx.x * 256 + threadIdx.x;
  int cq = (int)(e0 % cpr);                                     // piece column of e0; advances by stride % cpr per piece
  const int cstep = (int)(stride % cpr);
  for (; e0 < pieces; e0 += U * stride) {
    uint4 av[U], bv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long e = min(e0 + u * stride, pieces - 1);             // unconditional, clamped
      av[u] = *reinterpret_cast<const uint4*>(a + e * EPP);
      bv[u] = kind == 1 ? *reinterpret_cast<const uint4*>(b + e * EPP) : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int c0 = cq * EPP;
      cq += cstep; cq -= cq >= cpr ? cpr : 0;
      if (e0 + u * stride >= pieces) continue;
      const T* aa = reinterpret_cast<const T*>(&av[u]);
      const T* bb = reinterpret_cast<const T*>(&bv[u]);
#pragma unroll
      for (int k = 0; k < EPP; ++k) {
        if (!masked || c0 + k < C) {
          const float x = to_f(aa[k]);
          if (kind == 0) { const float dlt = x - target; acc += dlt * dlt; }
          else acc += fabsf(x - to_f(bb[k]));
        }
      }
    }
  }
  // workgroup sums are stored and added by the last workgroup in index order (fold_arrive_last): the loss value has the
  // same bits on every run (the earlier form added them with a float atomic each)
  const float t = block_sum(acc, red);
  if (threadIdx.x == 0) p2phd::fold_store(part + blockIdx.x, t);
  if (!p2phd::fold_arrive_last(ticket, gridDim.x)) return;
  float s = 0.f;
  for (int b = threadIdx.x; b < (int)gridDim.x; b += 256) s += p2phd::fold_load(part + b);
  const float tot = block_sum(s, red);
  if (threadIdx.x == 0) *out += tot * coeff / (float)(P * C);
}

// da[p][c] = (*gup) * coeff * f'(a) / (P*C), pad channels zero
template <typename T>
__global__ __launch_bounds__(256) void loss_bwd_kernel(int kind, const T* __restrict__ a, const T* __restrict__ b, float target,
                                                       long P, int C, int Cp, float coeff, const float* __restrict__ gup,
                                                       T* __restrict__ da) {
  constexpr int EPP = Elem<T>::EPP;
  const int cpr = Cp / EPP;
  const long pieces = P * cpr;
  const float s = (*gup) * coeff / (float)(P * C);
  constexpr int U = 4;                                           // (as loss_fwd_kernel: U pieces in flight, counter instead of e % cpr)
  const long stride = (long)gridDim.x * 256;
  const bool masked = C < Cp;
  long e0 = (long)blockIdx.x * 256 + threadIdx.x;
  int cq = (int)(e0 % cpr);
  const int cstep = (int)(stride % cpr);
  for (; e0 < pieces; e0 += U * stride) {
    uint4 av[U], bv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long e = min(e0 + u * stride, pieces - 1);
      av[u] = *reinterpret_cast<const uint4*>(a + e * EPP);
      bv[u] = kind == 1 ? *reinterpret_cast<const uint4*>(b + e * EPP) : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int c0 = cq * EPP;
      cq += cstep; cq -= cq >= cpr ? cpr : 0;
      const long e = e0 + u * stride;
      if (e >= pieces) continue;
      const T* aa = reinterpret_cast<const T*>(&av[u]);
      const T* bb = reinterpret_cast<const T*>(&bv[u]);
      uint4 ov;
      T* oo = reinterpret_cast<T*>(&ov);
#pragma unroll
      for (int k = 0; k < EPP; ++k) {
        float g = 0.f;
        if (!masked || c0 + k < C) {
          const float x = to_f(aa[k]);
          if (kind == 0) g = 2.f * (x - target) * s;
          else { const float dlt = x - to_f(bb[k]); g = dlt > 0.f ? s : (dlt < 0.f ? -s : 0.f); }
        }
        oo[k] = from_f<T>(g);
      }
      *reinterpret_cast<uint4*>(da + e * EPP) = ov;
    }
  }
}

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, long n, float lr, float b1, float b2, float eps,
                                                   float bc1, float bc2_sqrt, float gscale) {
  for (long e = ((long)blockIdx.x * 256 + threadIdx.x) * 4; e < n; e += (long)gridDim.x * 256 * 4) {
    if (e + 4 <= n) {
      float4 pp = *reinterpret_cast<float4*>(p + e);
      const float4 gg = *reinterpret_cast<const float4*>(g + e);
      float4 mm = *reinterpret_cast<float4*>(m + e);
      float4 vv = *reinterpret_cast<float4*>(v + e);
      float* pa = &pp.x; const float* ga = &gg.x; float* ma = &mm.x; float* va = &vv.x;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float gr = ga[k] * gscale;
        ma[k] = b1 * ma[k] + (1.f - b1) * gr;
        va[k] = b2 * va[k] + (1.f - b2) * gr * gr;
        pa[k] -= (lr / bc1) * ma[k] / (sqrtf(va[k]) / bc2_sqrt + eps);
      }
      *reinterpret_cast<float4*>(p + e) = pp;
      *reinterpret_cast<float4*>(m + e) = mm;
      *reinterpret_cast<float4*>(v + e) = vv;
    } else {
      for (long i = e; i < n; ++i) {
        const float gr = g[i] * gscale;
        m[i] = b1 * m[i] + (1.f - b1) * gr;
        v[i] = b2 * v[i] + (1.f - b2) * gr * gr;
        p[i] -= (lr / bc1) * m[i] / (sqrtf(v[i]) / bc2_sqrt + eps);
      }
    }
  }
}

inline int grid_for(long work, int cap = 2048) { return (int)std::max<long>(1, std::min<long>((work + 255) / 256, cap)); }


// Device-resident step counter and learning rate: the launch arguments no longer change from step to step, so a whole
// training step (this kernel included) can be captured once into a HIP graph and replayed.
__global__ __launch_bounds__(256) void adam_dev_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                       float* __restrict__ v, long n, const float* __restrict__ lr_dev,
                                                       const long long* __restrict__ step_dev, float b1, float b2, float eps,
                                                       float gscale, const float* __restrict__ scaler, int found_idx) {
  // loss scaling (fp16 storage): gradients arrive multiplied by scaler[0]; scaler[1] = 1 / scale; a non-finite gradient
  // anywhere in this buffer (scaler[3 + found_idx], set by grads_nonfinite_kernel) skips the whole update, as
  // torch.cuda.amp.GradScaler.step does (train.py:165-181)
  if (scaler != nullptr) {
    if (scaler[3 + found_idx] != 0.f) return;
    gscale *= scaler[1];
  }
  const double t = (double)(*step_dev + 1);
  const float lr = *lr_dev;
  const float bc1 = (float)(1.0 - pow((double)b1, t));
  const float bc2_sqrt = (float)sqrt(1.0 - pow((double)b2, t));
  for (long e = ((long)blockIdx.x * 256 + threadIdx.x) * 4; e < n; e += (long)gridDim.x * 256 * 4) {
    if (e + 4 <= n) {
      // the two moment buffers (and the gradient) are touched by nothing else in a step: streamed past the caches
      // (non-temporal), so that the weights stay resident for the next step's packs
      typedef __attribute__((ext_vector_type(4))) float f4v;
      float4 pp = *reinterpret_cast<float4*>(p + e);
      const f4v gq = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(g + e));
      const f4v mq = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(m + e));
      const f4v vq = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(v + e));
      const float4 gg = make_float4(gq.x, gq.y, gq.z, gq.w);
      float4 mm = make_float4(mq.x, mq.y, mq.z, mq.w);
      float4 vv = make_float4(vq.x, vq.y, vq.z, vq.w);
      float* pa = &pp.x; const float* ga = &gg.x; float* ma = &mm.x; float* va = &vv.x;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float gr = ga[k] * gscale;
        ma[k] = b1 * ma[k] + (1.f - b1) * gr;
        va[k] = b2 * va[k] + (1.f - b2) * gr * gr;
        pa[k] -= (lr / bc1) * ma[k] / (sqrtf(va[k]) / bc2_sqrt + eps);
      }
      *reinterpret_cast<float4*>(p + e) = pp;
      {
        const f4v mo = {mm.x, mm.y, mm.z, mm.w}, vo = {vv.x, vv.y, vv.z, vv.w};
        __builtin_nontemporal_store(mo, reinterpret_cast<f4v*>(m + e));
        __builtin_nontemporal_store(vo, reinterpret_cast<f4v*>(v + e));
      }
    } else {
      for (long i = e; i < n; ++i) {
        const float gr = g[i] * gscale;
        m[i] = b1 * m[i] + (1.f - b1) * gr;
        v[i] = b2 * v[i] + (1.f - b2) * gr * gr;
        p[i] -= (lr / bc1) * m[i] / (sqrtf(v[i]) / bc2_sqrt + eps);
      }
    }
  }
}

__global__ void adam_tick_kernel(long long* step_dev, const float* scaler, int found_idx) {
  if (scaler == nullptr || scaler[3 + found_idx] == 0.f) *step_dev += 1;          // a skipped update does not count
}

// flag[0] = 1 if any of g[0..n) is inf or nan (every thread that sees one stores the same value: no atomics)
__global__ __launch_bounds__(256) void grads_nonfinite_kernel(const float* __restrict__ g, long n, float* __restrict__ flag) {
  typedef __attribute__((ext_vector_type(4))) float f4v;
  bool bad = false;
  const long n4 = n >> 2, stride = (long)gridDim.x * 256;
  for (long e0 = (long)blockIdx.x * 256 + threadIdx.x; e0 < n4; e0 += 4 * stride) {
    f4v q[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) q[u] = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(g) + min(e0 + u * stride, n4 - 1));
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int k = 0; k < 4; ++k) bad |= !(fabsf(q[u][k]) <= 3.4e38f);             // false for inf and for nan
  }
  for (long e = (n4 << 2) + (long)blockIdx.x * 256 + threadIdx.x; e < n; e += stride) bad |= !(fabsf(g[e]) <= 3.4e38f);
  if (bad) *flag = 1.f;
}

// GradScaler.update(): state = (scale, 1 / scale, growth tracker, found_0, found_1, ...)
__global__ void scaler_update_kernel(float* state, float growth, float backoff, float interval) {
  const bool found = state[3] != 0.f || state[4] != 0.f;
  float scale = state[0], tracker = state[2];
  if (found) { scale *= backoff; tracker = 0.f; }
  else {
    tracker += 1.f;
    if (tracker >= interval) { scale *= growth; tracker = 0.f; }
  }
  state[0] = scale; state[1] = 1.f / scale; state[2] = tracker; state[3] = 0.f; state[4] = 0.f;
}

// base[off .. off + len) = 0 for every (off, len) pair: one launch for all the small segments of a flat buffer
__global__ __launch_bounds__(256) void zero_segments_kernel(float* __restrict__ base, const long long* __restrict__ seg, int n) {
  const int s = blockIdx.x;
  if (s >= n) return;
  const long long off = seg[2 * s], len = seg[2 * s + 1];
  for (long long i = threadIdx.x; i < len; i += 256) base[off + i] = 0.f;
}

}  // namespace

extern "C" int p2phd_loss_fwd(int kind, int dtype, const void* a, const void* b, float target, int64_t P, int C,
                              float coeff, float* out, void* stream) {
  P2PHD_REQUIRE(kind == 0 || kind == 1, "loss: kind must be 0 (mse vs constant) or 1 (l1)");
  P2PHD_REQUIRE(a && out && (kind == 0 || b) && C >= 1, "loss_fwd: bad arguments");
  if (P == 0) return P2PHD_OK;
  const int Cp = (C + 7) & ~7;
  hipStream_t st = (hipStream_t)stream;
  const p2phd::FoldScratch fs = p2phd::fold_scratch(p2phd::FOLD_LOSS, st);
  if (fs.part == nullptr) return P2PHD_EINVAL;                   // (refused: error text set by fold_scratch)
  if (dtype == P2PHD_BF16)
    hipLaunchKernelGGL(loss_fwd_kernel<bf16_t>, dim3(grid_for(P * Cp / 8, 1024)), dim3(256), 0, st, kind, (const bf16_t*)a, (const bf16_t*)b, target, (long)P, C, Cp, coeff, out, fs.part, fs.ticket);
  else if (dtype == P2PHD_F32)
    hipLaunchKernelGGL(loss_fwd_kernel<float>, dim3(grid_for(P * Cp / 8, 1024)), dim3(256), 0, st, kind, (const float*)a, (const float*)b, target, (long)P, C, Cp, coeff, out, fs.part, fs.ticket);
  else { p2phd::set_error("loss_fwd: unsupported dtype %d", dtype); return P2PHD_EUNSUPPORTED; }
  return p2phd::check_launch("loss_fwd");
}

extern "C" int p2phd_loss_bwd(int kind, int dtype, const void* a, const void* b, float target, int64_t P, int C,
                              float coeff, const float* grad_out, void* da, void* stream) {
  P2PHD_REQUIRE(kind == 0 || kind == 1, "loss: kind must be 0 (mse vs constant) or 1 (l1)");
  P2PHD_REQUIRE(a && da && grad_out && (kind == 0 || b) && C >= 1, "loss_bwd: bad arguments");
  if (P == 0) return P2PHD_OK;
  const int Cp = (C + 7) & ~7;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == P2PHD_BF16)
    hipLaunchKernelGGL(loss_bwd_kernel<bf16_t>, dim3(grid_for(P * Cp / 8)), dim3(256), 0, st, kind, (const bf16_t*)a, (const bf16_t*)b, target, (long)P, C, Cp, coeff, grad_out, (bf16_t*)da);
  else if (dtype == P2PHD_F32)
    hipLaunchKernelGGL(loss_bwd_kernel<float>, dim3(grid_for(P * Cp / 8)), dim3(256), 0, st, kind, (const float*)a, (const float*)b, target, (long)P, C, Cp, coeff, grad_out, (float*)da);
  else { p2phd::set_error("loss_bwd: unsupported dtype %d", dtype); return P2PHD_EUNSUPPORTED; }
  return p2phd::check_launch("loss_bwd");
}

extern "C" int p2phd_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                               float beta1, float beta2, float eps, int64_t step, float grad_scale, void* stream) {
  P2PHD_REQUIRE(step >= 1 && n >= 0, "adam: step counts from 1");
  if (n == 0) return P2PHD_OK;
  P2PHD_REQUIRE(params && grads && exp_avg && exp_avg_sq, "adam: null pointer");
  P2PHD_REQUIRE(((uintptr_t)params | (uintptr_t)grads | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) % 16 == 0, "adam: buffers must be 16-byte aligned");
  const double bc1 = 1.0 - std::pow((double)beta1, (double)step);
  const double bc2 = 1.0 - std::pow((double)beta2, (double)step);
  hipLaunchKernelGGL(adam_kernel, dim3(grid_for((n + 3) / 4, 4096)), dim3(256), 0, (hipStream_t)stream, params, grads, exp_avg,
                     exp_avg_sq, (long)n, lr, beta1, beta2, eps, (float)bc1, (float)std::sqrt(bc2), grad_scale);
  return p2phd::check_launch("adam_step");
}

static int adam_step_dev_impl(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, const float* lr_dev,
                              int64_t* step_dev, float beta1, float beta2, float eps, float grad_scale, float* scaler, int found_idx,
                              void* stream) {
  P2PHD_REQUIRE(n >= 0, "adam_step_dev: negative size");
  P2PHD_REQUIRE(lr_dev && step_dev, "adam_step_dev: null state pointer");
  hipStream_t st = (hipStream_t)stream;
  if (n > 0) {
    P2PHD_REQUIRE(params && grads && exp_avg && exp_avg_sq, "adam_step_dev: null pointer");
    if (scaler != nullptr)
      hipLaunchKernelGGL(grads_nonfinite_kernel, dim3(grid_for((n + 3) / 4, 2048)), dim3(256), 0, st, grads, (long)n, scaler + 3 + found_idx);
    const dim3 grid(grid_for((n + 3) / 4, 4096));
    hipLaunchKernelGGL(adam_dev_kernel, grid, dim3(256), 0, st, params, grads, exp_avg, exp_avg_sq, (long)n, lr_dev,
                       reinterpret_cast<const long long*>(step_dev), beta1, beta2, eps, grad_scale, (const float*)scaler, found_idx);
  }
  hipLaunchKernelGGL(adam_tick_kernel, dim3(1), dim3(1), 0, st, reinterpret_cast<long long*>(step_dev), (const float*)scaler, found_idx);
  return p2phd::check_launch("adam_step_dev");
}

extern "C" int p2phd_adam_step_dev(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n,
                                   const float* lr_dev, int64_t* step_dev, float beta1, float beta2, float eps,
                                   float grad_scale, void* stream) {
  return adam_step_dev_impl(params, grads, exp_avg, exp_avg_sq, n, lr_dev, step_dev, beta1, beta2, eps, grad_scale, nullptr, 0, stream);
}

extern "C" int p2phd_adam_step_scaled(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n,
                                      const float* lr_dev, int64_t* step_dev, float beta1, float beta2, float eps,
                                      float grad_scale, float* scaler_state, int found_index, void* stream) {
  P2PHD_REQUIRE(scaler_state != nullptr && (found_index == 0 || found_index == 1), "adam_step_scaled: bad scaler arguments");
  return adam_step_dev_impl(params, grads, exp_avg, exp_avg_sq, n, lr_dev, step_dev, beta1, beta2, eps, grad_scale, scaler_state,
                            found_index, stream);
}

extern "C" int p2phd_scaler_update(float* scaler_state, float growth_factor, float backoff_factor, int growth_interval, void* stream) {
  P2PHD_REQUIRE(scaler_state != nullptr && growth_factor >= 1.f && backoff_factor > 0.f && backoff_factor <= 1.f && growth_interval >= 1,
                "scaler_update: bad arguments");
  hipLaunchKernelGGL(scaler_update_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, scaler_state, growth_factor, backoff_factor,
                     (float)growth_interval);
  return p2phd::check_launch("scaler_update");
}

extern "C" int p2phd_zero_segments(float* base, const int64_t* seg_dev, int n, void* stream) {
  if (n <= 0) return P2PHD_OK;
  P2PHD_REQUIRE(base && seg_dev, "zero_segments: null pointer");
  hipLaunchKernelGGL(zero_segments_kernel, dim3((unsigned)n), dim3(256), 0, (hipStream_t)stream, base, reinterpret_cast<const long long*>(seg_dev), n);
  return p2phd::check_launch("zero_segments");
}
