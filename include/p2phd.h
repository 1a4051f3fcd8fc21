/*
 * p2phd.h -- C ABI of libp2phd_hip.so: the MI355X (gfx950) hot path of pix2pixHD audio
 * super-resolution (batched MDCT4/IMDCT4 + generator/discriminator conv stack).
 *
 * The reference (ishine/pix2pixHDAudioSR) has no FFI on this path: its boundary is the Python
 * module API (models/mdct.py, models/networks.py, models/pix2pixHD_model.py).  Every entry
 * point below names the reference code it replaces (file:line relative to the reference
 * checkout); the Python mirror of those modules in pix2pixhdaudiosr_amd/ binds these symbols
 * with ctypes (see INTEGRATION.md for the stub a reference maintainer would add).
 *
 * Conventions
 *   - all data pointers are DEVICE pointers, caller-allocated (torch tensors own them);
 *   - `stream` is a hipStream_t passed as void* (the caller's current stream); no entry point
 *     allocates, frees or synchronises, so calls are graph-capturable;
 *   - return 0 on success, a negative P2PHD_E* code otherwise; p2phd_last_error() gives text
 *     (thread-local);
 *   - thread-compatible, not thread-safe per output buffer.
 *
 * Streams
 *   The kernels that reduce across workgroups without float atomics (InstanceNorm backward sums, bias column sums, loss
 *   accumulators, the split-K tail of the conv GEMM) keep their partial rows and arrival tickets in a scratch that belongs
 *   to the library, one region per kernel family.  A region serves ONE stream at a time: every such launch records an event
 *   behind itself, and a call of the same family that arrives on another stream first makes that stream wait for the event
 *   (hipStreamWaitEvent) -- the launches are ordered on the device, no partials mix, nothing is refused and no stream handle
 *   is ever queried after its owner may have destroyed it.  Launches recorded into a graph under capture run in graph order;
 *   replaying two graphs that use one family concurrently on two streams is the caller's to order.  p2phd_reduction_reset(stream)
 *   re-arms the tickets; call it once per training step (a faulted or aborted launch could otherwise leave one armed wrong).
 */
#ifndef P2PHD_H
#define P2PHD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define P2PHD_OK            0
#define P2PHD_EINVAL       -1   /* bad argument / unsupported geometry */
#define P2PHD_ELAUNCH      -2   /* HIP launch error */
#define P2PHD_EUNSUPPORTED -3   /* valid request this build does not implement */

/* element types of activation / weight buffers.  P2PHD_BF16 names THE LIBRARY'S 16-BIT STORAGE TYPE: bf16 in libp2phd_hip.so
 * (the benchmarked mode), IEEE fp16 in libp2phd_hip_f16.so -- the same sources built with -DP2PHD_F16 (the reference's actual AMP
 * type: train.py:62-67 runs torch.cuda.amp.autocast, i.e. fp16 activations with fp32 accumulation and a GradScaler).  Same entry
 * points, same layouts; p2phd_half_type() returns 1 (bf16) or 2 (fp16). */
#define P2PHD_F32  0
#define P2PHD_BF16 1
int p2phd_half_type(void);

const char* p2phd_last_error(void);
int p2phd_abi_version(void);
/* fills name (<= cap bytes) with the device's gcnArchName; returns CU count or <0 */
int p2phd_device_info(char* name, int cap);
/* Per-library options (process-wide; tests, A/B timing, and one that the data-parallel step sets).  Unknown name or value: P2PHD_EINVAL.
 *   "cus" n            CUs a conv launch may count on when the step runs on a CU-masked stream (0 = all of the device's)
 *   "gconv_bm" v       gather-GEMM tile: 0 heuristic | 128 | 192 | 256 | 258 (256 rows on the 2-slot ring) | 512 (256 x 256)
 *   "gconv_halo" 0|1   the HALO main loop of 3x3 stride-1 layers on 16-wide planes (1) or the generic loop (0)
 *   "tile128x192" 0|1  the 128 x 192 tile of small planes with 192-divisible outputs
 *   "cls_skip" 0|1     tap-skipping merged stride-2 launches (changes the packed layout of those layers: see p2phd_conv_pack_layout)
 *   "splitk_tail" 0|1|2  split-K of the last, partly filled tile round: off | where the cost model says so | wherever possible
 *   "march", "dfirst", "dlast", "c7_generic", "reflect_generic", "mdct_generic" 0|1   the dedicated kernels of the outermost
 *                      stride-2 pair, the discriminator's first / last layer, the two 7x7 layers, the reflection extras, the fast MDCT:
 *                      on (default; c7_generic / reflect_generic / mdct_generic = 1 select the generic path instead)
 *   "wgrad_tm" 0|128, "wgrad_xcd" 0|1, "mdct_iters" 0..8, "c7_abl"   weight-gradient row tile / XCD-aware order, experiment knobs
 *   "cw_inject" 0|1    libp2phd_hip_chk.so only: selects round 4's too-lax HALO wait (sensitivity check of p2phd_wait_check) */
int p2phd_set_option(const char* name, int value);
/* zeroes the arrival tickets of the fixed-order reductions on `stream` (13 KB memset; see "Streams" above) */
int p2phd_reduction_reset(void* stream);
/* Measurement hook (bench.py's roofline): while armed, every launch of the implicit-GEMM conv kernel whose gathered
 * tensor has `cin_pitch` channels, whose GEMM-K is `kk` and whose pixel grid is hg x wg is bracketed by HIP events on its
 * own launch stream (the kernel alone: none of the companion launches of p2phd_conv_fwd).  p2phd_probe_read waits for
 * the recorded events and returns up to `cap` durations in milliseconds; arm with enable = 0 to stop. */
int p2phd_probe_gconv(int enable, int cin_pitch, int kk, int hg, int wg);
/* Same with two more filters: gather_pad_mode = the padding rule of the launch's gather (0 zeros, 1 reflect = the forward of
 * a ReflectionPad2d conv, 2 = the reflect adjoint of its input gradient; -1 = any) and elem_bytes = operand element size
 * (1 e4m3, 2 bf16, 4 f32; 0 = any) -- so a forward launch is told from the same-shaped input-gradient launch. */
int p2phd_probe_gconv_ex(int enable, int cin_pitch, int kk, int hg, int wg, int gather_pad_mode, int elem_bytes);
int p2phd_probe_read(float* ms_out, int cap);
/* Launch counters: how many launches of a kernel family the library has made since the last reset -- "gconv" (every
 * gather-GEMM launch), "halo" (those on the patch-staged 3x3 main loop), "cls_skip" (tap-skipping merged stride-2 launches),
 * "splitk" (launches with a split-K tail), "tile256" (256 x 256 tiles), "tile128x192" (128 x 192 tiles of small planes),
 * "march" / "march_w" (marching kernels), "wgrad" (MFMA weight gradient).  family == NULL with reset != 0 clears all.
 * Returns the count before the reset, -1 for an unknown name.  Counts launches recorded under graph capture once (at capture).
 * Test hook: proves which kernels a whole training step really runs on (train.py:148-184 at the benchmarked batch). */
int64_t p2phd_launch_count(const char* family, int reset);
/* Checker of the hand-counted LDS-DMA waits (round 5; libp2phd_hip_chk.so = the same sources with -DP2PHD_CHECK_WAITS): every
 * wave of the gather-GEMM loops (generic and HALO) and of the weight-gradient loops logs the LDS buffer each piece it issues
 * fills and, at every relaxed `s_waitcnt vmcnt(n)` in front of a slab barrier, checks that none of its n youngest pieces targets
 * a buffer that is read behind that barrier.  out4[0] = violating kernel families (bit 0 generic loop, 1 HALO, 2 weight gradient,
 * 3 its f32 form), [1] = waits checked, [2] = first offender, [3] = pieces logged.  Synchronises the device.  The product
 * build returns P2PHD_EUNSUPPORTED.  p2phd_set_option("cw_inject", 1) (check build only) runs the HALO loop with the too-lax
 * wait that shipped for 1.5 h in round 4: the checker must flag it (tests/test_gpu_waits.py). */
int p2phd_wait_check(unsigned* out4, int reset);

/* ------------------------------------------------------------------------------------------
 * MDCT4 / IMDCT4 (models/mdct.py:461-566).  n_fft a power of two in [16, 4096].
 *
 * Tables: p2phd_mdct4_tables_floats(n_fft) floats, filled on the HOST by
 * p2phd_mdct4_tables_fill (fp64 trigonometry rounded once to fp32), uploaded by the caller and
 * passed back as the device pointer `tables`.  They replace exp1/exp2 of mdct.py:483-484,539-540.
 * ---------------------------------------------------------------------------------------- */
size_t p2phd_mdct4_tables_floats(int n_fft);
int p2phd_mdct4_tables_fill(int n_fft, float* host_out);

/* Frame geometry exactly as MDCT4.forward computes it (mdct.py:488-500), including the
 * len(signal) quirk: dim0 is the size of the first dimension of the input (the batch size for a
 * [B,T] signal, T for a 1-D one).  Pure host integer arithmetic. */
int p2phd_mdct4_frame_layout(int64_t dim0, int64_t T, int hop, int win, int center,
                             int64_t* start_pad, int64_t* end_pad, int64_t* n_frames);

/* Framed transform: out[b,t,k] = scale * sum_n w[n] xpad[b, t*hop+n] cos(2pi/N (n+1/2+N/4)(k+1/2)),
 * xpad = x shifted right by start_pad with zeros outside [0,T).  x [B,T] f32, window [win] f32,
 * out [B,F,N/2] f32.  MDCT4.forward (mdct.py:486-513) = this with scale 1; the backward of
 * IMDCT4 = this with x = grad, start_pad = crop, scale 4/N. */
int p2phd_mdct4_fwd(const float* x, int64_t B, int64_t T, int n_fft, int hop, int win,
                    const float* window, const float* tables, int64_t start_pad, int64_t n_frames,
                    float scale, float* out, void* stream);

/* Inverse framed transform with windowed overlap-add:
 * out[b,m] = scale * sum_t w[q] y_t[q], q = m + crop_start - t*hop in [0,win),
 * y_t[n] = sum_k spec[b,t,k] cos(2pi/N (n+1/2+N/4)(k+1/2)).   spec [B,F,N/2] f32, out [B,out_len] f32.
 * IMDCT4.forward (mdct.py:542-566) = this with scale 4/N, crop_start = win/2 (center) and
 * out_len = min(out_length, (F-1)*hop [+win if !center]); the backward of MDCT4 = this with scale 1,
 * crop_start = start_pad, out_len = T. */
int p2phd_imdct4_fwd(const float* spec, int64_t B, int64_t n_frames, int n_fft, int hop, int win,
                     const float* window, const float* tables, int64_t crop_start, int64_t out_len,
                     float scale, float* out, void* stream);

/* ------------------------------------------------------------------------------------------
 * MDCT2 / IMDCT2 and the DCT-II / DCT-III operators DCT_2N_native / IDCT_2N_native
 * (models/mdct.py:352-454, dct/dct_native.py:7-68; native counterparts dct/src/dct_2N_cuda.cpp,
 * dct_cuda_kernel.cu:267-406).  n_fft a power of two in [16, 2048]; tables as for MDCT4.
 *   forward  out[b,t,k] = scale * c_k * (2/N) * sum_i w[i] xpad[b, t*hop+i] cos(pi (2i+1) k / 2N), c_0 = k0_scale
 *   inverse  y_t[i]     = k0_scale * S[b,t,0] + 2 * sum_{k>=1} S[b,t,k] cos(pi (2i+1) k / 2N);
 *            out[b,m]   = scale * sum_t w[q] y_t[q], q = m + crop_start - t*hop in [0,win)
 * MDCT2.forward = forward(scale 1, k0 1); IMDCT2.forward = inverse(scale 1/2, k0 1); a plain DCT_2N_native on rows
 * [R,N] = forward with B = R, T = N, hop = win = N, window of ones, one frame per row (IDCT likewise);
 * autograd adjoints: forward^T = inverse(scale s/N, k0 2*k0), inverse^T = forward(scale s*N, k0 k0/2).
 * ---------------------------------------------------------------------------------------- */
size_t p2phd_dct_tables_floats(int n_fft);
int p2phd_dct_tables_fill(int n_fft, float* host_out);
int p2phd_mdct2_fwd(const float* x, int64_t B, int64_t T, int n_fft, int hop, int win, const float* window,
                    const float* tables, int64_t start_pad, int64_t n_frames, float scale, float k0_scale,
                    float* out, void* stream);
int p2phd_imdct2_fwd(const float* spec, int64_t B, int64_t n_frames, int n_fft, int hop, int win, const float* window,
                     const float* tables, int64_t crop_start, int64_t out_len, float scale, float k0_scale,
                     float* out, void* stream);

/* ------------------------------------------------------------------------------------------
 * Activation tensors of the conv stack are NHWC ("channels last": [N, H, W, Cp]) with the channel
 * pitch Cp = p2phd_channel_pitch(C) = C rounded up to 8; pad channels hold zeros.  dtype is
 * P2PHD_F32 (exact-f32 MFMA, parity runs) or P2PHD_BF16 (bf16 MFMA, fp32 accumulate).
 * ---------------------------------------------------------------------------------------- */
int p2phd_channel_pitch(int channels);

#define P2PHD_ACT_NONE  0
#define P2PHD_ACT_LRELU 1   /* LeakyReLU(0.2), models/networks.py:342,350,358 */
#define P2PHD_ACT_TANH  2   /* models/networks.py:160,207 */
#define P2PHD_ACT_RELU  3

/* One Conv2d / ConvTranspose2d layer of models/networks.py.
 *   transposed = 0: nn.Conv2d(C, K, (R,S), stride, padding=pad); with pad_mode = 1 the layer is
 *                   nn.ReflectionPad2d(pad) followed by nn.Conv2d(..., padding=0) (networks.py:190,223-231);
 *   transposed = 1: nn.ConvTranspose2d(C, K, (R,S), stride, padding=pad, output_padding=opad) (:205).
 * Master weights keep the PyTorch layouts ([K,C,R,S], resp. [C,K,R,S]) in f32. */
typedef struct p2phd_conv_desc {
  int32_t N, C, H, W;
  int32_t K, R, S;
  int32_t stride, pad, pad_mode, transposed, opad;
  int32_t dtype;
  int32_t w_layout;   /* layout of the f32 master weights (and of dw): 0 = PyTorch ([K,C,R,S] / [C,K,R,S]), 1 = K-major [K][R][S][C] */
} p2phd_conv_desc;

/* K-major master weights (round 3): for plain stride-1 Conv2d layers without channel or K padding (the residual trunk, the
 * discriminator's 256 -> 512 layers: 95 % of the parameters) the packed forward row [tap][channel] IS the master row when the
 * optimiser keeps the weights as [K][R][S][C].  The forward pack is then a cast, the input-gradient pack a bf16 transpose per
 * tap, and the weight gradient lands with coalesced rows instead of through a re-ordering tile.  The Python mirror hands such
 * parameters to torch as permuted views of the flat buffer (state_dict and checkpoints are unchanged).
 * p2phd_conv_kmajor_ok: 1 if desc (w_layout ignored) may set w_layout = 1. */
int p2phd_conv_kmajor_ok(const p2phd_conv_desc* c);
int p2phd_conv_out_size(const p2phd_conv_desc* c, int* Ho, int* Wo);

/* Packed (K-contiguous, tap-major, zero-padded) weights for the forward (which = 0) or the input-gradient
 * (which = 1) launches; repack after every optimizer step. */
size_t p2phd_conv_packed_bytes(const p2phd_conv_desc* c, int which);
int p2phd_conv_pack_weights(const p2phd_conv_desc* c, int which, const float* w, void* packed, void* stream);
/* Variant id (>= 0; -1: bad descriptor) of the packed layout that the launches of (desc, which) read.  The layout depends on
 * more than the layer: a tap-skipping merged stride-2 launch orders the taps of half its sub-pixel classes differently, and
 * whether a launch skips depends on N, H, W and the "cls_skip" option.  A buffer packed for one desc serves another desc of
 * the same layer only if both report the same id -- a caller that caches packed weights keys the cache on it
 * (the reference runs inference on a smaller last batch between training steps: train.py:206 -> eval_model). */
int p2phd_conv_pack_layout(const p2phd_conv_desc* c, int which);

/* y = act(conv(x) + bias).  If stats != NULL (float [N][Cp_out][2], overwritten; act must be NONE) it receives, per
 * (n, channel), the MEAN and the SUM OF SQUARED DEVIATIONS from it of conv(x)+bias over the sample's plane -- what
 * InstanceNorm2d needs (networks.py:22).  Every wave of the conv epilogue stores the partial of its rows about its own
 * mean (plain stores into a table in `workspace`), a small kernel merges them in a fixed order (Chan et al.): no float
 * atomics (bit-reproducible) and no E[x^2] - E[x]^2 cancellation. */
/* scratch of the forward launch: the folded tensor of <= 4-channel layers and the per-wave statistics partials */
size_t p2phd_conv_fwd_workspace_bytes(const p2phd_conv_desc* c);
int p2phd_conv_fwd(const p2phd_conv_desc* c, const void* x, const void* packed_fwd, const float* bias, int act,
                   void* y, float* stats, void* workspace, void* stream);

/* fp8 forward of the wide stride-1 layers (BASELINE configs[4]: bf16 + fp8 MFMA conv weights; the reference analogue is its
 * AMP path, train.py:62-67,148-149).  Operands are OCP e4m3 on the block-scaled v_mfma_scale_f32_32x32x64_f8f6f4 (unit e8m0 block scales: the layer's one
 * scale is applied in the epilogue; twice the bf16 MFMA rate), accumulation fp32, outputs bf16;
 * fp32 master weights, the bf16 activations and the whole backward pass are unchanged.
 *   p2phd_conv_fp8_eligible   : 1 if the layer can run this way (Conv2d, stride 1, C %% 16 == 0, R*S*C %% 128 == 0, desc.dtype BF16)
 *   p2phd_conv_fp8_pack_weights: per-layer scale = max|w| / 448 found on the device (no host sync), weights quantised into
 *                               `packed8` (p2phd_conv_fp8_packed_bytes); repack after every optimizer step
 *   p2phd_conv_fwd_fp8        : x8 = the e4m3 twin of the bf16 input written by p2phd_instnorm_act_fwd_q8 (scale 1);
 *                               y, stats, workspace as for p2phd_conv_fwd */
int p2phd_conv_fp8_eligible(const p2phd_conv_desc* c);
size_t p2phd_conv_fp8_packed_bytes(const p2phd_conv_desc* c);
int p2phd_conv_fp8_pack_weights(const p2phd_conv_desc* c, const float* w, void* packed8, void* stream);
int p2phd_conv_fwd_fp8(const p2phd_conv_desc* c, const void* x8, const void* packed8, const float* bias, int act,
                       void* y, float* stats, void* workspace, void* stream);

/* dx = conv^T(dy) (+ addend, same layout as dx).  Replaces autograd of F.conv2d / F.conv_transpose2d and, for
 * pad_mode = 1, of ReflectionPad2d as well (needs p2phd_conv_dgrad_workspace_bytes of scratch). */
size_t p2phd_conv_dgrad_workspace_bytes(const p2phd_conv_desc* c);
int p2phd_conv_dgrad(const p2phd_conv_desc* c, const void* dy, const void* packed_dgrad, const void* addend, void* dx,
                     void* workspace, void* stream);

/* Input gradient with the FIRST pass of the producer's InstanceNorm backward fused into its store loop (round 2).  dx is
 * the gradient of the tensor this conv read, which a previous layer produced as act(InstanceNorm(prev_y)); prev_y
 * [N,H,W,Cp(C)] are that layer's pre-normalisation values, prev_stats [N,Cp(C),2] its (mean, M2) statistics, prev_act its
 * activation (NONE / RELU / LRELU).  Besides dx (+ addend) the call leaves bstats [N,Cp(C),2] = per (n, c)
 * (sum g', sum g' * yhat), g' = dx * act'(yhat) -- exactly what p2phd_instnorm_act_bwd's reduce pass computes from (dx,
 * prev_y) in two more tensor reads -- summed in a fixed order (no atomics).  Feed it to p2phd_instnorm_act_bwd_apply.
 * Available when p2phd_conv_dgrad_bsum_ok(desc) (one direct gather-GEMM launch: no reflect padding, no W-fold, not the
 * dedicated 7x7 kernel); workspace: p2phd_conv_dgrad_bsum_workspace_bytes (this call does not need the plain dgrad workspace). */
int p2phd_conv_dgrad_bsum_ok(const p2phd_conv_desc* c);
/* advisory: 0 where the fused form is available but slower than p2phd_conv_dgrad + the two-pass backward (the plain input
 * gradient would run on the 256 x 256 tile, which has no fused store loop: the discriminator's 256 -> 512 layer) */
int p2phd_conv_dgrad_bsum_pays(const p2phd_conv_desc* c);
size_t p2phd_conv_dgrad_bsum_workspace_bytes(const p2phd_conv_desc* c);
int p2phd_conv_dgrad_bsum(const p2phd_conv_desc* c, const void* dy, const void* packed, const void* addend, void* dx,
                          const void* prev_y, const float* prev_stats, int prev_act, float eps, float* bstats,
                          void* workspace, void* stream);
/* The same store loop for a producer WITHOUT InstanceNorm (Conv + ReLU / LeakyReLU(0.2)): x_act is that block's output --
 * the tensor this conv read -- and dx leaves multiplied by act'(x_act), i.e. as the gradient of the producer's
 * pre-activation; its activation-backward pass (p2phd_act_bwd / p2phd_act_bwd_db) is then not needed.  Same availability
 * and workspace as p2phd_conv_dgrad_bsum. */
int p2phd_conv_dgrad_act(const p2phd_conv_desc* c, const void* dy, const void* packed, const void* addend, void* dx,
                         const void* x_act, int prev_act, void* workspace, void* stream);

/* dw (master layout, f32, overwritten) and db (f32 [K], overwritten, may be NULL) from x and dy. */
size_t p2phd_conv_wgrad_workspace_bytes(const p2phd_conv_desc* c);
int p2phd_conv_wgrad(const p2phd_conv_desc* c, const void* x, const void* dy, float* dw, float* db, void* workspace,
                     void* stream);
/* Same, but dw += and db += : the gradient lands straight in the optimiser's (zeroed) flat gradient buffer, which
 * replaces one temporary and one `grad += new` launch per parameter of autograd's AccumulateGrad. */
int p2phd_conv_wgrad_acc(const p2phd_conv_desc* c, const void* x, const void* dy, float* dw, float* db, void* workspace,
                         void* stream);

/* ------------------------------------------------------------------------------------------
 * HBM-bound companions (csrc/norm.hip).  NHWC tensors, channel pitch = p2phd_channel_pitch(C).
 * ---------------------------------------------------------------------------------------- */

/* out = act((y - mean) * rstd) + residual: InstanceNorm2d(affine=False, eps) (networks.py:22) + ReLU /
 * LeakyReLU(0.2) / none, + the ResnetBlock skip (networks.py:252) or the LocalEnhancer sum (:180) when
 * residual != NULL.  stats = the float [N][Cp][2] (mean, sum of squared deviations) p2phd_conv_fwd wrote. */
int p2phd_instnorm_act_fwd(int dtype, const void* y, const float* stats, const void* residual, void* out,
                           int N, int64_t HW, int C, float eps, int act, void* stream);
/* Same, and additionally out8 = out as OCP e4m3 bytes (scale 1, saturated at +-448; dtype must be BF16): the operand of
 * the next layer's p2phd_conv_fwd_fp8. */
int p2phd_instnorm_act_fwd_q8(int dtype, const void* y, const float* stats, const void* residual, void* out, void* out8,
                              int N, int64_t HW, int C, float eps, int act, void* stream);
/* dy from g = dL/d(out) through act and InstanceNorm; bstats: float [N][Cp][2] scratch (zeroed inside).
 * db (float [C], may be NULL): the conv bias gradient = column sums of dy, accumulated in the same pass. */
int p2phd_instnorm_act_bwd(int dtype, const void* g, const void* y, const float* stats, float* bstats, void* dy,
                           float* db, int N, int64_t HW, int C, float eps, int act, void* stream);
/* Same, with db += instead of db = (see p2phd_conv_wgrad_acc). */
int p2phd_instnorm_act_bwd_acc(int dtype, const void* g, const void* y, const float* stats, float* bstats, void* dy,
                               float* db, int N, int64_t HW, int C, float eps, int act, void* stream);
/* The second (apply) pass alone, with the sums already in bstats (p2phd_conv_dgrad_bsum); db_accumulate != 0 adds the bias
 * gradient into db instead of overwriting it.  Only for planes that take the two-pass form:
 * p2phd_instnorm_act_bwd_two_pass(dtype, N, HW, C) != 0 (small planes use a single register-resident launch). */
int p2phd_instnorm_act_bwd_two_pass(int dtype, int N, int64_t HW, int C);
/* Residual trunk (networks.py:231-252: ReflectionPad2d(1) + Conv2d 3x3 + InstanceNorm): the input gradient of such a conv reads
 * dy plus the pair-sum rows / columns of the reflection's adjoint.  For planes that take the single-launch InstanceNorm backward
 * the kernel that WRITES dy appends them: allocate dy with p2phd_conv_reflect_extras_elems(desc) extra elements right behind its
 * N*H*W*Cp (0 = this layer / plane has no such form), hand the extras pointer (= dy + N*H*W*Cp) to p2phd_instnorm_act_bwd_rx
 * (the single-launch backward: no bstats), and take the input gradient with p2phd_conv_dgrad_rx -- no expansion pass, no workspace.
 * db / db_accumulate as in p2phd_instnorm_act_bwd_apply. */
size_t p2phd_conv_reflect_extras_elems(const p2phd_conv_desc* c);
int p2phd_instnorm_act_bwd_rx(int dtype, const void* g, const void* y, const float* stats, void* dy, float* db, int db_accumulate,
                              int N, int H, int W, int C, float eps, int act, void* reflect_extras, void* stream);
int p2phd_conv_dgrad_rx(const p2phd_conv_desc* c, const void* dy_with_extras, const void* packed_dgrad, const void* addend, void* dx,
                        void* stream);
int p2phd_instnorm_act_bwd_apply(int dtype, const void* g, const void* y, const float* stats, const float* bstats, void* dy,
                                 float* db, int db_accumulate, int N, int64_t HW, int C, float eps, int act, void* stream);
/* dx = g * act'(.) evaluated from the saved activation OUTPUT a (tanh, LeakyReLU, ReLU). */
int p2phd_act_bwd(int dtype, const void* g, const void* a, void* dx, int64_t n_elems, int act, void* stream);
/* Same over [n_pixels][Cp] tensors, plus db[c] (+)= sum over pixels of dx[., c]: the conv bias gradient of a layer with a
 * fused activation and no norm (networks.py:342-343), without a second read of dx. */
int p2phd_act_bwd_db(int dtype, const void* g, const void* a, void* dx, int64_t n_pixels, int C, int act, float* db,
                     int db_accumulate, void* stream);

/* nn.AvgPool2d(3, stride=2, padding=[1,1], count_include_pad=False) (networks.py:165,308). */
int p2phd_avgpool3s2_fwd(int dtype, const void* x, void* y, int N, int H, int W, int C, void* stream);
int p2phd_avgpool3s2_bwd(int dtype, const void* dy, void* dx, int N, int H, int W, int C, void* stream);

/* Module-boundary layout converters: f32 NCHW [N,C,HW] <-> channels [ch_off, ch_off+C) of NHWC [N,HW,Cp]. */
int p2phd_nchw_to_nhwc(int dtype, const float* src, void* dst, int N, int C, int64_t HW, int Cp, int ch_off, void* stream);
int p2phd_nhwc_to_nchw(int dtype, const void* src, float* dst, int N, int C, int64_t HW, int Cp, int ch_off, void* stream);
/* torch.cat((x0, x1, ...), dim=1) of up to four f32 NCHW tensors (pix2pixHD_model.py:56,307,360: the discriminator's
 * input) straight into NHWC [N,HW,Cp]: `srcs` / `chans` are HOST arrays of nsrc device pointers / channel counts; every
 * 16-byte piece of dst is written once, pad channels as zeros (no memset of dst, one launch). */
int p2phd_nchw_cat_to_nhwc(int dtype, const float* const* srcs, const int* chans, int nsrc, void* dst, int N, int64_t HW, int Cp,
                           void* stream);

/* ------------------------------------------------------------------------------------------
 * Losses and optimiser (csrc/loss.hip).
 * kind 0: mean((a - target)^2)  -- GANLoss with use_lsgan (networks.py:68-110)
 * kind 1: mean(|a - b|)         -- criterionFeat = L1Loss (pix2pixHD_model.py:99,391-398)
 * a, b: [P][Cp] activations with C valid channels; *out += coeff * mean; the backward reads the upstream
 * gradient from device memory (*grad_out) so no host synchronisation is needed.
 * ---------------------------------------------------------------------------------------- */
int p2phd_loss_fwd(int kind, int dtype, const void* a, const void* b, float target, int64_t P, int C, float coeff,
                   float* out, void* stream);
int p2phd_loss_bwd(int kind, int dtype, const void* a, const void* b, float target, int64_t P, int C, float coeff,
                   const float* grad_out, void* da, void* stream);
/* torch.optim.Adam (no amsgrad / weight decay) over one flat f32 buffer; grads are scaled by grad_scale first
 * (1/world_size after a summing all-reduce).  step counts from 1. */
int p2phd_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                    float beta1, float beta2, float eps, int64_t step, float grad_scale, void* stream);
/* Same update with the learning rate and the step counter read from DEVICE memory (lr_dev: one float; step_dev: one
 * int64 holding the number of steps taken so far, incremented on the stream after the update): every launch argument
 * is then constant from step to step and the whole training step can be captured into a HIP graph and replayed. */
int p2phd_adam_step_dev(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, const float* lr_dev,
                        int64_t* step_dev, float beta1, float beta2, float eps, float grad_scale, void* stream);

/* Lazily normalised input (round 4; networks.py:190-195 and :205-207: the generator's outermost stride-2 layers at ngf 48, bf16).
 * p2phd_conv_lazy_ok(desc) = 1: this layer's forward and weight gradient can take `x_raw`, the PRE-normalisation output of the
 * InstanceNorm block in front of it, with that block's statistics [N][Cp][2] (mean, sum of squared deviations), activation and
 * eps, and apply (x - mean) * rstd + activation while staging rows -- the p2phd_instnorm_act_fwd pass over that plane is not run.
 * Results equal those computed from the materialised tensor, bit for bit.  Other arguments as p2phd_conv_fwd / p2phd_conv_wgrad
 * (`accumulate`: 0 = overwrite dw / db, 1 = add). */
int p2phd_conv_lazy_ok(const p2phd_conv_desc* c);
int p2phd_conv_fwd_lazy(const p2phd_conv_desc* c, const void* x_raw, const float* x_stats, int x_act, float x_eps,
                        const void* packed_fwd, const float* bias, void* y, float* stats, void* workspace, void* stream);
int p2phd_conv_wgrad_lazy(const p2phd_conv_desc* c, const void* x_raw, const float* x_stats, int x_act, float x_eps,
                          const void* dy, float* dw, float* db, int accumulate, void* workspace, void* stream);

/* Loss scaling for fp16 storage (torch.cuda.amp.GradScaler of train.py:62-67,165-181, device-resident so that a captured step
 * replays): scaler_state = 8 floats in device memory: [0] scale, [1] 1 / scale, [2] growth tracker, [3], [4] non-finite flags of
 * two gradient buffers (generator, discriminator).  The caller multiplies the loss by state[0] before its backward pass;
 * p2phd_adam_step_scaled = p2phd_adam_step_dev that first scans `grads` for inf / nan (flag 3 + found_index), unscales by
 * state[1] and SKIPS the update (and the step count) when the flag is set; p2phd_scaler_update = GradScaler.update(): backoff on
 * a flag, growth after growth_interval clean steps, flags cleared. */
int p2phd_adam_step_scaled(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, const float* lr_dev,
                           int64_t* step_dev, float beta1, float beta2, float eps, float grad_scale, float* scaler_state,
                           int found_index, void* stream);
int p2phd_scaler_update(float* scaler_state, float growth_factor, float backoff_factor, int growth_interval, void* stream);

/* base[off, off + len) = 0 for n (off, len) pairs of int64 in DEVICE memory: one launch for the small segments of a flat
 * gradient buffer (the bias gradients, which several kernels add into; the weight gradients are overwritten by their first
 * writer of the step: p2phd_conv_wgrad vs p2phd_conv_wgrad_acc). */
int p2phd_zero_segments(float* base, const int64_t* seg_dev, int n, void* stream);

/* ------------------------------------------------------------------------------------------
 * Spectrogram codec (csrc/spectro.hip): Pix2PixHDModel.to_spectro / denormalize / to_audio with
 * explicit_encoding (pix2pixHD_model.py:142-249).
 * encode: spec [B,F,M] f32 (MDCT4 output) -> log_spectro [B,2,M,F] in [0,1], pha [B,1,M,F],
 *   norm8 = (min, max, mean, std, noise_min, noise_max, -, -) on the device.  The top mask_rows bins are
 *   replaced by min-max scaled `noise` [B,2,mask_rows,F] (mask_mode 'mode2'), or zeros if noise == NULL.
 *   partials: scratch of p2phd_spectro_partials_floats(B,F,M) floats.
 * decode: log_spectro [B,2,M,F] + (min,max) -> spec [B,F,M] ready for IMDCT4.
 * ---------------------------------------------------------------------------------------- */
int64_t p2phd_spectro_partials_floats(int64_t B, int64_t F, int64_t M);
int p2phd_spectro_encode(const float* spec, int64_t B, int64_t F, int64_t M, float alpha, float min_value,
                         int mask_rows, const float* noise, float* log_spectro, float* pha, float* norm8,
                         float* partials, void* stream);
/* The general form (pix2pixHD_model.py:149-162,196-226).  channels = 2: explicit encoding as above; channels = 1:
 * log_spectro [B,1,M,F] = amplitude_to_DB(|spec| + min_value) min-max scaled, the sign in pha.  mask_mode 0 / 1 / 2 =
 * the reference's 'mode0' (noise / (max - min)), 'mode1' (min-max scaled noise times noise_sign, a +-1 tensor shaped
 * like noise) and 'mode2' (min-max scaled noise); noise == NULL = zeros (mask_mode None).  noise: [B,channels,mask_rows,F]. */
int p2phd_spectro_encode_ex(const float* spec, int64_t B, int64_t F, int64_t M, int channels, float alpha, float min_value,
                            int mask_rows, int mask_mode, const float* noise, const float* noise_sign, float* log_spectro,
                            float* pha, float* norm8, float* partials, void* stream);
int p2phd_spectro_decode(const float* log_spectro, const float* norm_min_max, int64_t B, int64_t F, int64_t M,
                         float alpha, float min_value, float* spec, void* stream);

/* util.imdct's decode (util/util.py:104-126; caller generate_audio.py:40-42): log_spectro [B,channels,M,F] (magnitudes are
 * taken), pha [B,M,F] (+-1), (min,max) -> signed amplitudes spec [B,F,M] * scale.  channels = 2 (explicit encoding):
 * amplitude = ch0 + ch1, sign = pha on rows < keep_rows and sign(ch0 - ch1) on the rest; channels = 1: sign = pha on every
 * row (the caller splices its random signs into pha, :122-123).  keep_rows = int(M * (1 / up_ratio)), or M. */
int p2phd_spectro_decode_signed(const float* log_spectro, const float* pha, const float* norm_min_max, int64_t B, int64_t F,
                                int64_t M, int channels, int keep_rows, float min_value, float scale, float* spec, void* stream);

/* ------------------------------------------------------------------------------------------
 * Evaluation metrics (csrc/metrics.hip): compute_matrics (util/util.py:133-184).
 * hr, lr, sr [B,T] f32.  sr_matched [B,T] receives sr moment-matched to hr (:139-140); result4 (device) receives
 * (mse, snr_sr, snr_lr, lsd).  The LSD spectrogram is |STFT|^2 with n_fft2 = 2*opt.n_fft (power of two <= 4096),
 * hop2 = 2*opt.hop_length, window2 = kbdwin(2*opt.win_length) of win2 samples, reflect-centred when center != 0
 * (torchaudio.functional.spectrogram(pad=0, power=2, normalized=False) = torch.stft, :178-179).
 * tables: p2phd_stft_tables_floats(n_fft2) floats filled on the host; workspace: p2phd_metrics_workspace_bytes bytes.
 * ---------------------------------------------------------------------------------------- */
size_t p2phd_stft_tables_floats(int n_fft2);
int p2phd_stft_tables_fill(int n_fft2, float* host_out);
size_t p2phd_metrics_workspace_bytes(int64_t B, int64_t T, int n_fft2, int hop2, int win2, int center);
int p2phd_audio_metrics(const float* hr, const float* lr, const float* sr, int64_t B, int64_t T, int n_fft2, int hop2, int win2,
                        const float* window2, const float* tables, int center, float* sr_matched, float* result4,
                        void* workspace, void* stream);

/* ------------------------------------------------------------------------------------------
 * Input feeder resampler (csrc/resample.hip): the HR -> LR -> HR conversions of data/audio_dataset.py:55-57,109-113
 * (torchaudio.functional.resample there; its source is not in the reference, so the definition -- Hann-windowed sinc,
 * lowpass_filter_width 6, rolloff 0.99 -- is this build's own, stated in oracle/feeder.py).
 * x [B,T] f32 -> out [B,T_out], T_out = p2phd_resample_out_len(T, orig, new) = ceil(new*T/orig).
 * kernel: p2phd_resample_kernel_floats floats, filled on the host by p2phd_resample_kernel_fill, then copied to the device.
 * ---------------------------------------------------------------------------------------- */
int p2phd_resample_geometry(int orig_freq, int new_freq, int lowpass_filter_width, double rolloff, int* o, int* n, int* width,
                            int* klen);
size_t p2phd_resample_kernel_floats(int orig_freq, int new_freq, int lowpass_filter_width, double rolloff);
int p2phd_resample_kernel_fill(int orig_freq, int new_freq, int lowpass_filter_width, double rolloff, float* host_out);
int64_t p2phd_resample_out_len(int64_t T, int orig_freq, int new_freq);
int p2phd_resample_fwd(const float* x, int64_t B, int64_t T, int orig_freq, int new_freq, int lowpass_filter_width,
                       double rolloff, const float* kernel, float* out, int64_t T_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* P2PHD_H */
