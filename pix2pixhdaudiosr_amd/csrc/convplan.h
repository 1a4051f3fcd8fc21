// Internal descriptor of one gather-convolution launch and the launchers implemented in conv.hip.
#pragma once
#include "common.h"
#include <algorithm>

namespace p2phd {

struct GDesc {
  int N, Hin, Win, Cp_in;          // gathered tensor (NHWC, channel pitch Cp_in)
  int Hg, Wg;                      // GEMM-row grid per sample
  int sh, sw, pad_mode;            // gather stride; 0 = zeros outside, 1 = reflect, 2 / 3 = adjoint of ReflectionPad2d(1) (see gconv_kernel)
  int rx_base;                     // pad_mode 3: pixel index (from the tensor's first pixel) of the reflection extras block
  int Hout, Wout, Cp_out;          // written tensor
  int oh_mul, oh_off, ow_mul, ow_off;   // output lattice: (ho*oh_mul+oh_off, wo*ow_mul+ow_off)
  int Kout, KK, act;               // valid output channels, padded GEMM-K (row length of packed weights)
  int nth, ntw, dh0, dh_step, dw0, dw_step;   // tap (a,b): offset (dh0+a*dh_step, dw0+b*dw_step)
  int wr0, wr_step, ws0, ws_step;             // tap (a,b): kernel coordinate (wr0+a*wr_step, ws0+b*ws_step)
  int flat_m;                                 // M tiles run over all N*Hg*Wg pixels (set by the launcher)
  int cw_inject;                              // check build only (-DP2PHD_CHECK_WAITS): 1 = run the HALO loop with round 4's too-lax slab wait (the checker's sensitivity test)
  int cls_cp;                                 // > 0: merged sub-pixel launch, GEMM column = class * cls_cp + channel,
                                              //      class (pi,pj) = (col / cls_cp) writes output pixel (2*ho+pi, 2*wo+pj)
  int n_extent;                               // GEMM N extent (= Cp_out, or 4 * cls_cp when merged)
  int cls_skip;                               // merged 2 x 2-tap launch of a 3 x 3 stride-2 op (round 4): class (pi,pj) only uses the taps
                                              //   (ta <= pi, tb <= pj); the K order of a class row is [(0,0),(0,1),(1,0),(1,1)] for pi = 0 and
                                              //   [(0,0),(1,0),(0,1),(1,1)] for pi = 1, so every class's taps are a PREFIX of its K rows and a
                                              //   256 x 192 tile stops after the taps of its highest class (1, 2 or 4 instead of always 4)
  unsigned in_bytes, w_bytes;                 // extents of the gathered tensor / this launch's packed weights (buffer descriptors)
  int stats_slots;                            // slots per sample of the statistics table (set by the launcher)
  const float* out_scale;                     // fp8 operands: device pointer to the weights' de-quantisation factor
  // Fused first pass of the CONSUMER's InstanceNorm backward (input-gradient launches only): the tensor written here is
  // the gradient g of an InstanceNorm + activation output whose pre-normalisation values are bs_y (same shape as the
  // written tensor) with statistics bs_stats [N][Cp_out][2] = (mean, M2).  Every tile leaves its partial
  // (sum g', sum g' * yhat) per column in bs_out [N][tiles per sample][n_extent][2]; launch_bsum_merge folds them.
  const void* bs_y;
  const float* bs_stats;
  float* bs_out;                              // nullptr = off
  float bs_inv_hw, bs_eps, bs_slope;          // 1 / pixels per plane, eps, negative slope of the activation (1 = none)
  // Fused activation backward of a producer WITHOUT InstanceNorm (Conv + (Leaky)ReLU, e.g. the discriminator's first
  // layer): as_x is that block's output (= the tensor this launch's conv read, same shape as the written gradient); the
  // stored gradient is multiplied by act'(as_x) (1 above zero, bs_slope below) after the addend -- it leaves as dL/d(pre-act)
  const void* as_x;
  // launch geometry, set by launch_gconv_cfg: 1-D grid over tiles (tile = n_tile * grid_m + m_tile); workgroups >= sk_first
  // are K-parts of the last sk_tail tiles (sk_parts parts of sk_steps K slabs each; partials in sk_part, tickets in sk_ticket)
  int grid_m, sk_first, sk_tail, sk_parts, sk_steps;
  float* sk_part;
  unsigned* sk_ticket;
};

// Index map between a master weight tensor (PyTorch layout, f32) and a packed [rows][tap][inner] matrix:
//   offset(row, c, r, s) = (row % row_mod) * s_row + (row / row_mod) * s_rowq
//                        + (c % c_mod) * s_inner + (c / c_mod) * s_innerq + r * S + s
// row_mod / c_mod split a folded index (horizontal tap, channel) when the W taps ride on the channel axis.
//                        + ((r * S + s) * s_tap)          [s_tap = 1 for PyTorch layouts; = C for the K-major master layout]
struct WMap {
  int rows, inner;
  long s_row, s_inner;
  int row_mod; long s_rowq;
  int c_mod; long s_innerq;
  int S;
  long s_tap;
};
inline WMap plain_map(int rows, int inner, long s_row, long s_inner, int S, long s_tap = 1) {
  return WMap{rows, inner, s_row, s_inner, rows > 0 ? rows : 1, 0, inner > 0 ? inner : 1, 0, S, s_tap};
}

// operand type code of the fp8 forward launches (not part of the public dtype enum: activations stay bf16 at the ABI)
constexpr int P2PHD_FP8_INTERNAL = 2;
int launch_pack_fp8(const GDesc& d, const WMap& m, const float* w, void* wp8, int rows_pad, float* scale2, unsigned* amax_bits,
                    hipStream_t st);

// launch counters (p2phd_launch_count): which kernel family a call really took -- tests assert that the benchmarked step runs
// on the round-4/5 kernels and not on the generic loop behind them
enum LaunchFamily { LC_GCONV = 0, LC_HALO, LC_CLS_SKIP, LC_MARCH, LC_MARCH_W, LC_WGRAD, LC_SPLITK, LC_TILE256, LC_TILE128X192, LC_FAMILIES };
extern unsigned long long g_launch_count[LC_FAMILIES];

// tuning overrides (p2phd_set_option): 0 = heuristic
extern int g_opt_gconv_bm;
extern int g_opt_wgrad_tm;
extern int g_opt_tile128x192;      // 1 (default): small planes with 192-divisible outputs take the 128 x 192 tile where it saves a round
extern int g_opt_wgrad_xcd;          // 1 (default): XCD-aware tile order of the weight-gradient grid, 0: plain order (A/B)
extern int g_opt_c7_generic;
extern int g_opt_splitk_tail;       // 0: every tile is one workgroup, 1 (default): split-K tail where the cost model says so, 2: wherever possible (tests)
extern int g_opt_cus;               // > 0: CUs a conv launch may count on (a CU-masked compute stream); 0 = all of the device's
int device_cus();                   // multiProcessorCount of the current device, cached per device
extern int g_opt_reflect_generic;   // 1: reflect-padded 3x3 input gradients on the padded grid + fold (the general form)
extern int g_opt_c7_abl;          // timing experiments only (tools/time_c7.py): skip parts of c7_out_fwd      // 1: the 7x7 2-channel layers always take the generic W-fold path

// c7.hip: dedicated bf16 kernels of the generator's 7x7 end layers (full tiles of 8 x 128 pixels only)
bool c7_in_ok(const p2phd_conv_desc* c, bool ignore_option = false);
size_t c7_in_packed_elems(const p2phd_conv_desc* c);
int c7_in_pack(const p2phd_conv_desc* c, const float* w, void* wf, hipStream_t st);
int c7_in_slots(const p2phd_conv_desc* c);
int c7_in_fwd(const p2phd_conv_desc* c, const void* x, const void* wf, const float* bias, void* y, float* table, hipStream_t st);
bool c7_out_ok(const p2phd_conv_desc* c, bool ignore_option = false);
size_t c7_out_packed_elems(const p2phd_conv_desc* c);
int c7_out_pack(const p2phd_conv_desc* c, const float* w, void* wf, hipStream_t st);
int c7_out_fwd(const p2phd_conv_desc* c, const void* x, const void* wf, const float* bias, int act, void* y, hipStream_t st);
bool c7_out_dgrad_ok(const p2phd_conv_desc* c, bool ignore_option = false);
size_t c7_out_dgrad_packed_elems(const p2phd_conv_desc* c);
int c7_out_dgrad_pack(const p2phd_conv_desc* c, const float* w, void* wf, hipStream_t st);
int c7_out_dgrad(const p2phd_conv_desc* c, const void* dy, const void* wf, const float* w_master, void* dx, hipStream_t st);
// dfirst.hip: forward of the discriminator's first layer (Conv2d(<= 8, 64, 4, stride 2, padding 2) + activation, no statistics), bf16
extern int g_opt_dfirst;
bool dfirst_ok(const p2phd_conv_desc* c, bool ignore_option = false);
size_t dfirst_packed_elems(const p2phd_conv_desc* c);
int dfirst_pack(const p2phd_conv_desc* c, const float* w, void* wf, hipStream_t st);
int dfirst_fwd(const p2phd_conv_desc* c, const void* x, const void* wf, const float* bias, int act, void* y, hipStream_t st);
// dlast.hip: the discriminator's last layer (Conv2d(C, 1, 4, stride 1, padding 2), C a multiple of 128 up to 512), bf16: forward and input gradient
extern int g_opt_dlast;
bool dlast_ok(const p2phd_conv_desc* c, bool ignore_option = false);
size_t dlast_packed_elems(const p2phd_conv_desc* c, int which);
int dlast_pack(const p2phd_conv_desc* c, int which, const float* w, void* wfrag, hipStream_t st);
size_t dlast_fwd_workspace_floats(const p2phd_conv_desc* c);
int dlast_fwd(const p2phd_conv_desc* c, const void* x, const void* wf, const float* bias, void* y, float* part, hipStream_t st);
void dlast_dgrad_plan(const p2phd_conv_desc* c, int* bpw, int* slots);
size_t dlast_bsum_table_floats(const p2phd_conv_desc* c);
int dlast_dgrad(const p2phd_conv_desc* c, const void* dy, const void* wg, const void* addend, void* dx, const void* bs_y,
                const float* bs_stats, float* bs_out, float bs_inv_hw, float bs_eps, float bs_slope, hipStream_t st);
// march.hip: marching kernels of the generator's outermost stride-2 3x3 layers (bf16); which: 0 = forward, 1 = input gradient
extern int g_opt_march; extern int g_opt_cls_skip; extern int g_opt_gconv_halo; extern int g_opt_cw_inject;            // 1 (default): eligible layers take the marching kernels, 0: the generic gather-GEMM (A/B, parity tests)
int march_kind(const p2phd_conv_desc* c, int which);
int march_shape_kind(const p2phd_conv_desc* c, int which);     // the shape rule without the option (pack / workspace sizes)
size_t march_packed_elems(const p2phd_conv_desc* c, int which);
int march_pack(const p2phd_conv_desc* c, int which, const float* w, void* wf, hipStream_t st);
void march_plan(const p2phd_conv_desc* c, int which, int* slots, int* ncls, int* slot_rows, long* npix_cls, int* bs_tiles);
// in_stats != nullptr (forward launches): `in` is the raw output of an InstanceNorm block, normalised + activated on load
int march_run(const p2phd_conv_desc* c, int which, const void* in, const void* wf, const float* bias, void* out, float* table,
              const void* bs_y, const float* bs_stats, float* bs_out, float bs_inv_hw, float bs_eps, float bs_slope, hipStream_t st,
              const float* in_stats = nullptr, float in_slope = 1.f, float in_eps = 0.f);
bool march_w_ok(const p2phd_conv_desc* c);
size_t march_w_workspace_floats(const p2phd_conv_desc* c);
int march_w_run(const p2phd_conv_desc* c, const void* x, const void* dy, float* dw, int accumulate, const float* x_stats, float x_slope,
                float x_eps, float* slabs, hipStream_t st);
// thinwgrad.hip: weight gradient of the layers with <= 4 channels on one side (bf16); kind 0 = not eligible
int thin_wgrad_kind(const p2phd_conv_desc* c);
size_t thin_wgrad_workspace_floats(const p2phd_conv_desc* c);
int thin_wgrad(const p2phd_conv_desc* c, const void* x, const void* dy, float* dw, int accumulate, float* slabs, hipStream_t st);

inline int cpitch(int c) { return (c + 7) & ~7; }
inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

// `stats` (optional): [N][slots][classes][Cp_out][2] table of per-wave InstanceNorm partials (sum, squared deviations
// from the wave's mean), slots = ceil(Hg * Wg / *slot_rows), classes = 4 for merged sub-pixel launches else 1; merge
// with launch_stats_merge.  stat_table_floats() bounds its size for any tile configuration.
int launch_gconv(const GDesc& d, int dtype, const void* in, const void* wp, const float* bias, const void* addend,
                 void* out, float* stats, hipStream_t st, int* slot_rows = nullptr);
inline size_t stat_table_floats(const GDesc& d) {
  return (size_t)d.N * ((d.Hg * d.Wg + 31) / 32) * (d.cls_cp > 0 ? 4 : 1) * d.Cp_out * 2;
}
// stats[n][c] = (mean, sum of squared deviations) over the sample's plane, merged from `slots` partials per class in a
// fixed order (Chan et al.); slot s holds rows [s * slot_rows, (s+1) * slot_rows) of the npix rows of a sample.
// partial table of the fused InstanceNorm-backward sums (see GDesc::bs_out): floats needed for any tile height >= 128,
// and the merge into bstats [N][Cp][2]; `tile_rows` = the BM the launch used (returned through launch_gconv's slot_rows)
inline size_t bsum_table_floats(const GDesc& d) {
  const size_t tiles = ((size_t)d.Hg * d.Wg + 127) / 128;
  return (size_t)d.N * tiles * (size_t)(d.n_extent ? d.n_extent : d.Cp_out) * 2;
}
int launch_bsum_merge(const float* table, float* bstats, int N, long npix, int tile_rows, int n_extent, int cls_cp, int Cp, int C,
                      hipStream_t st);
int launch_stats_merge(const float* table, float* stats, int N, int slots, int ncls, int Cp, int C, long npix, int slot_rows,
                       hipStream_t st);
size_t wgrad_workspace_floats(const GDesc& d, int dtype, int M_rows, int M_rows_pad);
int launch_wgrad(const GDesc& d, const WMap& m, int dtype, const void* rows, int Cp_r, int M_rows, int M_rows_pad,
                 const void* gat, float* dwp, float* dw, int accumulate, hipStream_t st);
int launch_pack_merged(const GDesc& d, int dtype, const float* w, void* wp, int rows_pad, int K, int C, int R, int S, int pad,
                       long s_k, long s_c, hipStream_t st);
int launch_pack(const GDesc& d, const WMap& m, int dtype, const float* w, void* wp, int rows_pad, hipStream_t st);
bool gconv_plain_launch_takes_256x256(const GDesc& d, int dtype);
int launch_reflect_expand(int dtype, const void* dy, void* e_out, int N, int H, int W, int Cp, hipStream_t st);
int launch_reflect_fold(int dtype, const void* dxp, const void* addend, void* dx, int N, int H, int W, int Cp, int P,
                        hipStream_t st);
int launch_colsum(int dtype, const void* x, long P, int Cp, int K, float* db, int accumulate, hipStream_t st);

// stand-alone statistics of a [N][HW][Cp] tensor (W-folded layers): `scratch` = plane_stats_scratch_floats() floats
size_t plane_stats_scratch_floats(int N, long HW, int C);
int launch_plane_stats(int dtype, const void* y, float* stats, float* scratch, int N, long HW, int C, hipStream_t st);
int launch_expand_in(int dtype, const void* x, void* xe, int N, int H, int W, int Wo, int C, int S, int pad, int pad_mode,
                     hipStream_t st);
int launch_expand_dy(int dtype, const void* dy, void* dye, int N, int Ho, int Wo, int Wy, int K, int S, hipStream_t st);
int launch_hsum(int dtype, const void* Y, const float* bias, void* y, int N, int H, int Wo, int Wy, int K, int S,
                int act, hipStream_t st);

}  // namespace p2phd
