// C ABI of the convolution family: turns a reference-level layer description (Conv2d / ConvTranspose2d with
// stride, padding and an optional ReflectionPad2d in front) into gather-convolution launches (conv.hip), using the
// W-fold forms (wfold.hip) for layers with <= 4 input or output channels.
#include "convplan.h"
#include <vector>

namespace {
using namespace p2phd;

struct Plan {
  GDesc d;
  size_t w_off;      // element offset of this launch's block inside the packed weight buffer
  int rows_pad;
};

enum { FOLD_NONE = 0, FOLD_IN = 1, FOLD_OUT = 2 };

int elem_size(int dtype) { return dtype == P2PHD_BF16 ? 2 : 4; }
size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

void out_size(const p2phd_conv_desc* c, int* Ho, int* Wo) {
  if (c->transposed) {
    *Ho = (c->H - 1) * c->stride - 2 * c->pad + c->R + c->opad;
    *Wo = (c->W - 1) * c->stride - 2 * c->pad + c->S + c->opad;
  } else {
    *Ho = (c->H + 2 * c->pad - c->R) / c->stride + 1;
    *Wo = (c->W + 2 * c->pad - c->S) / c->stride + 1;
  }
}

// K-major master weights ([K][R][S][C], round 3): plain stride-1 Conv2d whose packed forward row IS the master row -- no
// channel padding (C % 8 == 0), no K padding (R*S*C % 64 == 0), no W-fold / dedicated 7x7 path, a weight-gradient tile height
bool kmajor_shape_ok(const p2phd_conv_desc* c) {
  return !c->transposed && c->stride == 1 && c->C % 8 == 0 && c->C >= 64 && c->K >= 64 && c->R * c->S <= 16 &&
         (c->R * c->S * c->C) % 64 == 0 && c->K > 4 && c->C > 4;
}

int check_desc(const p2phd_conv_desc* c) {
  P2PHD_REQUIRE(c != nullptr, "conv: null descriptor");
  P2PHD_REQUIRE(c->N >= 0 && c->C >= 1 && c->H >= 1 && c->W >= 1 && c->K >= 1 && c->R >= 1 && c->S >= 1, "conv: bad sizes");
  P2PHD_REQUIRE(c->stride >= 1 && c->stride <= 2 && c->pad >= 0 && c->pad < 8, "conv: stride must be 1 or 2, pad < 8");
  P2PHD_REQUIRE(c->dtype == P2PHD_F32 || c->dtype == P2PHD_BF16, "conv: dtype must be f32 or bf16");
  P2PHD_REQUIRE(c->pad_mode == 0 || c->pad_mode == 1, "conv: pad_mode must be 0 (zeros) or 1 (reflect)");
  if (c->pad_mode == 1) {
    P2PHD_REQUIRE(!c->transposed, "conv: reflect padding is not defined for ConvTranspose2d");
    P2PHD_REQUIRE(c->pad < c->H && c->pad < c->W, "conv: reflect padding %d needs a larger image than %dx%d", c->pad, c->H, c->W);
  }
  P2PHD_REQUIRE(c->w_layout == 0 || c->w_layout == 1, "conv: w_layout must be 0 (PyTorch) or 1 (K-major)");
  P2PHD_REQUIRE(c->w_layout == 0 || kmajor_shape_ok(c), "conv: this layer cannot keep its master weights K-major (p2phd_conv_kmajor_ok)");
  if (c->transposed) P2PHD_REQUIRE(c->opad >= 0 && c->opad < c->stride, "conv: output_padding must be < stride");
  else P2PHD_REQUIRE(c->H + 2 * c->pad >= c->R && c->W + 2 * c->pad >= c->S, "conv: kernel larger than padded input");
  int Ho, Wo;
  out_size(c, &Ho, &Wo);
  P2PHD_REQUIRE(Ho >= 1 && Wo >= 1, "conv: empty output");
  return P2PHD_OK;
}

// shape tests only (what the packed buffer must hold whatever the options say at the moment: an option can change between
// the pack and the call, the buffer layout must not).  Pure functions: no global is touched (advisor, round 4).
bool c7_fast_shape(const p2phd_conv_desc* c) { return c7_in_ok(c, true); }
bool c7_out_shape(const p2phd_conv_desc* c) { return c7_out_ok(c, true); }
bool c7_dgrad_shape(const p2phd_conv_desc* c) { return c7_out_dgrad_ok(c, true); }
size_t march_shape_elems(const p2phd_conv_desc* c, int which) { return march_packed_elems(c, which); }
size_t generic_packed_elems(const std::vector<struct Plan>& plans);

int fold_mode(const p2phd_conv_desc* c) {
  if (c->transposed || c->stride != 1) return FOLD_NONE;
  if (c->K <= 4 && c->S * c->K <= 32) return FOLD_OUT;
  if (c->C <= 4 && c->S * c->C <= 32) return FOLD_IN;
  return FOLD_NONE;
}

size_t generic_packed_elems(const std::vector<Plan>& plans) {
  size_t n = 0;
  for (auto& p : plans) n += (size_t)p.rows_pad * p.d.KK;
  return n;
}

GDesc base_desc(int N) {
  GDesc d{};
  d.N = N;
  d.sh = d.sw = 1;
  d.oh_mul = d.ow_mul = 1;
  d.dh_step = d.dw_step = 1;
  d.wr_step = d.ws_step = 1;
  return d;
}

// "direct" form: out[o] = sum_r in[o*stride - pad + r] * w[r]
Plan direct_plan(int N, int Hin, int Win, int Cin, int Hg, int Wg, int Kout, int R, int S, int stride, int pad, int pad_mode) {
  Plan p{};
  GDesc& d = p.d;
  d = base_desc(N);
  d.Hin = Hin; d.Win = Win; d.Cp_in = cpitch(Cin);
  d.Hg = Hg; d.Wg = Wg; d.sh = d.sw = stride; d.pad_mode = pad_mode;
  d.Hout = Hg; d.Wout = Wg; d.Cp_out = cpitch(Kout); d.Kout = Kout;
  d.nth = R; d.ntw = S; d.dh0 = -pad; d.dw0 = -pad;
  d.KK = std::max(64, round_up(R * S * d.Cp_in, 64));
  p.rows_pad = round_up(Kout, 128);
  return p;
}

// "transposed" form: out[o] = sum_{r : (o + pad - r) % stride == 0} in[(o + pad - r)/stride] * w[r], one plan per
// output residue class (pi, pj) modulo the stride
void transposed_plans(int N, int Hin, int Win, int Cin, int Hout, int Wout, int Kout, int R, int S, int stride, int pad,
                      std::vector<Plan>& out) {
  size_t off = 0;
  for (int pi = 0; pi < stride; ++pi)
    for (int pj = 0; pj < stride; ++pj) {
      const int ch = (Hout - pi + stride - 1) / stride, cw = (Wout - pj + stride - 1) / stride;
      if (ch <= 0 || cw <= 0) continue;
      Plan p{};
      GDesc& d = p.d;
      d = base_desc(N);
      d.Hin = Hin; d.Win = Win; d.Cp_in = cpitch(Cin);
      d.Hg = ch; d.Wg = cw; d.pad_mode = 0;
      d.Hout = Hout; d.Wout = Wout; d.Cp_out = cpitch(Kout); d.Kout = Kout;
      d.oh_mul = stride; d.oh_off = pi; d.ow_mul = stride; d.ow_off = pj;
      const int r0 = (pi + pad) % stride, s0 = (pj + pad) % stride;
      d.nth = r0 < R ? (R - r0 + stride - 1) / stride : 0;
      d.ntw = s0 < S ? (S - s0 + stride - 1) / stride : 0;
      if (d.nth == 0 || d.ntw == 0) { d.nth = d.ntw = 0; }
      d.dh0 = (pi + pad - r0) / stride; d.dh_step = -1;
      d.dw0 = (pj + pad - s0) / stride; d.dw_step = -1;
      d.wr0 = r0; d.wr_step = stride; d.ws0 = s0; d.ws_step = stride;
      d.KK = std::max(64, round_up(d.nth * d.ntw * d.Cp_in, 64));
      p.rows_pad = round_up(Kout, 128);
      p.w_off = off;
      off += (size_t)p.rows_pad * d.KK;
      out.push_back(p);
    }
}

// Merged sub-pixel plan for stride 2 (the four residue classes of a transposed-form op in ONE launch): GEMM row =
// input-resolution pixel (a, b), GEMM columns = (class, channel), taps (dh, dw) in {lo..hi}^2 shared by all classes with
// zero weights where a class does not use a tap.  The gathered tensor is read once instead of four times and a
// workgroup writes whole runs of adjacent output pixels instead of every other one.
bool merged_plan(int N, int Hin, int Win, int Cin, int Hout, int Wout, int Kout, int R, int S, int pad, Plan* out, int dtype = P2PHD_F32) {
  // dh = (pi + pad - r) / 2 over all classes and kernel rows with (pi + pad - r) even
  int lo = 1 << 30, hi = -(1 << 30);
  for (int pi = 0; pi < 2; ++pi)
    for (int r = 0; r < R; ++r)
      if (((pi + pad - r) & 1) == 0) { const int dh = (pi + pad - r) / 2; lo = std::min(lo, dh); hi = std::max(hi, dh); }
  int lo_w = 1 << 30, hi_w = -(1 << 30);
  for (int pj = 0; pj < 2; ++pj)
    for (int s = 0; s < S; ++s)
      if (((pj + pad - s) & 1) == 0) { const int dw = (pj + pad - s) / 2; lo_w = std::min(lo_w, dw); hi_w = std::max(hi_w, dw); }
  if (lo > hi || lo_w > hi_w) return false;
  Plan p{};
  GDesc& d = p.d;
  d = base_desc(N);
  d.Hin = Hin; d.Win = Win; d.Cp_in = cpitch(Cin);
  d.Hg = (Hout + 1) / 2; d.Wg = (Wout + 1) / 2; d.pad_mode = 0;
  d.Hout = Hout; d.Wout = Wout; d.Cp_out = cpitch(Kout); d.Kout = Kout;
  d.cls_cp = d.Cp_out; d.n_extent = 4 * d.Cp_out;
  d.nth = hi - lo + 1; d.ntw = hi_w - lo_w + 1; d.dh0 = lo; d.dw0 = lo_w;
  d.KK = std::max(64, round_up(d.nth * d.ntw * d.Cp_in, 64));
  p.rows_pad = round_up(4 * d.Cp_out, 128);
  // 3 x 3, stride 2, pad 1 (the generator's down path backwards, its up path forwards): class (pi,pj) uses (pi + 1)(pj + 1) of the
  // 2 x 2 taps -- 9 of the 16 (class, tap) pairs; the rest multiplies packed zeros.  With per-class tap order the launch skips
  // them (GDesc::cls_skip; gconv_kernel).  Needs the 256 x 192 tile (16-bit types) to sit inside one pi: cls_cp a multiple or a
  // divisor (>= 96) of 192, and planes that fill 256-row tiles.
  {
    const long npix = (long)d.Hg * d.Wg;
    const bool taps_3x3 = R == 3 && S == 3 && pad == 1 && d.nth == 2 && d.ntw == 2 && lo == 0 && lo_w == 0;
    const bool aligned = d.cls_cp >= 96 && (d.cls_cp % 192 == 0 || 192 % d.cls_cp == 0);
    d.cls_skip = g_opt_cls_skip != 0 && dtype == P2PHD_BF16 && taps_3x3 && aligned && npix % 256 == 0 && d.Cp_in % 32 == 0 &&
                 (long)N * (npix / 256) * (d.n_extent / 192) >= 192 ? 1 : 0;
  }
  *out = p;
  return true;
}

// ---- W-fold plans (stride 1, not transposed) ----
// output fold, forward: Y[n,ho,w',(tw,k)] on the grid Ho x Wy, Wy = Wo + S - 1
Plan kfold_fwd_plan(const p2phd_conv_desc* c, int Ho, int Wo) {
  Plan p = direct_plan(c->N, c->H, c->W, c->C, Ho, Wo + c->S - 1, c->S * c->K, c->R, 1, 1, c->pad, c->pad_mode);
  return p;   // taps (th, -): dh = th - pad, dw = -pad
}
// output fold, input gradient: gathers dyE[n, i' + pe - th, j' + pe, (tw,k)]
Plan kfold_dgrad_plan(const p2phd_conv_desc* c, int Ho, int Wo) {
  const bool reflect = c->pad_mode == 1;
  const int P = reflect ? c->pad : 0, pe = reflect ? 0 : c->pad;
  Plan p{};
  GDesc& d = p.d;
  d = base_desc(c->N);
  d.Hin = Ho; d.Win = Wo + c->S - 1; d.Cp_in = cpitch(c->S * c->K);
  d.Hg = c->H + 2 * P; d.Wg = c->W + 2 * P; d.pad_mode = 0;
  d.Hout = d.Hg; d.Wout = d.Wg; d.Cp_out = cpitch(c->C); d.Kout = c->C;
  d.nth = c->R; d.ntw = 1; d.dh0 = pe; d.dh_step = -1; d.dw0 = pe;
  d.KK = std::max(64, round_up(c->R * d.Cp_in, 64));
  p.rows_pad = round_up(c->C, 128);
  return p;
}
// input fold, forward: gathers Xe[n, ho + th - pad, wo, (tw,c)]
Plan cfold_fwd_plan(const p2phd_conv_desc* c, int Ho, int Wo) {
  Plan p = direct_plan(c->N, c->H, Wo, c->S * c->C, Ho, Wo, c->K, c->R, 1, 1, c->pad, c->pad_mode);
  p.d.dw0 = 0;
  return p;
}

WMap kfold_rows_map(const p2phd_conv_desc* c) {     // rows (tw,k), inner c
  const long RS = (long)c->R * c->S;
  return WMap{c->S * c->K, c->C, c->C * RS, RS, c->K, 1, c->C, 0, c->S, 1};
}
WMap kfold_inner_map(const p2phd_conv_desc* c) {    // rows c, inner (tw,k)
  const long RS = (long)c->R * c->S;
  return WMap{c->C, c->S * c->K, RS, c->C * RS, c->C, 0, c->K, 1, c->S, 1};
}
WMap cfold_map(const p2phd_conv_desc* c) {          // rows k, inner (tw,c)
  const long RS = (long)c->R * c->S;
  return WMap{c->K, c->S * c->C, c->C * RS, RS, c->K, 0, c->C, 1, c->S, 1};
}

// which: 0 = forward, 1 = input gradient
void make_plans(const p2phd_conv_desc* c, int which, std::vector<Plan>& plans, WMap* m) {
  int Ho, Wo;
  out_size(c, &Ho, &Wo);
  const long RS = (long)c->R * c->S;
  const int fold = fold_mode(c);
  if (which == 0 && fold == FOLD_OUT) {
    plans.push_back(kfold_fwd_plan(c, Ho, Wo)); *m = kfold_rows_map(c);
  } else if (which == 0 && fold == FOLD_IN) {
    plans.push_back(cfold_fwd_plan(c, Ho, Wo)); *m = cfold_map(c);
  } else if (which == 1 && fold == FOLD_OUT) {
    plans.push_back(kfold_dgrad_plan(c, Ho, Wo)); *m = kfold_inner_map(c);
  } else if (which == 0 && !c->transposed) {
    plans.push_back(direct_plan(c->N, c->H, c->W, c->C, Ho, Wo, c->K, c->R, c->S, c->stride, c->pad, c->pad_mode));
    *m = c->w_layout == 1 ? plain_map(c->K, c->C, c->C * RS, 1, c->S, c->C) : plain_map(c->K, c->C, c->C * RS, RS, c->S);
  } else if (which == 0 && c->transposed) {
    Plan mp;
    *m = plain_map(c->K, c->C, RS, c->K * RS, c->S);                          // weight [C][K][R][S]
    if (c->stride == 2 && merged_plan(c->N, c->H, c->W, c->C, Ho, Wo, c->K, c->R, c->S, c->pad, &mp, c->dtype)) plans.push_back(mp);
    else transposed_plans(c->N, c->H, c->W, c->C, Ho, Wo, c->K, c->R, c->S, c->stride, c->pad, plans);
  } else if (which == 1 && !c->transposed) {
    const int P = c->pad_mode == 1 ? c->pad : 0;                               // reflect: gradient on the padded grid
    Plan mp;
    *m = plain_map(c->C, c->K, RS, c->C * RS, c->S);                          // weight [K][C][R][S]
    if (c->w_layout == 1) *m = plain_map(c->C, c->K, 1, c->C * RS, c->S, c->C);   // ... kept as [K][R][S][C]
    if (c->stride == 2 && merged_plan(c->N, Ho, Wo, c->K, c->H + 2 * P, c->W + 2 * P, c->C, c->R, c->S, c->pad_mode == 1 ? 0 : c->pad, &mp, c->dtype))
      plans.push_back(mp);
    else
      transposed_plans(c->N, Ho, Wo, c->K, c->H + 2 * P, c->W + 2 * P, c->C, c->R, c->S, c->stride, c->pad_mode == 1 ? 0 : c->pad, plans);
  } else {
    plans.push_back(direct_plan(c->N, Ho, Wo, c->K, c->H, c->W, c->C, c->R, c->S, c->stride, c->pad, 0));
    *m = plain_map(c->C, c->K, c->K * RS, RS, c->S);                          // weight [C][K][R][S]
  }
}

size_t padded_dx_bytes(const p2phd_conv_desc* c) {
  if (c->pad_mode != 1) return 0;
  return align256((size_t)c->N * (c->H + 2 * c->pad) * (c->W + 2 * c->pad) * cpitch(c->C) * elem_size(c->dtype));
}
size_t folded_dy_bytes(const p2phd_conv_desc* c, int Ho, int Wo) {
  return align256((size_t)c->N * Ho * (Wo + c->S - 1) * cpitch(c->S * c->K) * elem_size(c->dtype));
}
size_t folded_x_bytes(const p2phd_conv_desc* c, int Wo) {
  return align256((size_t)c->N * c->H * Wo * cpitch(c->S * c->C) * elem_size(c->dtype));
}

}  // namespace

extern "C" int p2phd_channel_pitch(int channels) { return p2phd::cpitch(channels); }

extern "C" int p2phd_conv_kmajor_ok(const p2phd_conv_desc* c) {
  if (c == nullptr) return 0;
  p2phd_conv_desc t = *c;
  t.w_layout = 0;
  if (check_desc(&t) != P2PHD_OK) return 0;
  return (kmajor_shape_ok(c) && fold_mode(c) == FOLD_NONE && !c7_fast_shape(c) && !c7_out_shape(c) && !c7_dgrad_shape(c) &&
          !thin_wgrad_kind(c)) ? 1 : 0;
}

extern "C" int p2phd_conv_out_size(const p2phd_conv_desc* c, int* Ho, int* Wo) {
  if (int rc = check_desc(c)) return rc;
  out_size(c, Ho, Wo);
  return P2PHD_OK;
}

extern "C" size_t p2phd_conv_packed_bytes(const p2phd_conv_desc* c, int which) {
  if (check_desc(c) != P2PHD_OK || (which != 0 && which != 1)) return 0;
  std::vector<Plan> plans; WMap m;
  make_plans(c, which, plans, &m);
  size_t n = 0;
  for (auto& p : plans) n += (size_t)p.rows_pad * p.d.KK;
  if (dlast_ok(c, true)) n += dlast_packed_elems(c, which);                 // fragment-ordered copy for dlast.hip, behind the W-fold pack
  if (which == 0 && dfirst_ok(c, true)) n += dfirst_packed_elems(c);        // fragment-ordered copy for dfirst.hip, behind the generic pack
  if (which == 0 && c7_fast_shape(c)) n += c7_in_packed_elems(c);           // fragment-ordered copy for c7.hip, behind the W-fold pack
  if (which == 0 && c7_out_shape(c)) n += c7_out_packed_elems(c);
  if (which == 1 && c7_dgrad_shape(c))                                       // + fragment-ordered copy + f32 master copy (border fix)
    return (n + c7_out_dgrad_packed_elems(c)) * elem_size(c->dtype) + (size_t)c->K * c->C * c->R * c->S * sizeof(float);
  n += march_shape_elems(c, which);                                          // fragment-ordered copy for march.hip, behind the generic pack
  return n * elem_size(c->dtype);
}

// Which variant of the packed layout the launches of (desc, which) read.  The packed buffer of a layer is NOT a function of the
// layer alone: a tap-skipping merged launch (GDesc::cls_skip) keeps the 2 x 2 taps of its pi = 1 classes in per-class order,
// and whether a launch skips depends on N, the plane and the "cls_skip" option.  A cache of packed buffers must key on this
// value next to (which, dtype) -- a pack made for one batch size may not serve another (advisor, round 4: train.py:206 runs
// inference on a smaller last batch between training steps).
extern "C" int p2phd_conv_pack_layout(const p2phd_conv_desc* c, int which) {
  if (check_desc(c) != P2PHD_OK || (which != 0 && which != 1)) return -1;
  std::vector<Plan> plans; WMap m;
  make_plans(c, which, plans, &m);
  int id = 0;
  for (auto& p : plans) if (p.d.cls_skip != 0) id |= 1;
  return id;
}

extern "C" int p2phd_conv_pack_weights(const p2phd_conv_desc* c, int which, const float* w, void* packed, void* stream) {
  if (int rc = check_desc(c)) return rc;
  P2PHD_REQUIRE(which == 0 || which == 1, "pack_weights: which must be 0 (forward) or 1 (input gradient)");
  P2PHD_REQUIRE(w && packed, "pack_weights: null pointer");
  std::vector<Plan> plans; WMap m;
  make_plans(c, which, plans, &m);
  for (auto& p : plans) {
    char* dst = static_cast<char*>(packed) + p.w_off * elem_size(c->dtype);
    if (p.d.cls_cp > 0) {
      // rows = m.rows output channels, inner = m.inner reduction channels, master strides from the plain map
      const int pad_eff = (which == 1 && c->pad_mode == 1) ? 0 : c->pad;
      if (int rc = launch_pack_merged(p.d, c->dtype, w, dst, p.rows_pad, m.rows, m.inner, c->R, c->S, pad_eff, m.s_row, m.s_inner,
                                      (hipStream_t)stream)) return rc;
    } else if (int rc = launch_pack(p.d, m, c->dtype, w, dst, p.rows_pad, (hipStream_t)stream)) return rc;
  }
  if (march_shape_elems(c, which) > 0) {
    return march_pack(c, which, w, static_cast<char*>(packed) + generic_packed_elems(plans) * elem_size(c->dtype), (hipStream_t)stream);
  }
  if (dlast_ok(c, true))
    return dlast_pack(c, which, w, static_cast<char*>(packed) + generic_packed_elems(plans) * elem_size(c->dtype), (hipStream_t)stream);
  if (which == 0 && dfirst_ok(c, true))
    return dfirst_pack(c, w, static_cast<char*>(packed) + generic_packed_elems(plans) * elem_size(c->dtype), (hipStream_t)stream);
  if (which == 0 && c7_fast_shape(c)) {
    size_t n = 0;
    for (auto& p : plans) n += (size_t)p.rows_pad * p.d.KK;
    return c7_in_pack(c, w, static_cast<char*>(packed) + n * elem_size(c->dtype), (hipStream_t)stream);
  }
  if (which == 0 && c7_out_shape(c)) {
    size_t n = 0;
    for (auto& p : plans) n += (size_t)p.rows_pad * p.d.KK;
    return c7_out_pack(c, w, static_cast<char*>(packed) + n * elem_size(c->dtype), (hipStream_t)stream);
  }
  if (which == 1 && c7_dgrad_shape(c)) {
    size_t n = 0;
    for (auto& p : plans) n += (size_t)p.rows_pad * p.d.KK;
    char* frag = static_cast<char*>(packed) + n * elem_size(c->dtype);
    if (int rc = c7_out_dgrad_pack(c, w, frag, (hipStream_t)stream)) return rc;
    char* master = frag + c7_out_dgrad_packed_elems(c) * elem_size(c->dtype);
    if (hipMemcpyAsync(master, w, (size_t)c->K * c->C * c->R * c->S * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream) != hipSuccess) {
      set_error("pack_weights: copy of the master weights failed");
      return P2PHD_ELAUNCH;
    }
  }
  return P2PHD_OK;
}

// per-wave InstanceNorm partials of the conv epilogues: [N][slots][classes][Cp][2] floats behind the layer's other scratch
size_t stat_table_bytes(const std::vector<Plan>& plans) {
  size_t n = 0;
  for (const auto& p : plans) n = std::max(n, stat_table_floats(p.d));
  return align256(n * sizeof(float));
}

extern "C" size_t p2phd_conv_fwd_workspace_bytes(const p2phd_conv_desc* c) {
  if (check_desc(c) != P2PHD_OK) return 0;
  int Ho, Wo;
  out_size(c, &Ho, &Wo);
  const int fold = fold_mode(c);
  if (fold == FOLD_OUT)                                          // Y has the shape of the folded dy; statistics: plane pass  (dlast.hip: per-pixel tap partials)
    return std::max(folded_dy_bytes(c, Ho, Wo) + align256(plane_stats_scratch_floats(c->N, (long)Ho * Wo, c->K) * sizeof(float)),
                    dlast_ok(c, true) ? align256(dlast_fwd_workspace_floats(c) * sizeof(float)) : (size_t)0);
  std::vector<Plan> plans; WMap m;
  make_plans(c, 0, plans, &m);
  const size_t table = stat_table_bytes(plans);
  if (fold == FOLD_IN) return folded_x_bytes(c, Wo) + table;
  return table;
}

extern "C" int p2phd_conv_fwd(const p2phd_conv_desc* c, const void* x, const void* wp, const float* bias, int act,
                              void* y, float* stats, void* workspace, void* stream) {
  if (int rc = check_desc(c)) return rc;
  P2PHD_REQUIRE(act >= P2PHD_ACT_NONE && act <= P2PHD_ACT_RELU, "conv_fwd: bad activation %d", act);
  P2PHD_REQUIRE(stats == nullptr || act == P2PHD_ACT_NONE, "conv_fwd: InstanceNorm statistics are taken of the pre-activation output");
  if (c->N == 0) return P2PHD_OK;
  P2PHD_REQUIRE(x && wp && y, "conv_fwd: null pointer");
  hipStream_t st = (hipStream_t)stream;
  std::vector<Plan> plans; WMap m;
  make_plans(c, 0, plans, &m);
  int Ho, Wo;
  out_size(c, &Ho, &Wo);
  const int fold = fold_mode(c);
  P2PHD_REQUIRE((fold == FOLD_NONE && stats == nullptr) || workspace,
                "conv_fwd: this layer needs p2phd_conv_fwd_workspace_bytes of scratch (W-fold image / statistics partials)");
  if (fold == FOLD_OUT && stats == nullptr && act == P2PHD_ACT_NONE && dlast_ok(c)) {
    // the discriminator's head (512 -> 1, 4 x 4): one pass over x with the 16 taps as an MFMA dimension, then a 16-term gather (dlast.hip)
    const void* wf = static_cast<const char*>(wp) + generic_packed_elems(plans) * elem_size(c->dtype);
    return dlast_fwd(c, x, wf, bias, y, static_cast<float*>(workspace), st);
  }
  if (fold == FOLD_OUT && stats == nullptr && c7_out_ok(c)) {     // the generator head: marching kernel of c7.hip
    size_t n = 0;
    for (auto& p : plans) n += (size_t)p.rows_pad * p.d.KK;
    return c7_out_fwd(c, x, static_cast<const char*>(wp) + n * elem_size(c->dtype), bias, act, y, st);
  }
  if (fold == FOLD_OUT) {
    Plan& p = plans[0];
    if (int rc = launch_gconv(p.d, c->dtype, x, wp, nullptr, nullptr, workspace, nullptr, st)) return rc;
    if (int rc = launch_hsum(c->dtype, workspace, bias, y, c->N, Ho, Wo, Wo + c->S - 1, c->K, c->S, act, st)) return rc;
    if (stats == nullptr) return P2PHD_OK;
    float* scratch = reinterpret_cast<float*>(static_cast<char*>(workspace) + folded_dy_bytes(c, Ho, Wo));
    return launch_plane_stats(c->dtype, y, stats, scratch, c->N, (long)Ho * Wo, c->K, st);
  }
  const void* src = x;
  float* table = static_cast<float*>(workspace);
  if (fold == FOLD_NONE && act == P2PHD_ACT_NONE && march_kind(c, 0)) {
    // the generator's outermost stride-2 layer: marching kernel (march.hip), weights behind the generic pack
    const void* wf = static_cast<const char*>(wp) + generic_packed_elems(plans) * elem_size(c->dtype);
    if (int rc = march_run(c, 0, x, wf, bias, y, stats ? table : nullptr, nullptr, nullptr, nullptr, 0.f, 0.f, 0.f, st)) return rc;
    if (stats == nullptr) return P2PHD_OK;
    int slots = 0, ncls = 1, slot_rows = 0;
    long npix_cls = 0;
    march_plan(c, 0, &slots, &ncls, &slot_rows, &npix_cls, nullptr);
    return launch_stats_merge(table, stats, c->N, slots, ncls, cpitch(c->K), c->K, npix_cls, slot_rows, st);
  }
  if (fold == FOLD_NONE && stats == nullptr && dfirst_ok(c)) {
    // the discriminator's first layer (4 -> 64, 4 x 4 stride 2): pixels straight into MFMA fragments, weights in registers (dfirst.hip)
    const void* wf = static_cast<const char*>(wp) + generic_packed_elems(plans) * elem_size(c->dtype);
    return dfirst_fwd(c, x, wf, bias, act, y, st);
  }
  if (fold == FOLD_IN && act == P2PHD_ACT_NONE && c7_in_ok(c)) {
    // dedicated 2-channel 7x7 kernel (c7.hip): halo once through LDS, weights in registers, whole-row stores
    size_t n = 0;
    for (auto& p : plans) n += (size_t)p.rows_pad * p.d.KK;
    const void* wf = static_cast<const char*>(wp) + n * elem_size(c->dtype);
    if (int rc = c7_in_fwd(c, x, wf, bias, y, stats ? table : nullptr, st)) return rc;
    if (stats == nullptr) return P2PHD_OK;
    return launch_stats_merge(table, stats, c->N, c7_in_slots(c), 1, cpitch(c->K), c->K, (long)Ho * Wo, 256, st);
  }
  if (fold == FOLD_IN) {
    if (int rc = launch_expand_in(c->dtype, x, workspace, c->N, c->H, c->W, Wo, c->C, c->S, c->pad, c->pad_mode, st)) return rc;
    src = workspace;
    table = reinterpret_cast<float*>(static_cast<char*>(workspace) + folded_x_bytes(c, Wo));
  }
  // InstanceNorm statistics: every wave of the conv epilogue stores the partial of its rows (plain stores), one merge
  // launch turns them into (mean, sum of squared deviations) per (sample, channel)
  P2PHD_REQUIRE(stats == nullptr || plans.size() == 1, "conv_fwd: statistics need a single-launch plan");
  int slot_rows = 0;
  for (auto& p : plans) {
    p.d.act = act;
    const char* w = static_cast<const char*>(wp) + p.w_off * elem_size(c->dtype);
    if (int rc = launch_gconv(p.d, c->dtype, src, w, bias, nullptr, y, stats ? table : nullptr, st, &slot_rows)) return rc;
  }
  if (stats == nullptr) return P2PHD_OK;
  const GDesc& d = plans[0].d;
  const long npix = (long)d.Hg * d.Wg;
  return launch_stats_merge(table, stats, c->N, (int)((npix + slot_rows - 1) / slot_rows), d.cls_cp > 0 ? 4 : 1, cpitch(c->K), c->K,
                            npix, slot_rows, st);
}

// ------------------------------------------------------------------------------------------------------
// fp8 forward (BASELINE configs[4]: "bf16 + fp8 MFMA conv weights"): OCP e4m3 operands on the block-scaled v_mfma_scale_f32_32x32x64_f8f6f4 (unit scales) for
// the wide stride-1 layers (residual trunk, discriminator 256 -> 512), fp32 master weights and the bf16 backward unchanged.
// Weights are quantised per layer (scale = amax / 448, found on the device: no host synchronisation), activations arrive
// already quantised (scale 1: they are InstanceNorm outputs) from p2phd_instnorm_act_fwd_q8.
// ------------------------------------------------------------------------------------------------------
namespace {
bool fp8_eligible(const p2phd_conv_desc* c) {
  return c->dtype == P2PHD_BF16 && !c->transposed && c->stride == 1 && fold_mode(c) == FOLD_NONE && c->C % 16 == 0 &&
         (c->R * c->S * c->C) % 128 == 0;
}
size_t fp8_weight_bytes(const p2phd_conv_desc* c) { return align256((size_t)round_up(c->K, 128) * c->R * c->S * c->C); }
}  // namespace

extern "C" int p2phd_conv_fp8_eligible(const p2phd_conv_desc* c) { return check_desc(c) == P2PHD_OK && fp8_eligible(c) ? 1 : 0; }

extern "C" size_t p2phd_conv_fp8_packed_bytes(const p2phd_conv_desc* c) {
  if (check_desc(c) != P2PHD_OK || !fp8_eligible(c)) return 0;
  return fp8_weight_bytes(c) + 256;                               // + {scale, 1/scale, amax bits}
}

extern "C" int p2phd_conv_fp8_pack_weights(const p2phd_conv_desc* c, const float* w, void* packed8, void* stream) {
  if (int rc = check_desc(c)) return rc;
  P2PHD_REQUIRE(fp8_eligible(c), "conv_fp8: layer not eligible (stride-1 Conv2d, channels %% 16, taps * channels %% 128)");
  P2PHD_REQUIRE(w && packed8, "conv_fp8_pack_weights: null pointer");
  int Ho, Wo;
  out_size(c, &Ho, &Wo);
  Plan p = direct_plan(c->N, c->H, c->W, c->C, Ho, Wo, c->K, c->R, c->S, c->stride, c->pad, c->pad_mode);
  p.d.KK = c->R * c->S * c->C;
  const long RS = (long)c->R * c->S;
  const WMap m = c->w_layout == 1 ? plain_map(c->K, c->C, c->C * RS, 1, c->S, c->C) : plain_map(c->K, c->C, c->C * RS, RS, c->S);
  char* tail = static_cast<char*>(packed8) + fp8_weight_bytes(c);
  return launch_pack_fp8(p.d, m, w, packed8, p.rows_pad, reinterpret_cast<float*>(tail), reinterpret_cast<unsigned*>(tail + 16),
                         (hipStream_t)stream);
}

extern "C" int p2phd_conv_fwd_fp8(const p2phd_conv_desc* c, const void* x8, const void* packed8, const float* bias, int act,
                                  void* y, float* stats, void* workspace, void* stream) {
  if (int rc = check_desc(c)) return rc;
  P2PHD_REQUIRE(fp8_eligible(c), "conv_fwd_fp8: layer not eligible");
  P2PHD_REQUIRE(act >= P2PHD_ACT_NONE && act <= P2PHD_ACT_RELU, "conv_fwd_fp8: bad activation %d", act);
  P2PHD_REQUIRE(stats == nullptr || act == P2PHD_ACT_NONE, "conv_fwd_fp8: statistics are taken of the pre-activation output");
  if (c->N == 0) return P2PHD_OK;
  P2PHD_REQUIRE(x8 && packed8 && y && (stats == nullptr || workspace), "conv_fwd_fp8: null pointer");
  hipStream_t st = (hipStream_t)stream;
  int Ho, Wo;
  out_size(c, &Ho, &Wo);
  Plan p = direct_plan(c->N, c->H, c->W, c->C, Ho, Wo, c->K, c->R, c->S, c->stride, c->pad, c->pad_mode);
  p.d.KK = c->R * c->S * c->C;
  p.d.act = act;
  p.d.out_scale = reinterpret_cast<const float*>(static_cast<const char*>(packed8) + fp8_weight_bytes(c));
  float* table = static_cast<float*>(workspace);
  int slot_rows = 0;
  if (int rc = launch_gconv(p.d, P2PHD_FP8_INTERNAL, x8, packed8, bias, nullptr, y, stats ? table : nullptr, st, &slot_rows)) return rc;
  if (stats == nullptr) return P2PHD_OK;
  const long npix = (long)p.d.Hg * p.d.Wg;
  return launch_stats_merge(table, stats, c->N, (int)((npix + slot_rows - 1) / slot_rows), 1, cpitch(c->K), c->K, npix, slot_rows, st);
}

// ---- reflect-padded 3x3 input gradient whose pair-sum rows / columns were written by the producer of dy ----------------
namespace {
bool reflect3x3_exact(const p2phd_conv_desc* c) {
  return c->pad_mode == 1 && fold_mode(c) != FOLD_OUT && !c->transposed && c->R == 3 && c->S == 3 && c->pad == 1 && c->stride == 1 &&
         c->H >= 4 && c->W >= 4 && !g_opt_reflect_generic;
}
}  // namespace

extern "C" size_t p2phd_conv_reflect_extras_elems(const p2phd_conv_desc* c) {
  if (check_desc(c) != P2PHD_OK || !reflect3x3_exact(c) || c->N == 0) return 0;
  if (p2phd_instnorm_act_bwd_two_pass(c->dtype, c->N, (int64_t)c->H * c->W, c->K)) return 0;   // (only the single-launch backward appends them)
  return (size_t)c->N * (2 * (c->W + 2) + 2 * c->H) * cpitch(c->K);
}

extern "C" int p2phd_conv_dgrad_rx(const p2phd_conv_desc* c, const void* dy, const void* wp, const void* addend, void* dx, void* stream) {
  if (int rc = check_desc(c)) return rc;
  if (c->N == 0) return P2PHD_OK;
  P2PHD_REQUIRE(dy && wp && dx, "conv_dgrad_rx: null pointer");
  P2PHD_REQUIRE(p2phd_conv_reflect_extras_elems(c) > 0, "conv_dgrad_rx: layer has no reflection-extras form (p2phd_conv_reflect_extras_elems)");
  std::vector<Plan> plans, ex; WMap m;
  make_plans(c, 1, plans, &m);
  transposed_plans(c->N, c->H, c->W, c->K, c->H, c->W, c->C, c->R, c->S, 1, 1, ex);
  P2PHD_REQUIRE(ex.size() == 1 && ex[0].d.KK == plans[0].d.KK && ex[0].rows_pad == plans[0].rows_pad, "conv_dgrad_rx: plan mismatch");
  ex[0].d.pad_mode = 3;
  ex[0].d.rx_base = c->N * c->H * c->W;                          // dy [N, H, W, Cp(K)], then the extras [N][2 (W + 2) + 2 H][Cp(K)]
  return launch_gconv(ex[0].d, c->dtype, dy, static_cast<const char*>(wp) + plans[0].w_off * elem_size(c->dtype), nullptr, addend, dx,
                      nullptr, (hipStream_t)stream);
}

extern "C" size_t p2phd_conv_dgrad_workspace_bytes(const p2phd_conv_desc* c) {
  if (check_desc(c) != P2PHD_OK) return 0;
  int Ho, Wo;
  out_size(c, &Ho, &Wo);
  size_t n = padded_dx_bytes(c);
  if (fold_mode(c) == FOLD_OUT) n += folded_dy_bytes(c, Ho, Wo);
  // the exact-grid form of a reflect-padded 3x3 input gradient keeps dy + its pair-sum rows / columns here instead:
  // [N][H + 2][W + 2][Cp(K)] -- K channels, where the padded-grid gradient has C
  if (c->pad_mode == 1 && c->R == 3 && c->S == 3 && c->pad == 1 && c->stride == 1 && !c->transposed)
    n = std::max(n, align256((size_t)c->N * (c->H + 2) * (c->W + 2) * cpitch(c->K) * elem_size(c->dtype)));
  return n;
}

extern "C" int p2phd_conv_dgrad(const p2phd_conv_desc* c, const void* dy, const void* wp, const void* addend, void* dx,
                                void* workspace, void* stream) {
  if (int rc = check_desc(c)) return rc;
  if (c->N == 0) return P2PHD_OK;
  P2PHD_REQUIRE(dy && wp && dx, "conv_dgrad: null pointer");
  hipStream_t st = (hipStream_t)stream;
  std::vector<Plan> plans; WMap m;
  make_plans(c, 1, plans, &m);
  int Ho, Wo;
  out_size(c, &Ho, &Wo);
  const bool reflect = c->pad_mode == 1;
  const bool kfold = fold_mode(c) == FOLD_OUT;
  if (addend == nullptr && c7_out_dgrad_ok(c)) {
    // Conv2d(ngf, 2, 7): the dedicated 2 -> ngf kernel with flipped weights and zero padding + the reflection fold of
    // the 3-pixel frame (c7.hip); no padded-grid tensor, no fold pass over the whole gradient
    size_t n = 0;
    for (auto& p : plans) n += (size_t)p.rows_pad * p.d.KK;
    const char* frag = static_cast<const char*>(wp) + n * elem_size(c->dtype);
    const float* master = reinterpret_cast<const float*>(frag + c7_out_dgrad_packed_elems(c) * elem_size(c->dtype));
    return c7_out_dgrad(c, dy, frag, master, dx, st);
  }
  if (addend == nullptr && march_kind(c, 1)) {
    const void* wf = static_cast<const char*>(wp) + generic_packed_elems(plans) * elem_size(c->dtype);
    return march_run(c, 1, dy, wf, nullptr, dx, nullptr, nullptr, nullptr, nullptr, 0.f, 0.f, 0.f, st);
  }
  if (kfold && dlast_ok(c)) {
    // the discriminator's head: dx in one pass, the 16 taps as the MFMA's K (dlast.hip)
    const void* wg = static_cast<const char*>(wp) + generic_packed_elems(plans) * elem_size(c->dtype);
    return dlast_dgrad(c, dy, wg, addend, dx, nullptr, nullptr, nullptr, 0.f, 0.f, 0.f, st);
  }
  P2PHD_REQUIRE(!(reflect || kfold) || workspace, "conv_dgrad: this layer needs p2phd_conv_dgrad_workspace_bytes of scratch");
  if (reflect && !kfold && !c->transposed && c->R == 3 && c->S == 3 && c->pad == 1 && c->stride == 1 && c->H >= 4 && c->W >= 4 &&
      !g_opt_reflect_generic) {
    // 3x3 behind ReflectionPad2d(1) (the residual trunk, networks.py:231-252): the adjoint of the reflection is moved
    // in front of the GEMM -- dy gets two virtual rows / columns holding the pair sums the mirrored taps need -- and the
    // input gradient runs on the exact H x W grid (pad_mode 2 gather), straight into dx with the skip gradient as
    // addend: no padded-grid tensor (+19.5 % rows at 32 x 16) and no fold pass
    std::vector<Plan> ex;
    transposed_plans(c->N, c->H + 2, c->W + 2, c->K, c->H, c->W, c->C, c->R, c->S, 1, 1, ex);
    P2PHD_REQUIRE(ex.size() == 1 && ex[0].d.KK == plans[0].d.KK && ex[0].rows_pad == plans[0].rows_pad, "conv_dgrad: plan mismatch");
    ex[0].d.pad_mode = 2;
    if (int rc = launch_reflect_expand(c->dtype, dy, workspace, c->N, c->H, c->W, cpitch(c->K), st)) return rc;
    return launch_gconv(ex[0].d, c->dtype, workspace, static_cast<const char*>(wp) + plans[0].w_off * elem_size(c->dtype), nullptr,
                        addend, dx, nullptr, st);
  }
  char* ws = static_cast<char*>(workspace);
  void* dxp = ws;                                   // padded-grid gradient (reflect only)
  const void* src = dy;
  if (kfold) {
    void* dye = ws + padded_dx_bytes(c);
    if (int rc = launch_expand_dy(c->dtype, dy, dye, c->N, Ho, Wo, Wo + c->S - 1, c->K, c->S, st)) return rc;
    src = dye;
  }
  for (auto& p : plans) {
    const char* w = static_cast<const char*>(wp) + p.w_off * elem_size(c->dtype);
    if (int rc = launch_gconv(p.d, c->dtype, src, w, nullptr, reflect ? nullptr : addend, reflect ? dxp : dx, nullptr, st)) return rc;
  }
  if (reflect) return launch_reflect_fold(c->dtype, dxp, addend, dx, c->N, c->H, c->W, cpitch(c->C), c->pad, st);
  return P2PHD_OK;
}

// ---- input gradient with the consumer's InstanceNorm-backward sums fused into its store loop -------------------------
namespace {
// one plain or merged sub-pixel gather-GEMM launch writing dx directly (no reflect fold, no W-fold, not the 7x7 kernel),
// on a tile shape that has the fused store loop (gconv 256x256 does not: see launch_gconv_t)
bool dgrad_bsum_plans(const p2phd_conv_desc* c, std::vector<Plan>& plans) {
  if (check_desc(c) != P2PHD_OK || c->N == 0) return false;
  if (c->pad_mode == 1 || c7_out_dgrad_ok(c)) return false;     // (the output W-fold is fine: its launch writes dx directly too)
  WMap m;
  make_plans(c, 1, plans, &m);
  return plans.size() == 1;
}
}  // namespace

extern "C" int p2phd_conv_dgrad_bsum_ok(const p2phd_conv_desc* c) {
  std::vector<Plan> plans;
  return dgrad_bsum_plans(c, plans) ? 1 : 0;
}

extern "C" int p2phd_conv_dgrad_bsum_pays(const p2phd_conv_desc* c) {
  std::vector<Plan> plans;
  if (!dgrad_bsum_plans(c, plans)) return 0;
  // (deep reductions only: on the generator's 3072..3456-deep layers the two extra reduce launches cost what the tile gains)
  return (gconv_plain_launch_takes_256x256(plans[0].d, c->dtype) && plans[0].d.KK >= 6144) ? 0 : 1;
}

extern "C" size_t p2phd_conv_dgrad_bsum_workspace_bytes(const p2phd_conv_desc* c) {
  std::vector<Plan> plans;
  if (!dgrad_bsum_plans(c, plans)) return 0;
  size_t n = align256(bsum_table_floats(plans[0].d) * sizeof(float));
  if (fold_mode(c) == FOLD_OUT) {                                // + the W-folded dy image of the 1-channel head
    int Ho, Wo;
    out_size(c, &Ho, &Wo);
    n += folded_dy_bytes(c, Ho, Wo);
  }
  if (dlast_ok(c, true)) n = std::max(n, align256(dlast_bsum_table_floats(c) * sizeof(float)));
  return n;
}

extern "C" int p2phd_conv_dgrad_bsum(const p2phd_conv_desc* c, const void* dy, const void* wp, const void* addend, void* dx,
                                     const void* prev_y, const float* prev_stats, int prev_act, float eps, float* bstats,
                                     void* workspace, void* stream) {
  std::vector<Plan> plans;
  P2PHD_REQUIRE(dgrad_bsum_plans(c, plans), "conv_dgrad_bsum: this layer's input gradient has no fused-sums form (p2phd_conv_dgrad_bsum_ok)");
  P2PHD_REQUIRE(dy && wp && dx && prev_y && prev_stats && bstats && workspace, "conv_dgrad_bsum: null pointer");
  P2PHD_REQUIRE(prev_act == P2PHD_ACT_NONE || prev_act == P2PHD_ACT_RELU || prev_act == P2PHD_ACT_LRELU, "conv_dgrad_bsum: activation %d", prev_act);
  hipStream_t st = (hipStream_t)stream;
  Plan& p = plans[0];
  if (addend == nullptr && march_kind(c, 1)) {
    // marching kernel with the producer's sums riding on its store pass; one partial row per workgroup
    const void* wf = static_cast<const char*>(wp) + generic_packed_elems(plans) * elem_size(c->dtype);
    const float slope = prev_act == P2PHD_ACT_RELU ? 0.f : (prev_act == P2PHD_ACT_LRELU ? 0.2f : 1.f);
    float* part = static_cast<float*>(workspace);
    if (int rc = march_run(c, 1, dy, wf, nullptr, dx, nullptr, prev_y, prev_stats, part, 1.f / ((float)c->H * (float)c->W), eps, slope, st)) return rc;
    int tiles = 0;
    march_plan(c, 1, nullptr, nullptr, nullptr, nullptr, &tiles);
    const long npix = (long)c->H * c->W;
    return launch_bsum_merge(part, bstats, c->N, npix, (int)(npix / tiles), cpitch(c->C), 0, cpitch(c->C), c->C, st);
  }
  if (dlast_ok(c)) {
    // the discriminator's head (dlast.hip): one partial row per wave (slot) and sample
    const void* wg = static_cast<const char*>(wp) + generic_packed_elems(plans) * elem_size(c->dtype);
    const float slope = prev_act == P2PHD_ACT_RELU ? 0.f : (prev_act == P2PHD_ACT_LRELU ? 0.2f : 1.f);
    float* part = static_cast<float*>(workspace);
    if (int rc = dlast_dgrad(c, dy, wg, addend, dx, prev_y, prev_stats, part, 1.f / ((float)c->H * (float)c->W), eps, slope, st)) return rc;
    int bpw = 0;
    dlast_dgrad_plan(c, &bpw, nullptr);
    return launch_bsum_merge(part, bstats, c->N, (long)c->H * c->W, bpw * 16, cpitch(c->C), 0, cpitch(c->C), c->C, st);
  }
  p.d.bs_y = prev_y;
  p.d.bs_stats = prev_stats;
  p.d.bs_out = static_cast<float*>(workspace);
  p.d.bs_inv_hw = 1.f / ((float)c->H * (float)c->W);            // dx has the conv INPUT's geometry [N, H, W, C]
  p.d.bs_eps = eps;
  p.d.bs_slope = prev_act == P2PHD_ACT_RELU ? 0.f : (prev_act == P2PHD_ACT_LRELU ? 0.2f : 1.f);
  const char* w = static_cast<const char*>(wp) + p.w_off * elem_size(c->dtype);
  const void* src = dy;
  if (fold_mode(c) == FOLD_OUT) {
    int Ho, Wo;
    out_size(c, &Ho, &Wo);
    void* dye = static_cast<char*>(workspace) + align256(bsum_table_floats(p.d) * sizeof(float));
    if (int rc = launch_expand_dy(c->dtype, dy, dye, c->N, Ho, Wo, Wo + c->S - 1, c->K, c->S, st)) return rc;
    src = dye;
  }
  int tile_rows = 0;
  if (int rc = launch_gconv(p.d, c->dtype, src, w, nullptr, addend, dx, nullptr, st, &tile_rows)) return rc;
  const int n_extent = p.d.n_extent ? p.d.n_extent : p.d.Cp_out;
  return launch_bsum_merge(p.d.bs_out, bstats, c->N, (long)p.d.Hg * p.d.Wg, tile_rows, n_extent, p.d.cls_cp, cpitch(c->C), c->C, st);
}

extern "C" int p2phd_conv_dgrad_act(const p2phd_conv_desc* c, const void* dy, const void* wp, const void* addend, void* dx,
                                    const void* x_act, int prev_act, void* workspace, void* stream) {
  std::vector<Plan> plans;
  P2PHD_REQUIRE(dgrad_bsum_plans(c, plans), "conv_dgrad_act: this layer's input gradient has no fused form (p2phd_conv_dgrad_bsum_ok)");
  P2PHD_REQUIRE(dy && wp && dx && x_act, "conv_dgrad_act: null pointer");
  P2PHD_REQUIRE(prev_act == P2PHD_ACT_RELU || prev_act == P2PHD_ACT_LRELU, "conv_dgrad_act: activation %d", prev_act);
  hipStream_t st = (hipStream_t)stream;
  Plan& p = plans[0];
  p.d.as_x = x_act;
  p.d.bs_slope = prev_act == P2PHD_ACT_RELU ? 0.f : 0.2f;
  const char* w = static_cast<const char*>(wp) + p.w_off * elem_size(c->dtype);
  const void* src = dy;
  if (fold_mode(c) == FOLD_OUT) {
    P2PHD_REQUIRE(workspace, "conv_dgrad_act: this layer needs p2phd_conv_dgrad_bsum_workspace_bytes of scratch");
    int Ho, Wo;
    out_size(c, &Ho, &Wo);
    void* dye = static_cast<char*>(workspace) + align256(bsum_table_floats(p.d) * sizeof(float));
    if (int rc = launch_expand_dy(c->dtype, dy, dye, c->N, Ho, Wo, Wo + c->S - 1, c->K, c->S, st)) return rc;
    src = dye;
  }
  return launch_gconv(p.d, c->dtype, src, w, nullptr, addend, dx, nullptr, st);
}

namespace {
// everything p2phd_conv_wgrad needs, derived once for both the workspace query and the call
struct WgradSetup { Plan p; WMap m; int M; int Cp_r; int fold; size_t dwp_bytes; };
void wgrad_setup(const p2phd_conv_desc* c, WgradSetup* w) {
  int Ho, Wo;
  out_size(c, &Ho, &Wo);
  const long RS = (long)c->R * c->S;
  w->fold = fold_mode(c);
  if (w->fold == FOLD_OUT) {       // dWp[(tw,k)][th][c] = sum dyE[n,ho,w',(tw,k)] * x[n, ho+th-pad, w'-pad, c]
    w->p = kfold_fwd_plan(c, Ho, Wo); w->m = kfold_rows_map(c); w->M = c->S * c->K; w->Cp_r = cpitch(w->M);
  } else if (w->fold == FOLD_IN) { // dWp[k][th][(tw,c)] = sum dy[n,ho,wo,k] * Xe[n, ho+th-pad, wo, (tw,c)]
    w->p = cfold_fwd_plan(c, Ho, Wo); w->m = cfold_map(c); w->M = c->K; w->Cp_r = cpitch(c->K);
  } else if (!c->transposed) {     // dW[k][c][r][s] = sum dy[n,ho,wo,k] * x[n, ho*s-pad+r, wo*s-pad+s', c]
    w->p = direct_plan(c->N, c->H, c->W, c->C, Ho, Wo, c->K, c->R, c->S, c->stride, c->pad, c->pad_mode);
    w->m = plain_map(c->K, c->C, c->C * RS, RS, c->S); w->M = c->K; w->Cp_r = cpitch(c->K);
    if (c->w_layout == 1) w->m = plain_map(c->K, c->C, c->C * RS, 1, c->S, c->C);
  } else {                         // dW[ci][co][r][s] = sum x[n,i,j,ci] * dy[n, i*s-pad+r, j*s-pad+s', co]
    w->p = direct_plan(c->N, Ho, Wo, c->K, c->H, c->W, c->C, c->R, c->S, c->stride, c->pad, 0);
    w->m = plain_map(c->C, c->K, c->K * RS, RS, c->S); w->M = c->C; w->Cp_r = cpitch(c->C);
  }
  w->dwp_bytes = align256(wgrad_workspace_floats(w->p.d, c->dtype, w->M, round_up(w->M, 128)) * sizeof(float));
}
}  // namespace

extern "C" size_t p2phd_conv_wgrad_workspace_bytes(const p2phd_conv_desc* c) {
  if (check_desc(c) != P2PHD_OK) return 0;
  int Ho, Wo;
  out_size(c, &Ho, &Wo);
  if (thin_wgrad_kind(c)) return align256(thin_wgrad_workspace_floats(c) * sizeof(float));
  WgradSetup w;
  wgrad_setup(c, &w);
  size_t extra = 0;
  if (w.fold == FOLD_OUT) extra = folded_dy_bytes(c, Ho, Wo);
  else if (w.fold == FOLD_IN) extra = folded_x_bytes(c, Wo);
  // (the marching weight-gradient kernel keeps one slab per workgroup; sized for it whatever the option says at the moment)
  const size_t mw = align256(march_w_workspace_floats(c) * sizeof(float));
  return std::max(w.dwp_bytes + extra, mw);
}

static int conv_wgrad_impl(const p2phd_conv_desc* c, const void* x, const void* dy, float* dw, float* db, int accumulate,
                           void* workspace, void* stream) {
  if (int rc = check_desc(c)) return rc;
  P2PHD_REQUIRE(x && dy && dw && workspace, "conv_wgrad: null pointer");
  int Ho, Wo;
  out_size(c, &Ho, &Wo);
  hipStream_t st = (hipStream_t)stream;
  if (c->dtype == P2PHD_BF16 && march_w_ok(c)) {                  // the generator's outermost stride-2 layers (march.hip)
    if (int rc = march_w_run(c, x, dy, dw, accumulate, nullptr, 1.f, 0.f, static_cast<float*>(workspace), st)) return rc;
    if (db != nullptr) return launch_colsum(c->dtype, dy, (long)c->N * Ho * Wo, cpitch(c->K), c->K, db, accumulate, st);
    return P2PHD_OK;
  }
  if (thin_wgrad_kind(c)) {                                       // dedicated kernel for the <= 4-channel layers (thinwgrad.hip)
    if (int rc = thin_wgrad(c, x, dy, dw, accumulate, static_cast<float*>(workspace), st)) return rc;
    if (db != nullptr) return launch_colsum(c->dtype, dy, (long)c->N * Ho * Wo, cpitch(c->K), c->K, db, accumulate, st);
    return P2PHD_OK;
  }
  WgradSetup w;
  wgrad_setup(c, &w);
  float* dwp = static_cast<float*>(workspace);
  char* extra = static_cast<char*>(workspace) + w.dwp_bytes;
  const void *rows_t, *gat_t;
  if (w.fold == FOLD_OUT) {
    if (int rc = launch_expand_dy(c->dtype, dy, extra, c->N, Ho, Wo, Wo + c->S - 1, c->K, c->S, st)) return rc;
    rows_t = extra; gat_t = x;
  } else if (w.fold == FOLD_IN) {
    if (int rc = launch_expand_in(c->dtype, x, extra, c->N, c->H, c->W, Wo, c->C, c->S, c->pad, c->pad_mode, st)) return rc;
    rows_t = dy; gat_t = extra;
  } else if (!c->transposed) {
    rows_t = dy; gat_t = x;
  } else {
    rows_t = x; gat_t = dy;
  }
  if (int rc = launch_wgrad(w.p.d, w.m, c->dtype, rows_t, w.Cp_r, w.M, round_up(w.M, 128), gat_t, dwp, dw, accumulate, st)) return rc;
  if (db != nullptr) return launch_colsum(c->dtype, dy, (long)c->N * Ho * Wo, cpitch(c->K), c->K, db, accumulate, st);
  return P2PHD_OK;
}

extern "C" int p2phd_conv_wgrad(const p2phd_conv_desc* c, const void* x, const void* dy, float* dw, float* db,
                                void* workspace, void* stream) {
  return conv_wgrad_impl(c, x, dy, dw, db, 0, workspace, stream);
}

extern "C" int p2phd_conv_wgrad_acc(const p2phd_conv_desc* c, const void* x, const void* dy, float* dw, float* db,
                                    void* workspace, void* stream) {
  return conv_wgrad_impl(c, x, dy, dw, db, 1, workspace, stream);
}


// ---- lazily normalised input (round 4) ---------------------------------------------------------------------------------
// A layer whose forward and weight gradient run on the marching kernels can take the RAW output of the InstanceNorm block in
// front of it (pre-normalisation y + its statistics) and apply (y - mean) * rstd and the activation while it stages rows:
// the p2phd_instnorm_act_fwd pass over that plane never runs.  Values are those of the materialised form, bit for bit.
extern "C" int p2phd_conv_lazy_ok(const p2phd_conv_desc* c) {
  if (c == nullptr || check_desc(c) != P2PHD_OK || c->N == 0) return 0;
  return (c->dtype == P2PHD_BF16 && march_kind(c, 0) != 0 && march_w_ok(c)) ? 1 : 0;
}

extern "C" int p2phd_conv_fwd_lazy(const p2phd_conv_desc* c, const void* x_raw, const float* x_stats, int x_act, float x_eps,
                                   const void* wp, const float* bias, void* y, float* stats, void* workspace, void* stream) {
  if (int rc = check_desc(c)) return rc;
  P2PHD_REQUIRE(p2phd_conv_lazy_ok(c), "conv_fwd_lazy: this layer cannot normalise its input on load (p2phd_conv_lazy_ok)");
  P2PHD_REQUIRE(x_raw && x_stats && wp && y && (stats == nullptr || workspace), "conv_fwd_lazy: null pointer");
  P2PHD_REQUIRE(x_act == P2PHD_ACT_NONE || x_act == P2PHD_ACT_RELU || x_act == P2PHD_ACT_LRELU, "conv_fwd_lazy: activation %d", x_act);
  hipStream_t st = (hipStream_t)stream;
  std::vector<Plan> plans; WMap m;
  make_plans(c, 0, plans, &m);
  int Ho, Wo;
  out_size(c, &Ho, &Wo);
  const float slope = x_act == P2PHD_ACT_RELU ? 0.f : (x_act == P2PHD_ACT_LRELU ? 0.2f : 1.f);
  const void* wf = static_cast<const char*>(wp) + generic_packed_elems(plans) * elem_size(c->dtype);
  float* table = static_cast<float*>(workspace);
  if (int rc = march_run(c, 0, x_raw, wf, bias, y, stats ? table : nullptr, nullptr, nullptr, nullptr, 0.f, 0.f, 0.f, st, x_stats, slope, x_eps)) return rc;
  if (stats == nullptr) return P2PHD_OK;
  int slots = 0, ncls = 1, slot_rows = 0;
  long npix_cls = 0;
  march_plan(c, 0, &slots, &ncls, &slot_rows, &npix_cls, nullptr);
  return launch_stats_merge(table, stats, c->N, slots, ncls, cpitch(c->K), c->K, npix_cls, slot_rows, st);
}

extern "C" int p2phd_conv_wgrad_lazy(const p2phd_conv_desc* c, const void* x_raw, const float* x_stats, int x_act, float x_eps,
                                     const void* dy, float* dw, float* db, int accumulate, void* workspace, void* stream) {
  if (int rc = check_desc(c)) return rc;
  P2PHD_REQUIRE(p2phd_conv_lazy_ok(c), "conv_wgrad_lazy: this layer cannot normalise its input on load (p2phd_conv_lazy_ok)");
  P2PHD_REQUIRE(x_raw && x_stats && dy && dw && workspace, "conv_wgrad_lazy: null pointer");
  P2PHD_REQUIRE(x_act == P2PHD_ACT_NONE || x_act == P2PHD_ACT_RELU || x_act == P2PHD_ACT_LRELU, "conv_wgrad_lazy: activation %d", x_act);
  hipStream_t st = (hipStream_t)stream;
  int Ho, Wo;
  out_size(c, &Ho, &Wo);
  const float slope = x_act == P2PHD_ACT_RELU ? 0.f : (x_act == P2PHD_ACT_LRELU ? 0.2f : 1.f);
  if (int rc = march_w_run(c, x_raw, dy, dw, accumulate, x_stats, slope, x_eps, static_cast<float*>(workspace), st)) return rc;
  if (db != nullptr) return launch_colsum(c->dtype, dy, (long)c->N * Ho * Wo, cpitch(c->K), c->K, db, accumulate, st);
  return P2PHD_OK;
}
