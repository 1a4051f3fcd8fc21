// C ABI of the convolution family: turns a reference-level layer description (Conv2d / ConvTranspose2d with
// stride, padding and an optional ReflectionPad2d in front) into gather-convolution launches (conv.hip).
#include "convplan.h"
#include <vector>

namespace {
using namespace p2phd;

struct Plan {
  GDesc d;
  size_t w_off;      // element offset of this launch's block inside the packed weight buffer
  int rows_pad;
};

int elem_size(int dtype) { return dtype == P2PHD_BF16 ? 2 : 4; }

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
  if (c->transposed) P2PHD_REQUIRE(c->opad >= 0 && c->opad < c->stride, "conv: output_padding must be < stride");
  int Ho, Wo;
  if (c->transposed) {
    Ho = (c->H - 1) * c->stride - 2 * c->pad + c->R + c->opad;
    Wo = (c->W - 1) * c->stride - 2 * c->pad + c->S + c->opad;
  } else {
    Ho = (c->H + 2 * c->pad - c->R) / c->stride + 1;
    Wo = (c->W + 2 * c->pad - c->S) / c->stride + 1;
    P2PHD_REQUIRE(c->H + 2 * c->pad >= c->R && c->W + 2 * c->pad >= c->S, "conv: kernel larger than padded input");
  }
  P2PHD_REQUIRE(Ho >= 1 && Wo >= 1, "conv: empty output");
  return P2PHD_OK;
}

void out_size(const p2phd_conv_desc* c, int* Ho, int* Wo) {
  if (c->transposed) {
    *Ho = (c->H - 1) * c->stride - 2 * c->pad + c->R + c->opad;
    *Wo = (c->W - 1) * c->stride - 2 * c->pad + c->S + c->opad;
  } else {
    *Ho = (c->H + 2 * c->pad - c->R) / c->stride + 1;
    *Wo = (c->W + 2 * c->pad - c->S) / c->stride + 1;
  }
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

// which: 0 = forward, 1 = input gradient
void make_plans(const p2phd_conv_desc* c, int which, std::vector<Plan>& plans, int* rows, int* inner, long* s_row, long* s_inner) {
  int Ho, Wo;
  out_size(c, &Ho, &Wo);
  const long RS = (long)c->R * c->S;
  if (which == 0 && !c->transposed) {
    plans.push_back(direct_plan(c->N, c->H, c->W, c->C, Ho, Wo, c->K, c->R, c->S, c->stride, c->pad, c->pad_mode));
    *rows = c->K; *inner = c->C; *s_row = c->C * RS; *s_inner = RS;
  } else if (which == 0 && c->transposed) {
    transposed_plans(c->N, c->H, c->W, c->C, Ho, Wo, c->K, c->R, c->S, c->stride, c->pad, plans);
    *rows = c->K; *inner = c->C; *s_row = RS; *s_inner = c->K * RS;          // weight [C][K][R][S]
  } else if (which == 1 && !c->transposed) {
    const int P = c->pad_mode == 1 ? c->pad : 0;                               // reflect: gradient on the padded grid
    transposed_plans(c->N, Ho, Wo, c->K, c->H + 2 * P, c->W + 2 * P, c->C, c->R, c->S, c->stride, c->pad_mode == 1 ? 0 : c->pad, plans);
    *rows = c->C; *inner = c->K; *s_row = RS; *s_inner = c->C * RS;          // weight [K][C][R][S]
  } else {
    plans.push_back(direct_plan(c->N, Ho, Wo, c->K, c->H, c->W, c->C, c->R, c->S, c->stride, c->pad, 0));
    *rows = c->C; *inner = c->K; *s_row = c->K * RS; *s_inner = RS;          // weight [C][K][R][S]
  }
}

}  // namespace

extern "C" int p2phd_channel_pitch(int channels) { return p2phd::cpitch(channels); }

extern "C" int p2phd_conv_out_size(const p2phd_conv_desc* c, int* Ho, int* Wo) {
  if (int rc = check_desc(c)) return rc;
  out_size(c, Ho, Wo);
  return P2PHD_OK;
}

extern "C" size_t p2phd_conv_packed_bytes(const p2phd_conv_desc* c, int which) {
  if (check_desc(c) != P2PHD_OK || (which != 0 && which != 1)) return 0;
  std::vector<Plan> plans; int rows, inner; long sr, si;
  make_plans(c, which, plans, &rows, &inner, &sr, &si);
  size_t n = 0;
  for (auto& p : plans) n += (size_t)p.rows_pad * p.d.KK;
  return n * elem_size(c->dtype);
}

extern "C" int p2phd_conv_pack_weights(const p2phd_conv_desc* c, int which, const float* w, void* packed, void* stream) {
  if (int rc = check_desc(c)) return rc;
  P2PHD_REQUIRE(which == 0 || which == 1, "pack_weights: which must be 0 (forward) or 1 (input gradient)");
  P2PHD_REQUIRE(w && packed, "pack_weights: null pointer");
  std::vector<Plan> plans; int rows, inner; long sr, si;
  make_plans(c, which, plans, &rows, &inner, &sr, &si);
  for (auto& p : plans) {
    char* dst = static_cast<char*>(packed) + p.w_off * elem_size(c->dtype);
    if (int rc = launch_pack(p.d, c->dtype, w, dst, rows, p.rows_pad, inner, sr, si, c->S, (hipStream_t)stream)) return rc;
  }
  return P2PHD_OK;
}

extern "C" int p2phd_conv_fwd(const p2phd_conv_desc* c, const void* x, const void* wp, const float* bias, int act,
                              void* y, float* stats, void* stream) {
  if (int rc = check_desc(c)) return rc;
  P2PHD_REQUIRE(act >= P2PHD_ACT_NONE && act <= P2PHD_ACT_RELU, "conv_fwd: bad activation %d", act);
  if (c->N == 0) return P2PHD_OK;
  P2PHD_REQUIRE(x && wp && y, "conv_fwd: null pointer");
  std::vector<Plan> plans; int rows, inner; long sr, si;
  make_plans(c, 0, plans, &rows, &inner, &sr, &si);
  for (auto& p : plans) {
    p.d.act = act;
    const char* w = static_cast<const char*>(wp) + p.w_off * elem_size(c->dtype);
    if (int rc = launch_gconv(p.d, c->dtype, x, w, bias, nullptr, y, stats, (hipStream_t)stream)) return rc;
  }
  return P2PHD_OK;
}

extern "C" size_t p2phd_conv_dgrad_workspace_bytes(const p2phd_conv_desc* c) {
  if (check_desc(c) != P2PHD_OK) return 0;
  if (c->pad_mode != 1) return 0;
  return (size_t)c->N * (c->H + 2 * c->pad) * (c->W + 2 * c->pad) * p2phd::cpitch(c->C) * elem_size(c->dtype);
}

extern "C" int p2phd_conv_dgrad(const p2phd_conv_desc* c, const void* dy, const void* wp, const void* addend, void* dx,
                                void* workspace, void* stream) {
  if (int rc = check_desc(c)) return rc;
  if (c->N == 0) return P2PHD_OK;
  P2PHD_REQUIRE(dy && wp && dx, "conv_dgrad: null pointer");
  std::vector<Plan> plans; int rows, inner; long sr, si;
  make_plans(c, 1, plans, &rows, &inner, &sr, &si);
  const bool reflect = c->pad_mode == 1;
  P2PHD_REQUIRE(!reflect || workspace, "conv_dgrad: reflect padding needs the workspace");
  for (auto& p : plans) {
    const char* w = static_cast<const char*>(wp) + p.w_off * elem_size(c->dtype);
    if (int rc = launch_gconv(p.d, c->dtype, dy, w, nullptr, reflect ? nullptr : addend, reflect ? workspace : dx, nullptr,
                              (hipStream_t)stream)) return rc;
  }
  if (reflect)
    return launch_reflect_fold(c->dtype, workspace, addend, dx, c->N, c->H, c->W, p2phd::cpitch(c->C), c->pad, (hipStream_t)stream);
  return P2PHD_OK;
}

extern "C" size_t p2phd_conv_wgrad_workspace_bytes(const p2phd_conv_desc* c) {
  if (check_desc(c) != P2PHD_OK) return 0;
  const int M = c->transposed ? c->C : c->K;
  const int inner = c->transposed ? c->K : c->C;
  const int KK = std::max(64, p2phd::round_up(c->R * c->S * p2phd::cpitch(inner), 64));
  return (size_t)p2phd::round_up(M, 128) * KK * sizeof(float);
}

extern "C" int p2phd_conv_wgrad(const p2phd_conv_desc* c, const void* x, const void* dy, float* dw, float* db,
                                void* workspace, void* stream) {
  if (int rc = check_desc(c)) return rc;
  P2PHD_REQUIRE(x && dy && dw && workspace, "conv_wgrad: null pointer");
  int Ho, Wo;
  out_size(c, &Ho, &Wo);
  const long RS = (long)c->R * c->S;
  hipStream_t st = (hipStream_t)stream;
  Plan p;
  int rows, inner; long sr, si;
  const void *rows_t, *gat_t;
  int Cp_r;
  if (!c->transposed) {   // dW[k][c][r][s] = sum dy[n,ho,wo,k] * x[n, ho*s-pad+r, wo*s-pad+s', c]
    p = direct_plan(c->N, c->H, c->W, c->C, Ho, Wo, c->K, c->R, c->S, c->stride, c->pad, c->pad_mode);
    rows = c->K; inner = c->C; sr = c->C * RS; si = RS;
    rows_t = dy; gat_t = x; Cp_r = p2phd::cpitch(c->K);
  } else {                // dW[ci][co][r][s] = sum x[n,i,j,ci] * dy[n, i*s-pad+r, j*s-pad+s', co]
    p = direct_plan(c->N, Ho, Wo, c->K, c->H, c->W, c->C, c->R, c->S, c->stride, c->pad, 0);
    rows = c->C; inner = c->K; sr = c->K * RS; si = RS;
    rows_t = x; gat_t = dy; Cp_r = p2phd::cpitch(c->C);
  }
  float* dwp = static_cast<float*>(workspace);
  if (int rc = launch_wgrad(p.d, c->dtype, rows_t, Cp_r, p2phd::round_up(rows, 128), gat_t, dwp, st)) return rc;
  if (int rc = launch_unpack_grad(p.d, dwp, dw, rows, inner, sr, si, c->S, st)) return rc;
  if (db != nullptr) return launch_colsum(c->dtype, dy, (long)c->N * Ho * Wo, p2phd::cpitch(c->K), c->K, db, st);
  return P2PHD_OK;
}
