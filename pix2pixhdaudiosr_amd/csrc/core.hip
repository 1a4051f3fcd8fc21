// Error channel, ABI version and device query of libp2phd_hip.so.
#include "common.h"
#include "convplan.h"
#include <cstring>

namespace p2phd {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace p2phd

namespace p2phd { int g_opt_gconv_bm = 0; int g_opt_wgrad_tm = 0; int g_opt_wgrad_xcd = 1; int g_opt_march = 1; int g_opt_cls_skip = 1; int g_opt_gconv_halo = 1; int g_opt_cw_inject = 0; int g_opt_tile128x192 = 1; int g_opt_mdct_generic = 0; int g_opt_mdct_iters = 0; int g_opt_c7_generic = 0; int g_opt_c7_abl = 0; int g_opt_reflect_generic = 0; int g_opt_splitk_tail = 1; int g_opt_cus = 0;
unsigned long long g_launch_count[LC_FAMILIES] = {};
int device_cus() {
  static int cached[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  if (cached[dev] == 0) {
    hipDeviceProp_t prop;
    cached[dev] = hipGetDeviceProperties(&prop, dev) == hipSuccess ? prop.multiProcessorCount : 256;
  }
  return cached[dev];
}
}

// scratch of the fixed-order cross-workgroup reductions (common.h): zero-initialised with the code object
namespace {
constexpr size_t kFoldFloats[5] = {size_t(4) << 20, size_t(1) << 20, size_t(1) << 20, 4096, size_t(16) << 20};
constexpr int kFoldTickets[5] = {2048, 1024, 8, 8, 256};
__device__ float g_fold_part[(size_t(22) << 20) + 4096];      // 88 MiB, zero-initialised with the code object
__device__ unsigned g_fold_ticket[2048 + 1024 + 16 + 256];
}  // namespace
namespace p2phd {
namespace {
struct FoldDev { float* part = nullptr; unsigned* ticket = nullptr; hipStream_t last[5] = {}; bool used[5] = {}; hipEvent_t ev[5] = {}; bool ev_set[5] = {}; };
FoldDev g_fold_dev[64];
thread_local hipStream_t g_fold_pending_stream = nullptr;
FoldDev* fold_dev() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  FoldDev& f = g_fold_dev[dev];
  if (f.part == nullptr) {                                      // (the symbol has one address per device)
    void* p = nullptr; void* t = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_fold_part)) != hipSuccess || hipGetSymbolAddress(&t, HIP_SYMBOL(g_fold_ticket)) != hipSuccess)
      return nullptr;
    f.part = static_cast<float*>(p); f.ticket = static_cast<unsigned*>(t);
  }
  return &f;
}
bool capturing(hipStream_t s) {
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  return hipStreamIsCapturing(s, &cs) == hipSuccess && cs == hipStreamCaptureStatusActive;
}
}  // namespace

thread_local int g_fold_pending = -1;

// The region's partial rows and tickets, for a launch on `stream`.  One region serves ONE stream at a time.  Round 5 (advisor):
// the region's last launch is remembered as an EVENT recorded behind it (fold_launched), never as a stream handle to be
// queried later -- the handle may belong to a stream that has been destroyed since, and hipStreamQuery on it is undefined.
// A launch that arrives on another stream makes that stream wait for the event: the two launches of the family are ordered
// on the device.  Streams under capture enqueue nothing (their launches run in graph order when the graph is replayed, and a
// capture is entered behind a synchronisation), so they neither wait nor record.
FoldScratch fold_scratch(int region, hipStream_t stream) {
  FoldDev* f = fold_dev();
  if (f == nullptr || region < 0 || region >= 5) { set_error("reduction scratch unavailable"); return FoldScratch{nullptr, nullptr, 0, 0}; }
  const bool cap = capturing(stream);
  if (f->used[region] && f->last[region] != stream && f->ev_set[region] && !cap) {
    if (hipStreamWaitEvent(stream, f->ev[region], 0) != hipSuccess) {
      (void)hipGetLastError();
      set_error("reduction-scratch region %d: cannot order this stream behind the region's previous launch "
                "(include/p2phd.h, \"Streams\")", region);
      return FoldScratch{nullptr, nullptr, 0, 0};
    }
  }
  f->used[region] = true;
  f->last[region] = stream;
  g_fold_pending = cap ? -1 : region;
  g_fold_pending_stream = stream;
  size_t fo = 0; int to = 0;
  for (int r = 0; r < region; ++r) { fo += kFoldFloats[r]; to += kFoldTickets[r]; }
  return FoldScratch{f->part + fo, f->ticket + to, kFoldFloats[region], kFoldTickets[region]};
}

void fold_launched() {
  const int region = g_fold_pending;
  g_fold_pending = -1;
  FoldDev* f = fold_dev();
  if (f == nullptr || region < 0 || region >= 5) return;
  if (f->ev[region] == nullptr && hipEventCreateWithFlags(&f->ev[region], hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); f->ev[region] = nullptr; return; }
  if (hipEventRecord(f->ev[region], g_fold_pending_stream) == hipSuccess) f->ev_set[region] = true;
  else (void)hipGetLastError();
}
}  // namespace p2phd

// Re-arms every arrival ticket of the fixed-order reductions (an aborted or faulted launch may have left one non-zero, which
// would make every later launch of that kernel family fold at the wrong moment): a 13 KB memset on `stream`; the Python
// mirror issues it once per training step beside its own per-step memset (_ops.begin_step).
extern "C" int p2phd_reduction_reset(void* stream) {
  void* t = nullptr;
  if (hipGetSymbolAddress(&t, HIP_SYMBOL(g_fold_ticket)) != hipSuccess) { p2phd::set_error("reduction_reset: no device symbol"); return P2PHD_ELAUNCH; }
  if (hipMemsetAsync(t, 0, sizeof(unsigned) * (2048 + 1024 + 16 + 256), (hipStream_t)stream) != hipSuccess) {
    p2phd::set_error("reduction_reset: memset failed");
    return P2PHD_ELAUNCH;
  }
  return P2PHD_OK;
}

extern "C" int p2phd_set_option(const char* name, int value) {
  if (name && !strcmp(name, "gconv_bm") && (value == 0 || value == 128 || value == 192 || value == 256 || value == 258 || value == 512)) { p2phd::g_opt_gconv_bm = value; return P2PHD_OK; }
  if (name && !strcmp(name, "wgrad_tm") && (value == 0 || value == 128)) { p2phd::g_opt_wgrad_tm = value; return P2PHD_OK; }
  if (name && !strcmp(name, "march") && (value == 0 || value == 1)) { p2phd::g_opt_march = value; return P2PHD_OK; }
  if (name && !strcmp(name, "cls_skip") && (value == 0 || value == 1)) { p2phd::g_opt_cls_skip = value; return P2PHD_OK; }   // (changes the packed layout: repack after a change)
  if (name && !strcmp(name, "gconv_halo") && (value == 0 || value == 1)) { p2phd::g_opt_gconv_halo = value; return P2PHD_OK; }
  if (name && !strcmp(name, "tile128x192") && (value == 0 || value == 1)) { p2phd::g_opt_tile128x192 = value; return P2PHD_OK; }   // (A/B: the 128 x 192 tile of small planes vs 128 x 128)
  if (name && !strcmp(name, "dlast") && (value == 0 || value == 1)) { p2phd::g_opt_dlast = value; return P2PHD_OK; }     // (A/B, parity tests: dlast.hip vs the W-fold path)
  if (name && !strcmp(name, "dfirst") && (value == 0 || value == 1)) { p2phd::g_opt_dfirst = value; return P2PHD_OK; }   // (A/B, parity tests: dfirst.hip vs the W-fold path)
#ifdef P2PHD_CHECK_WAITS
  if (name && !strcmp(name, "cw_inject") && (value == 0 || value == 1)) { p2phd::g_opt_cw_inject = value; return P2PHD_OK; }   // (check build only: see p2phd_wait_check)
#endif
  if (name && !strcmp(name, "wgrad_xcd") && (value == 0 || value == 1)) { p2phd::g_opt_wgrad_xcd = value; return P2PHD_OK; }
  if (name && !strcmp(name, "reflect_generic") && (value == 0 || value == 1)) { p2phd::g_opt_reflect_generic = value; return P2PHD_OK; }
  if (name && !strcmp(name, "mdct_iters") && value >= 0 && value <= 8) { p2phd::g_opt_mdct_iters = value; return P2PHD_OK; }
  if (name && !strcmp(name, "splitk_tail") && value >= 0 && value <= 2) { p2phd::g_opt_splitk_tail = value; return P2PHD_OK; }
  if (name && !strcmp(name, "cus") && value >= 0 && value <= 4096) { p2phd::g_opt_cus = value; return P2PHD_OK; }
  if (name && !strcmp(name, "c7_abl")) { p2phd::g_opt_c7_abl = value; return P2PHD_OK; }
  if (name && !strcmp(name, "c7_generic") && (value == 0 || value == 1)) { p2phd::g_opt_c7_generic = value; return P2PHD_OK; }
  if (name && !strcmp(name, "mdct_generic") && (value == 0 || value == 1)) { p2phd::g_opt_mdct_generic = value; return P2PHD_OK; }
  p2phd::set_error("set_option: unknown option or value (%s = %d)", name ? name : "(null)", value);
  return P2PHD_EINVAL;
}

extern "C" int64_t p2phd_launch_count(const char* family, int reset) {
  static const char* names[p2phd::LC_FAMILIES] = {"gconv", "halo", "cls_skip", "march", "march_w", "wgrad", "splitk", "tile256", "tile128x192"};
  if (family == nullptr) {                                       // all families at once
    if (reset) for (auto& c : p2phd::g_launch_count) c = 0;
    return 0;
  }
  for (int i = 0; i < p2phd::LC_FAMILIES; ++i)
    if (!strcmp(family, names[i])) {
      const int64_t v = (int64_t)p2phd::g_launch_count[i];
      if (reset) p2phd::g_launch_count[i] = 0;
      return v;
    }
  p2phd::set_error("launch_count: unknown kernel family %s", family);
  return -1;
}

extern "C" const char* p2phd_last_error(void) { return p2phd::g_err; }
extern "C" int p2phd_abi_version(void) { return 1; }
// 1: this library's 16-bit storage type (dtype code P2PHD_BF16) is bf16; 2: IEEE fp16 (the -DP2PHD_F16 build, libp2phd_hip_f16.so)
extern "C" int p2phd_half_type(void) {
#ifdef P2PHD_F16
  return 2;
#else
  return 1;
#endif
}

extern "C" int p2phd_device_info(char* name, int cap) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) {
    p2phd::set_error("no HIP device");
    return P2PHD_ELAUNCH;
  }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) != hipSuccess) {
    p2phd::set_error("hipGetDeviceProperties failed");
    return P2PHD_ELAUNCH;
  }
  if (name && cap > 0) {
    strncpy(name, prop.gcnArchName, cap - 1);
    name[cap - 1] = 0;
  }
  return prop.multiProcessorCount;
}
