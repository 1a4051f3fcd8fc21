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

namespace p2phd { int g_opt_gconv_bm = 0; int g_opt_wgrad_tm = 0; int g_opt_mdct_generic = 0; int g_opt_mdct_iters = 0; int g_opt_c7_generic = 0; int g_opt_c7_abl = 0; int g_opt_reflect_generic = 0; int g_opt_splitk_tail = 1; int g_opt_gconv_persist = 0; }

// scratch of the fixed-order cross-workgroup reductions (common.h): zero-initialised with the code object
namespace {
constexpr size_t kFoldFloats[5] = {size_t(4) << 20, size_t(1) << 20, size_t(1) << 20, 4096, size_t(16) << 20};
constexpr int kFoldTickets[5] = {2048, 1024, 8, 8, 256};
__device__ float g_fold_part[(size_t(22) << 20) + 4096];      // 88 MiB, zero-initialised with the code object
__device__ unsigned g_fold_ticket[2048 + 1024 + 16 + 256];
}  // namespace
namespace p2phd {
FoldScratch fold_scratch(int region) {
  static float* parts[64] = {};
  static unsigned* tickets[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return FoldScratch{nullptr, nullptr, 0, 0};
  if (parts[dev] == nullptr) {                                  // (the symbol has one address per device)
    void* p = nullptr; void* t = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_fold_part)) != hipSuccess || hipGetSymbolAddress(&t, HIP_SYMBOL(g_fold_ticket)) != hipSuccess)
      return FoldScratch{nullptr, nullptr, 0, 0};
    parts[dev] = static_cast<float*>(p); tickets[dev] = static_cast<unsigned*>(t);
  }
  float* part = parts[dev];
  unsigned* ticket = tickets[dev];
  size_t fo = 0; int to = 0;
  for (int r = 0; r < region; ++r) { fo += kFoldFloats[r]; to += kFoldTickets[r]; }
  return FoldScratch{part + fo, ticket + to, kFoldFloats[region], kFoldTickets[region]};
}
}  // namespace p2phd

extern "C" int p2phd_set_option(const char* name, int value) {
  if (name && !strcmp(name, "gconv_bm") && (value == 0 || value == 128 || value == 192 || value == 256 || value == 258 || value == 512)) { p2phd::g_opt_gconv_bm = value; return P2PHD_OK; }
  if (name && !strcmp(name, "wgrad_tm") && (value == 0 || value == 128)) { p2phd::g_opt_wgrad_tm = value; return P2PHD_OK; }
  if (name && !strcmp(name, "reflect_generic") && (value == 0 || value == 1)) { p2phd::g_opt_reflect_generic = value; return P2PHD_OK; }
  if (name && !strcmp(name, "mdct_iters") && value >= 0 && value <= 8) { p2phd::g_opt_mdct_iters = value; return P2PHD_OK; }
  if (name && !strcmp(name, "splitk_tail") && value >= 0 && value <= 2) { p2phd::g_opt_splitk_tail = value; return P2PHD_OK; }
  if (name && !strcmp(name, "gconv_persist") && (value == 0 || value == 1)) { p2phd::g_opt_gconv_persist = value; return P2PHD_OK; }
  if (name && !strcmp(name, "c7_abl")) { p2phd::g_opt_c7_abl = value; return P2PHD_OK; }
  if (name && !strcmp(name, "c7_generic") && (value == 0 || value == 1)) { p2phd::g_opt_c7_generic = value; return P2PHD_OK; }
  if (name && !strcmp(name, "mdct_generic") && (value == 0 || value == 1)) { p2phd::g_opt_mdct_generic = value; return P2PHD_OK; }
  p2phd::set_error("set_option: unknown option or value (%s = %d)", name ? name : "(null)", value);
  return P2PHD_EINVAL;
}

extern "C" const char* p2phd_last_error(void) { return p2phd::g_err; }
extern "C" int p2phd_abi_version(void) { return 1; }

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
