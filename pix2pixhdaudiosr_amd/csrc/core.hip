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

namespace p2phd { int g_opt_gconv_bm = 0; int g_opt_wgrad_tm = 0; int g_opt_mdct_generic = 0; int g_opt_c7_generic = 0; int g_opt_c7_abl = 0; int g_opt_reflect_generic = 0; }

extern "C" int p2phd_set_option(const char* name, int value) {
  if (name && !strcmp(name, "gconv_bm") && (value == 0 || value == 128 || value == 192 || value == 256 || value == 512)) { p2phd::g_opt_gconv_bm = value; return P2PHD_OK; }
  if (name && !strcmp(name, "wgrad_tm") && (value == 0 || value == 128)) { p2phd::g_opt_wgrad_tm = value; return P2PHD_OK; }
  if (name && !strcmp(name, "reflect_generic") && (value == 0 || value == 1)) { p2phd::g_opt_reflect_generic = value; return P2PHD_OK; }
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
