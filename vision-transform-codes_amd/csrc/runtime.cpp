// Version / error-string plumbing of libvtc_hip.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "../../include/vtc_hip.h"

namespace vtc {
static thread_local char g_error[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_error, sizeof(g_error), fmt, ap);
  va_end(ap);
}
}  // namespace vtc

extern "C" const char* vtc_version(void) {
  return "vtc_hip 0.1 (gfx950)";
}
extern "C" const char* vtc_last_error(void) { return vtc::g_error; }
extern "C" int vtc_abi_version(void) { return VTC_ABI_VERSION; }
