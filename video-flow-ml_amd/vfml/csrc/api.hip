// Error channel + ABI version of libvfml_hip.so.
#include "vfml_common.h"
#include <stdarg.h>

namespace {
thread_local char g_err[512] = "";
}

void vfml_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* vfml_last_error(void) { return g_err; }
extern "C" int vfml_abi_version(void) { return VFML_ABI_VERSION; }
