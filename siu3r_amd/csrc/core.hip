// Error reporting + ABI version for libsiu3r_hip.so.
#include <stdarg.h>

#include "common.h"

static thread_local char g_err[512] = "";

void siu3r_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* siu3r_last_error(void) { return g_err; }
extern "C" int siu3r_abi_version(void) { return SIU3R_ABI_VERSION; }
