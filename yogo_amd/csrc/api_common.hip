#include "common.h"
#include <cstdarg>

static thread_local char g_err[512] = "";

void yogo_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* yogo_hip_last_error(void) { return g_err; }

extern "C" int yogo_hip_abi_version(void) { return 1; }
