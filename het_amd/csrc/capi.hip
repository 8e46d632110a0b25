// Library-level entry points: error string, build info.
#include <stdarg.h>
#include <string.h>

#include "common.hip.h"

static thread_local char g_err[512] = "";

void het_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* het_last_error(void) { return g_err; }

#ifndef HET_GIT_SHA
#define HET_GIT_SHA "unknown"
#endif

extern "C" const char* het_build_info(void) {
  return "het_amd (libhet_amd.so) git " HET_GIT_SHA " | target gfx950 (MI355X, CDNA4) | hipcc " __VERSION__
         " | built " __DATE__ " " __TIME__;
}
