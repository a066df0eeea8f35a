// Error reporting and version for libess_hip.so.  No mutable global state besides the thread-local
// last-error buffer.
#include "common.h"

static thread_local char g_err[512] = "";

void ess_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* ess_last_error(void) { return g_err; }
extern "C" int ess_version(void) { return 100; }
