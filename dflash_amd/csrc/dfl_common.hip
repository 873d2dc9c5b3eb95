// Error string + version (host only).
#include "dfl_common.h"

static thread_local char g_err[512] = "";

void dfl_set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int dfl_version(void) { return DFL_ABI_VERSION; }
extern "C" const char *dfl_last_error(void) { return g_err; }
