// Host-side helpers shared by every translation unit of libtecmollm_hip.so.
#include "common.h"
#include <string.h>

static thread_local char g_err[512] = "no error";

void tecm_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* tecm_last_error(void) { return g_err; }
extern "C" int tecm_abi_version(void) { return TECM_ABI_VERSION; }
