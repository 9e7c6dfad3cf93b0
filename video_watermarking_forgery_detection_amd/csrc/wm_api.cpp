// Error plumbing + version for libwm_hip.so (host only).
#include <stdarg.h>
#include <stdio.h>
#include "wm_common.h"

static thread_local char g_err[512] = "";

void wm_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* wm_last_error_string(void) { return g_err; }
extern "C" int wm_abi_version(void) { return 1; }
