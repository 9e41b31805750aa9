// Error plumbing + version for libgenesis_hip.so.
#include <stdarg.h>
#include <stdio.h>

#include "gx_common.h"

static thread_local char g_err[512] = "";

void gx_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" {
const char* gx_last_error(void) { return g_err; }
int gx_version(void) { return 1; }
}
