// Thread-local last-error string of the C ABI.
#include <stdarg.h>
#include <stdio.h>
#include "../../include/gmd_hip.h"

static thread_local char g_err[512] = "";

void gmd_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" {
int gmd_abi_version(void) { return GMD_ABI_VERSION; }
const char* gmd_last_error(void) { return g_err; }
}
