// Host-side plumbing of the C ABI: thread-local error string + version.
#include <stdarg.h>
#include <stdio.h>
#include "common.h"
#include "../../include/speechclip_hip.h"

static thread_local char g_err[512] = "";

void sc_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* sc_last_error(void) { return g_err; }
extern "C" int sc_abi_version(void) { return 1; }
