// Error reporting + version of the C ABI (host only).
#include <cstdarg>
#include <cstdio>

#include "psld_hip.h"

static thread_local char g_err[512] = "";

void psld_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int psld_version(void) { return PSLD_ABI_VERSION; }
extern "C" const char* psld_last_error(void) { return g_err; }
