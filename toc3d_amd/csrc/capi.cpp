// Error reporting + version probe of the C ABI (no HIP context is created at load time).
#include <cstdarg>
#include <cstdio>

#include "capi.h"

namespace {
thread_local char g_err[512] = "";
}

void toc3d_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" {
int toc3d_abi_version(void) { return TOC3D_ABI_VERSION; }
const char* toc3d_last_error(void) { return g_err; }
}
