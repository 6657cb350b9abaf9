// Error reporting + version probe of the C ABI (no HIP context is created at load time).
#include <cmath>
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
// what toc3d_window_attention_rot's exp2-based softmax needs q to be scaled by (the convention is the library's, not the caller's to guess)
float toc3d_attn_rot_q_scale(int64_t head_dim) { return head_dim > 0 ? (float)(1.4426950408889634 / std::sqrt((double)head_dim)) : 0.0f; }
}
