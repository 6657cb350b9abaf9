// Host-side helpers shared by the C-ABI translation units (error reporting, launch checks).
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/toc3d.h"

void toc3d_set_error(const char* fmt, ...);

#define TOC3D_REQUIRE(cond, ...)                 \
    do {                                         \
        if (!(cond)) {                           \
            toc3d_set_error(__VA_ARGS__);        \
            return TOC3D_ERR_ARG;                \
        }                                        \
    } while (0)

#define TOC3D_LAUNCH_CHECK(name)                                                        \
    do {                                                                                \
        hipError_t e_ = hipGetLastError();                                              \
        if (e_ != hipSuccess) {                                                         \
            toc3d_set_error("%s: kernel launch failed: %s", name, hipGetErrorString(e_)); \
            return TOC3D_ERR_LAUNCH;                                                    \
        }                                                                               \
    } while (0)

static inline hipStream_t as_stream(toc3d_stream_t s) { return reinterpret_cast<hipStream_t>(s); }
