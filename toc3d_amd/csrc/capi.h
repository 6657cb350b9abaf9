// Host-side helpers shared by the C-ABI translation units (error reporting, launch checks, the launch recorder hook).
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <tuple>
#include <utility>

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

// ---- launch recorder (plan.cpp) ---------------------------------------------------------------------
// Every kernel of the library is launched through toc3d_launch().  Normally that is hipLaunchKernelGGL.  While a host
// thread is between toc3d_plan_begin() and toc3d_plan_end(), its launches are *recorded* instead (kernel, grid, block,
// LDS bytes, a copy of the argument values, and the lane the call's `stream` argument names) and nothing runs; the plan
// then replays the whole frame from one C call (toc3d_plan_run) on HIP streams or as an explicitly built hipGraph.
struct Toc3dPlan;
extern thread_local Toc3dPlan* toc3d_tls_recording;
void toc3d_plan_record(Toc3dPlan* plan, const void* func, dim3 grid, dim3 block, size_t lds, hipStream_t lane_handle,
                       const void* const* arg_ptrs, const size_t* arg_sizes, const size_t* arg_aligns, int nargs);

template <typename... KArgs, size_t... I>
inline void toc3d_record_packed(const void* func, dim3 grid, dim3 block, size_t lds, hipStream_t s, std::tuple<KArgs...>& packed,
                                std::index_sequence<I...>) {
    const void* ptrs[] = {static_cast<const void*>(&std::get<I>(packed))...};
    const size_t sizes[] = {sizeof(KArgs)...};
    const size_t aligns[] = {alignof(KArgs)...};
    toc3d_plan_record(toc3d_tls_recording, func, grid, block, lds, s, ptrs, sizes, aligns, (int)sizeof...(KArgs));
}

template <typename... KArgs, typename... Args>
inline void toc3d_launch(void (*kernel)(KArgs...), dim3 grid, dim3 block, size_t lds, hipStream_t s, Args&&... args) {
    static_assert(sizeof...(KArgs) == sizeof...(Args), "toc3d_launch: argument count does not match the kernel signature");
    if (toc3d_tls_recording) {
        std::tuple<KArgs...> packed{static_cast<KArgs>(args)...};     // the values exactly as the kernel receives them
        toc3d_record_packed(reinterpret_cast<const void*>(kernel), grid, block, lds, s, packed, std::index_sequence_for<KArgs...>{});
    } else {
        hipLaunchKernelGGL(kernel, grid, block, lds, s, static_cast<KArgs>(args)...);
    }
}

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is per function AND per device: one flag per device ordinal, set with a
// relaxed atomic (setting the attribute twice is harmless, so a race only costs a redundant call).
struct Toc3dLdsAttr {
    std::atomic<unsigned long long> done[2] = {};                    // devices 0..127
    void ensure(const void* func, int bytes) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 128) dev = 0;
        const unsigned long long bit = 1ull << (dev & 63);
        std::atomic<unsigned long long>& w = done[dev >> 6];
        if (w.load(std::memory_order_relaxed) & bit) return;
        (void)hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        w.fetch_or(bit, std::memory_order_relaxed);
    }
};
