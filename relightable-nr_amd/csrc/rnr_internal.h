// Internal helpers shared by the HIP translation units of librnr_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/rnr_hip.h"

namespace rnr {

// thread-local last-error text (rnr_last_error)
char* err_buf();
int fail(const char* fmt, ...);

inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail("%s: %s", what, hipGetErrorString(e));
    return 0;
}

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

#define RNR_REQUIRE(cond, ...)                     \
    do {                                           \
        if (!(cond)) return rnr::fail(__VA_ARGS__); \
    } while (0)

#define RNR_HIP(call)                                                                  \
    do {                                                                               \
        hipError_t e__ = (call);                                                       \
        if (e__ != hipSuccess) return rnr::fail("%s: %s", #call, hipGetErrorString(e__)); \
    } while (0)

}  // namespace rnr
