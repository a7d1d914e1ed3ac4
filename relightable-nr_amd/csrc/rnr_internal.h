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

// raster.hip: the regions of an rnr_rasterize_gbuffer workspace that are cleared per call (for rnr_frame_prepare)
void gbuffer_clear_regions(void* workspace, int num_views, int num_faces, int image_size, uint4** counters, long* counter_vec,
                           uint4** keys, long* key_vec);

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

#if defined(__HIPCC__)
// tanh(x) + 1 = 2 - 2 / (e^{2x} + 1) on v_exp_f32 / v_rcp_f32 (the light transport factor of the ray renderer): absolute
// error ~1e-7, exact limits at +-inf; ocml's tanhf costs ~25 instructions and this kernel is VALU-bound.
__device__ __forceinline__ float fast_tanh_plus1f(float x) {
    const float e = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);
    return __builtin_fmaf(-2.0f, __builtin_amdgcn_rcpf(e + 1.0f), 2.0f);
}
#endif

}  // namespace rnr
