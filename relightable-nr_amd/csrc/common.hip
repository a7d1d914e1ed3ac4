// Error reporting + ABI version of librnr_hip.so.
#include "rnr_internal.h"
#include <string.h>

namespace rnr {

static thread_local char g_err[512] = {0};

char* err_buf() { return g_err; }

int fail(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return 1;
}

}  // namespace rnr

// rnr_calibrate_mfma_f32: what THIS device sustains on the instruction the U-Net runs on, right now (a measurement aid for
// bench.py: MI355X boxes of one pool differ by several per cent in the clock they hold under matrix load, and a frames/s
// figure can only be compared across boxes beside such a number).  Register-resident loop, no memory traffic: four
// independent accumulator chains per wave, `waves_per_simd` waves per SIMD (two sustain the most: r03's micro-benchmark and
// r06: 1 / 2 / 3 / 4 / 8 waves measured in profiles/r06_one_view_frontend.txt), operands with full mantissas (the matrix
// cores' power draw depends on the operand bits).
typedef float rnr_floatx16 __attribute__((ext_vector_type(16)));
__global__ void __launch_bounds__(256) calibrate_mfma_f32_kernel(float* __restrict__ sink, int iters) {
    const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned x = tid * 2654435761u + 12345u;
    x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
    const float a = __builtin_bit_cast(float, (x & 0x007fffffu) | 0x3f800000u) - 1.5f;             // [-0.5, 0.5), 23 random mantissa bits
    const float b = __builtin_bit_cast(float, ((x * 3266489917u) & 0x007fffffu) | 0x3f800000u) - 1.5f;
    rnr_floatx16 acc[4];
    for (int c = 0; c < 4; c++)
        for (int i = 0; i < 16; i++) acc[c][i] = 0.f;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++)
#pragma unroll
            for (int c = 0; c < 4; c++) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
    }
    float s = 0.f;
    for (int c = 0; c < 4; c++) s += acc[c][0] + acc[c][7];
    if (s == 123.456f) sink[0] = s;      // never true for these operands: keeps the chains alive
}

extern "C" int rnr_calibrate_mfma_f32(int iters, int waves_per_simd, float* scratch, double* tflops, double* seconds, void* stream) {
    RNR_REQUIRE(iters > 0 && waves_per_simd >= 1 && waves_per_simd <= 8 && scratch && tflops, "rnr_calibrate_mfma_f32: bad arguments");
    int dev = 0, cus = 0;
    RNR_HIP(hipGetDevice(&dev));
    RNR_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    const int blocks = cus * waves_per_simd;          // 4 waves each, one per SIMD
    hipStream_t st = rnr::as_stream(stream);
    hipEvent_t e0, e1;
    RNR_HIP(hipEventCreate(&e0));
    RNR_HIP(hipEventCreate(&e1));
    RNR_HIP(hipEventRecord(e0, st));
    hipLaunchKernelGGL(calibrate_mfma_f32_kernel, dim3((unsigned)blocks), dim3(256), 0, st, scratch, iters);
    const int rc = rnr::check_launch("calibrate_mfma_f32_kernel");
    RNR_HIP(hipEventRecord(e1, st));
    RNR_HIP(hipEventSynchronize(e1));
    float ms = 0.f;
    RNR_HIP(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (rc) return rc;
    const double flops = (double)blocks * 4.0 * (double)iters * 32.0 * 4096.0;      // 32 x 32 x 2 x 2 per wave instruction
    *tflops = flops / ((double)ms * 1e-3) / 1e12;
    if (seconds) *seconds = (double)ms * 1e-3;
    return 0;
}

extern "C" int rnr_abi_version(void) { return RNR_ABI_VERSION; }
extern "C" const char* rnr_last_error(void) { return rnr::g_err; }
