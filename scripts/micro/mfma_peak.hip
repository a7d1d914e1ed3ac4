// What MFMA rate does an MI355X SUSTAIN, by instruction and by operand content?  Register-resident loops (no memory traffic
// at all: 4 independent accumulator chains per wave, operands fixed in registers), every SIMD of every CU busy, ~0.2 s
// per case so that the power controller settles.  Gives the denominators bench.py's emulation blocks quote beside the
// nominal peaks: the fp32 emulation kernels are power-bound, so "fraction of 2.5 PFLOP/s / products"
// understates them; "fraction of what the matrix cores sustain under the same operand statistics" is the fair figure.
// Build + run (GPU box): hipcc --offload-arch=gfx950 -O2 scripts/micro/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
// Output: one JSON object (cases: instruction x operand pattern -> TFLOP/s, effective MHz assuming the documented
// passes per instruction).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef _Float16 halfx8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

// MODE 0: v_mfma_f32_32x32x16_f16   1: v_mfma_f32_32x32x16_bf16   2: v_mfma_f32_32x32x2_f32
template <int MODE>
__global__ void __launch_bounds__(256) spin(const unsigned* __restrict__ seed, float* __restrict__ sink, int iters) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned w[4];
    for (int i = 0; i < 4; i++) w[i] = seed[(tid * 4 + i) & 4095];
    floatx16 acc[4];
    for (int c = 0; c < 4; c++)
        for (int i = 0; i < 16; i++) acc[c][i] = 0.f;
    if (MODE == 2) {
        const float a = __builtin_bit_cast(float, w[0]), b = __builtin_bit_cast(float, w[1]);
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int u = 0; u < 8; u++)
#pragma unroll
                for (int c = 0; c < 4; c++) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
        }
    } else if (MODE == 0) {
        struct { unsigned x[4]; } pa = {{w[0], w[1], w[2], w[3]}}, pb = {{w[1], w[2], w[3], w[0]}};
        const halfx8 a = __builtin_bit_cast(halfx8, pa), b = __builtin_bit_cast(halfx8, pb);
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int u = 0; u < 8; u++)
#pragma unroll
                for (int c = 0; c < 4; c++) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[c], 0, 0, 0);
        }
    } else {
        struct { unsigned x[4]; } pa = {{w[0], w[1], w[2], w[3]}}, pb = {{w[1], w[2], w[3], w[0]}};
        const bf16x8 a = __builtin_bit_cast(bf16x8, pa), b = __builtin_bit_cast(bf16x8, pb);
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int u = 0; u < 8; u++)
#pragma unroll
                for (int c = 0; c < 4; c++) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[c], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int c = 0; c < 4; c++) s += acc[c][0] + acc[c][7];
    if (s == 123.456f) sink[0] = s;      // never true: keeps the chains alive
}

static unsigned rnd(unsigned& s) { s = s * 1664525u + 1013904223u; return s; }

int main() {
    unsigned* d_seed; float* d_sink;
    hipMalloc(&d_seed, 4096 * 4); hipMalloc(&d_sink, 4);
    unsigned h[4096];
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char* inames[3] = {"v_mfma_f32_32x32x16_f16", "v_mfma_f32_32x32x16_bf16", "v_mfma_f32_32x32x2_f32"};
    const double flop_per[3] = {2.0 * 32 * 32 * 16, 2.0 * 32 * 32 * 16, 2.0 * 32 * 32 * 2};
    const int passes[3] = {8, 8, 16};            // 4-cycle passes per instruction (MI355X_MICROARCH.md): 32 / 32 / 64 cycles
    const char* pnames[3] = {"zero", "unit_normal", "conv_like"};
    printf("{\"device\": \"MI355X\", \"waves_per_simd\": 2, \"cases\": [\n");
    bool first = true;
    for (int mode = 0; mode < 3; mode++) {
        for (int pat = 0; pat < 3; pat++) {
            unsigned s = 12345u + 77u * pat;
            for (int i = 0; i < 4096; i++) {
                if (pat == 0) { h[i] = 0u; continue; }
                // two 16-bit (or one 32-bit) values per word: pat 1 ~ N(0,1)-ish activations, pat 2: one operand large-ish
                // activations, the other small weights (|w| ~ 1e-2), as a convolution sees them
                float v0 = ((rnd(s) >> 8) / 8388608.0f - 1.0f) * (pat == 2 && (i & 1) ? 0.03f : 1.7f);
                float v1 = ((rnd(s) >> 8) / 8388608.0f - 1.0f) * (pat == 2 && (i & 1) ? 0.03f : 1.7f);
                if (mode == 2) h[i] = __builtin_bit_cast(unsigned, v0);
                else if (mode == 0) {
                    _Float16 a = (_Float16)v0, b = (_Float16)v1;
                    unsigned short ua, ub; memcpy(&ua, &a, 2); memcpy(&ub, &b, 2);
                    h[i] = (unsigned)ua | ((unsigned)ub << 16);
                } else {
                    h[i] = (__builtin_bit_cast(unsigned, v0) >> 16) | (__builtin_bit_cast(unsigned, v1) & 0xffff0000u);
                }
            }
            hipMemcpy(d_seed, h, sizeof(h), hipMemcpyHostToDevice);
            const int blocks = 256 * 2;                  // 2 workgroups of 4 waves per CU = 2 waves per SIMD
            int iters = mode == 2 ? 4000 : 8000;
            auto launch = [&](int it) {
                if (mode == 0) hipLaunchKernelGGL(spin<0>, dim3(blocks), dim3(256), 0, 0, d_seed, d_sink, it);
                else if (mode == 1) hipLaunchKernelGGL(spin<1>, dim3(blocks), dim3(256), 0, 0, d_seed, d_sink, it);
                else hipLaunchKernelGGL(spin<2>, dim3(blocks), dim3(256), 0, 0, d_seed, d_sink, it);
            };
            launch(iters / 8);
            hipDeviceSynchronize();
            // repeat until ~0.25 s have been spent in one launch sequence (power state settles within tens of ms)
            float ms = 0.f;
            int reps = 0;
            hipEventRecord(e0);
            while (reps < 200) {
                launch(iters);
                reps++;
                if (reps % 4 == 0) {
                    hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
                    if (ms > 250.f) break;
                }
            }
            hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
            const double n_mfma = (double)reps * iters * 32.0 * (blocks * 4.0);       // per wave: iters * 8 * 4
            const double tflops = n_mfma * flop_per[mode] / (ms * 1e-3) / 1e12;
            // every SIMD issues one MFMA at a time: cycles = MFMAs per SIMD x passes x 4
            const double mhz = (n_mfma / (256.0 * 4.0)) * passes[mode] * 4.0 / (ms * 1e-3) / 1e6;
            printf("%s  {\"instruction\": \"%s\", \"operands\": \"%s\", \"tflops\": %.1f, \"effective_mhz\": %.0f, \"ms\": %.1f}", first ? "" : ",\n",
                   inames[mode], pnames[pat], tflops, mhz, ms);
            first = false;
            fflush(stdout);
        }
    }
    printf("\n]}\n");
    return 0;
}
