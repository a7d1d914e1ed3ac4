// What do other instructions cost beside v_mfma_f32_32x32x2_f32?  A register-resident loop of 8 independent MFMAs with F
// filler instructions placed behind every MFMA (sched_group_barrier), for one, two and three waves per SIMD: cycles of the
// matrix pipe per MFMA (64 = the pipe is never idle).  Fillers: v_fma_f32 on private registers (VALU), ds_read_b64 from a
// private LDS slot (LDS), buffer-free global_load_dwordx4 of an L2-resident line (VMEM).  The Winograd kernels
// issue ~3 VALU + 0.5 LDS + 0.3 VMEM instructions per MFMA at two to three waves per SIMD.
// Build + run (GPU box): hipcc --offload-arch=gfx950 -O2 scripts/micro/mfma_fillers.hip -o /tmp/mf && /tmp/mf
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float floatx16 __attribute__((ext_vector_type(16)));

// KIND 0: VALU fillers, 1: LDS reads (b64), 2: global loads (dwordx4, L2-resident), 3: global loads (dword), 4: LDS reads (b128)
template <int F, int KIND>
__global__ void __launch_bounds__(256) spin(const float* __restrict__ src, float* __restrict__ sink, int iters) {
    __shared__ float lds[4096];
    const int tid = threadIdx.x;
    lds[tid & 4095] = src[tid & 1023];
    __syncthreads();
    floatx16 acc[8];
    for (int c = 0; c < 8; c++)
        for (int i = 0; i < 16; i++) acc[c][i] = 0.f;
    float a = src[tid & 1023], b = src[(tid + 7) & 1023];
    float f[8] = {a, b, a + 1.f, b + 1.f, a + 2.f, b + 2.f, a + 3.f, b + 3.f};
    const float* lp = lds + 2 * (tid & 1023);
    typedef float f4 __attribute__((ext_vector_type(4)));
    const f4* gp = reinterpret_cast<const f4*>(src) + (tid & 63);
    float4 g4 = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int c = 0; c < 8; c++) {
            acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
#pragma unroll
            for (int k = 0; k < F; k++) {
                if (KIND == 0) f[(c + k) & 7] = __builtin_fmaf(f[(c + k) & 7], 1.0001f, 0.5f);
                else if (KIND == 1) { typedef float f2 __attribute__((ext_vector_type(2))); const f2 v = *reinterpret_cast<const f2*>(lp + 8 * ((c + k + it) & 7)); f[(c + k) & 7] += v.x; }
                else if (KIND == 2) { const f4 v = gp[64 * ((c * F + k + it) & 15)]; g4.x += v.x; }
                else if (KIND == 3) { g4.x += src[(tid & 63) + 64 * ((c * F + k + it) & 63)]; }
                else { const f4 v = *reinterpret_cast<const f4*>(lds + 4 * (tid & 255) + 1024 * ((c + k + it) & 3)); f[(c + k) & 7] += v.x; }
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (F > 0) __builtin_amdgcn_sched_group_barrier(KIND == 0 ? 0x002 : ((KIND == 1 || KIND == 4) ? 0x100 : 0x020), F, 0);
            if (F > 0 && KIND != 0) __builtin_amdgcn_sched_group_barrier(0x002, F, 0);
        }
    }
    float s = g4.x;
    for (int c = 0; c < 8; c++) s += acc[c][0] + acc[c][9] + f[c];
    if (s == 123.456f) sink[0] = s;
}

template <int F, int KIND>
static void run(const float* src, float* sink, int waves_per_simd, double mhz) {
    const int threads = 256, blocks = 256 * waves_per_simd;     // 4 waves per block = one per SIMD
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((spin<F, KIND>), dim3(blocks), dim3(threads), 0, 0, src, sink, 200);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((spin<F, KIND>), dim3(blocks), dim3(threads), 0, 0, src, sink, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const double mfma_per_simd = (double)iters * 8 * waves_per_simd;
    const double cyc = ms * 1e-3 * mhz * 1e6 / mfma_per_simd;
    printf("  {\"kind\": \"%s\", \"fillers_per_mfma\": %d, \"waves_per_simd\": %d, \"ms\": %.3f, \"pipe_cycles_per_mfma_at_%.0f_MHz\": %.1f, \"tflops\": %.1f},\n",
           KIND == 0 ? "valu" : (KIND == 1 ? "lds_b64" : (KIND == 2 ? "vmem_dwordx4" : (KIND == 3 ? "vmem_dword" : "lds_b128"))), F, waves_per_simd, ms, mhz, cyc,
           mfma_per_simd * 1024.0 * 4096.0 / (ms * 1e-3) / 1e12);
}

int main() {
    float *src, *sink;
    hipMalloc(&src, 1 << 20); hipMalloc(&sink, 64);
    hipMemset(src, 0, 1 << 20);
    float h[1024];
    unsigned s = 12345u;
    for (int i = 0; i < 1024; i++) { s = s * 1664525u + 1013904223u; h[i] = (float)(s >> 8) / (1 << 24) - 0.5f; }
    hipMemcpy(src, h, sizeof(h), hipMemcpyHostToDevice);
    const double mhz = 2400.0;
    printf("{\"instruction\": \"v_mfma_f32_32x32x2_f32\", \"cases\": [\n");
    for (int w = 1; w <= 3; w++) {
        run<0, 0>(src, sink, w, mhz);
        run<2, 0>(src, sink, w, mhz); run<4, 0>(src, sink, w, mhz); run<8, 0>(src, sink, w, mhz); run<12, 0>(src, sink, w, mhz);
        run<1, 1>(src, sink, w, mhz); run<2, 1>(src, sink, w, mhz);
        run<1, 2>(src, sink, w, mhz); run<2, 2>(src, sink, w, mhz);
        run<1, 3>(src, sink, w, mhz); run<2, 3>(src, sink, w, mhz);
        run<1, 4>(src, sink, w, mhz); run<2, 4>(src, sink, w, mhz);
    }
    printf("  {}\n]}\n");
    return 0;
}
