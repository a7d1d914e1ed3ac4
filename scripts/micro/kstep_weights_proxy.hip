// K-step proxy for conv_wino_kernel: do the weights cost less through LDS?  Both kernels run the
// instruction mix of one Winograd K step per wave — 2 LDS patch reads (ds_read2_b64), 12 transform VALU, 8 independent
// v_mfma_f32_32x32x2_f32 whose A operands come from the transform and whose B operands are the freshly fetched weights — at two
// waves per SIMD, and differ only in how a wave gets its 2 KB of weights per K step:
//   G  (shipped): two buffer_load_dwordx4 per wave and K step straight into the operand registers, five K steps ahead (ring of 8);
//   L  (proposed): 8-wave workgroups = two pixel-tile groups x four plane rows sharing the weights: a wave fetches ONE 1 KB piece
//      per K step by LDS-DMA (global_load_lds_dwordx4) into a 4-deep LDS ring, the two groups' waves of a plane row read the same
//      2 KB block with two ds_read_b128 each; one workgroup barrier per K step publishes the pieces.
// The weight image (256 KB) is L2-resident.  Output: matrix-pipe cycles per MFMA (64 = the pipe never idles).
// Build + run (GPU box): hipcc --offload-arch=gfx950 -O3 scripts/micro/kstep_weights_proxy.hip -o /tmp/kp && /tmp/kp
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

constexpr int STEPS = 64;               // K steps per "tile" (the weight image holds STEPS x 8 KB = 512 KB; reused every tile)
constexpr int PATCH = 8 * 2 * 20 * 24;  // a halo-like LDS image for the patch reads

__device__ __forceinline__ void transform(const float (&d)[8], float sigma, float (&v)[4]) {
    float t[4];
#pragma unroll
    for (int j = 0; j < 4; j++) t[j] = __builtin_fmaf(sigma, d[4 + j], d[j]);
    v[0] = t[0] - t[2]; v[1] = t[1] + t[2]; v[2] = t[2] - t[1]; v[3] = t[1] - t[3];
}

__global__ void __launch_bounds__(256, 2) kstep_g(const float* __restrict__ w, float* __restrict__ sink, int tiles) {
    __shared__ __attribute__((aligned(16))) float lds[PATCH];
    const int tid = threadIdx.x, lane = tid & 63, xi = tid >> 6;
    for (int i = tid; i < PATCH; i += 256) lds[i] = w[i & 4095];
    __syncthreads();
    floatx16 acc[8];
    for (int p = 0; p < 8; p++) for (int g = 0; g < 16; g++) acc[p][g] = 0.f;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(w), 0, 0x7fffffff, 0x27000);
    const unsigned voff = (unsigned)lane * 16u;
    const float* pa = lds + (lane & 31) * 6 + (lane >> 5) * 480;
    const float sigma = xi == 1 ? 1.f : -1.f;
    for (int t = 0; t < tiles; t++) {
        unsigned soff = (unsigned)xi * 2048u;
        floatx4 b[8][2];
#pragma unroll
        for (int k = 0; k < 5; k++) {
            b[k][0] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)voff, (int)soff, 0));
            b[k][1] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(voff + 1024u), (int)soff, 0));
            soff += 8192u;
        }
#pragma unroll 8
        for (int s = 0; s < STEPS; s++) {
            const int r = (s + 5) & 7;
            b[r][0] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)voff, (int)(soff & 0x7ffffu), 0));
            b[r][1] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(voff + 1024u), (int)(soff & 0x7ffffu), 0));
            soff += 8192u;
            float d[8], v[4];
            const float* q = pa + (s & 7) * 960;
#pragma unroll
            for (int j = 0; j < 4; j++) { d[j] = q[j]; d[4 + j] = q[48 + j]; }
            transform(d, sigma, v);
#pragma unroll
            for (int p = 0; p < 8; p++)
                acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[p >> 1], b[s & 7][p & 1][p >> 1], acc[p], 0, 0, 0);
#pragma unroll
            for (int p = 0; p < 8; p++) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (p < 2) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
                if (p >= 4 && p < 6) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
        }
    }
    float sum = 0.f;
    for (int p = 0; p < 8; p++) sum += acc[p][0] + acc[p][7];
    if (sum == 123.456f) sink[0] = sum;
}

// RING K-step blocks of 8 KB in LDS; wave w fetches piece w of a block (8 pieces of 1 KB)
constexpr int RING = 4;
__global__ void __launch_bounds__(512, 1) kstep_l(const float* __restrict__ w, float* __restrict__ sink, int tiles) {
    __shared__ __attribute__((aligned(16))) float lds[PATCH];
    __shared__ __attribute__((aligned(16))) float wl[RING * 2048];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, xi = wave & 3;
    for (int i = tid; i < PATCH; i += 512) lds[i] = w[i & 4095];
    __syncthreads();
    floatx16 acc[8];
    for (int p = 0; p < 8; p++) for (int g = 0; g < 16; g++) acc[p][g] = 0.f;
    const float* pa = lds + (lane & 31) * 6 + (lane >> 5) * 480;
    const float sigma = xi == 1 ? 1.f : -1.f;
    const float* gsrc = w + wave * 256 + lane * 4;              // this wave's 1 KB piece of a step block
    auto dma = [&](int step) {
        __builtin_amdgcn_global_load_lds(gsrc + ((step * 2048) & 0x1ffff), wl + (step & (RING - 1)) * 2048 + wave * 256, 16, 0, 0);
    };
    for (int t = 0; t < tiles; t++) {
        for (int k = 0; k < RING - 1; k++) dma(k);
        for (int s = 0; s < STEPS; s++) {
            dma(s + RING - 1);
            // pieces of step s have landed (all but the RING - 1 newest DMA of this wave) and are visible to the workgroup
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(RING - 1) : "memory");
            __builtin_amdgcn_s_barrier();
            const floatx4* wb = reinterpret_cast<const floatx4*>(wl + (s & (RING - 1)) * 2048 + xi * 512);
            const floatx4 b0 = wb[lane], b1 = wb[64 + lane];
            float d[8], v[4];
            const float* q = pa + (s & 7) * 960;
#pragma unroll
            for (int j = 0; j < 4; j++) { d[j] = q[j]; d[4 + j] = q[48 + j]; }
            transform(d, sigma, v);
#pragma unroll
            for (int p = 0; p < 8; p++)
                acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[p >> 1], (p & 1) ? b1[p >> 1] : b0[p >> 1], acc[p], 0, 0, 0);
        }
        __syncthreads();
    }
    float sum = 0.f;
    for (int p = 0; p < 8; p++) sum += acc[p][0] + acc[p][7];
    if (sum == 123.456f) sink[0] = sum;
}

int main() {
    float *w, *sink;
    hipMalloc(&w, 1 << 20); hipMalloc(&sink, 64);
    float* h = (float*)malloc(1 << 20);
    unsigned s = 12345u;
    for (int i = 0; i < (1 << 18); i++) { s = s * 1664525u + 1013904223u; h[i] = (float)(s >> 8) / (1 << 24) - 0.5f; }
    hipMemcpy(w, h, 1 << 20, hipMemcpyHostToDevice);
    const int tiles = 200;
    const double mhz = 2400.0;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    printf("{\"what\": \"conv_wino K-step proxy: weights by global loads into registers (G) vs once per workgroup through LDS (L)\", \"cases\": [\n");
    for (int variant = 0; variant < 2; variant++) {
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0);
            if (variant == 0) hipLaunchKernelGGL(kstep_g, dim3(512), dim3(256), 0, 0, w, sink, rep ? tiles : 4);
            else hipLaunchKernelGGL(kstep_l, dim3(256), dim3(512), 0, 0, w, sink, rep ? tiles : 4);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            if (!rep) continue;
            float ms = 0.f;
            hipEventElapsedTime(&ms, e0, e1);
            const double mfma_per_simd = (double)tiles * STEPS * 8 * 2;     // two waves per SIMD in both variants
            printf("  {\"variant\": \"%s\", \"ms\": %.3f, \"pipe_cycles_per_mfma_at_%.0f_MHz\": %.1f, \"fraction_of_pipe\": %.3f}%s\n",
                   variant == 0 ? "G: 2 buffer_load_dwordx4 per wave and K step" : "L: 1 LDS-DMA piece + 2 ds_read_b128 per wave and K step, barrier per step",
                   ms, mhz, ms * 1e-3 * mhz * 1e6 / mfma_per_simd, 64.0 / (ms * 1e-3 * mhz * 1e6 / mfma_per_simd), variant == 0 ? "," : "");
        }
    }
    printf("]}\n");
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { fprintf(stderr, "HIP error: %s\n", hipGetErrorString(e)); return 1; }
    return 0;
}
