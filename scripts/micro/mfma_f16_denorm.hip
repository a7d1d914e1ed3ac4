// Does v_mfma_f32_32x32x16_f16 honour fp16 subnormal inputs on gfx950?  (decides whether the f16x3 emulation may leave
// residual terms in the subnormal range).  Build: hipcc --offload-arch=gfx950 -O2 mfma_f16_denorm.hip -o /tmp/mfd && /tmp/mfd
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 halfx8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
__global__ void k(float a_val, float b_val, float* out) {
    halfx8 a, b;
    for (int i = 0; i < 8; i++) { a[i] = (_Float16)a_val; b[i] = (_Float16)b_val; }
    floatx16 acc;
    for (int i = 0; i < 16; i++) acc[i] = 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    if (threadIdx.x == 0) out[0] = acc[0];
}
int main() {
    float* d; hipMalloc(&d, 4);
    const float cases[][2] = {{9.5367431640625e-07f, 1.0f}, {1.0f, 9.5367431640625e-07f}, {5.9604644775390625e-08f, 1.0f},
                              {3.0517578125e-05f, 3.0517578125e-05f}, {6.103515625e-05f, 1.0f}, {65504.f, 65504.f}};
    for (auto& c : cases) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, c[0], c[1], d);
        float h; hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
        printf("a=%.6e b=%.6e  acc=%.9e  expected(16*a*b)=%.9e  %s\n", c[0], c[1], h, 16.0 * c[0] * c[1],
               h == (float)(16.0 * c[0] * c[1]) ? "EXACT" : (h == 0.f ? "FLUSHED" : "DIFFERENT"));
    }
    return 0;
}
