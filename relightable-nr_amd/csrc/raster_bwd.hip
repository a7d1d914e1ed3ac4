// Rasterizer backward passes for gfx950: gradients of the rgb/alpha maps (silhouette sweep), of the depth map and of
// the per-face texture cubes.  Replaces rasterize_cuda_kernel.cu:244-592 (launchers 693-800) behind the C ABI.
//
// Built with -ffp-contract=off: the silhouette sweep reproduces the reference's binary32 operation order so that its
// result is bit-identical to the reference kernel (each gradient entry is accumulated by ONE thread in edge order).
#include "rnr_internal.h"

namespace rnr {

// float -> int with the GPU conversion rules the reference relies on (NaN -> 0, saturating), made explicit.
__device__ __forceinline__ int cvt_i32(float x) {
    if (!(x == x)) return 0;
    if (x >= 2147483520.f) return 2147483647;
    if (x <= -2147483648.f) return -2147483647 - 1;
    return (int)x;
}

struct PixelBwdParams {
    const float* faces;
    const int32_t* face_index_map;
    const float* rgb_map;
    const float* alpha_map;
    const float* grad_rgb_map;
    const float* grad_alpha_map;
    float* grad_faces;
    long total;   // B * nf
    int nf, is;
    float eps;
    int return_rgb, return_alpha;
};

// One thread per (face, axis).  Axis 0 sweeps columns (d0 = x, d1 = y) and owns the y-gradients of the three
// vertices (plus the always-zero z entries); axis 1 sweeps rows and owns the x-gradients.  The reference runs both
// axes in one thread per face (rasterize_cuda_kernel.cu:258-498); the two axes never touch the same entry, so splitting
// them keeps every entry's accumulation order (edge 0, 1, 2; d0 ascending; "out" run then "in" run) and doubles the
// parallelism.  Consecutive threads share a face, so its 9 floats are fetched once per pair.
__global__ void __launch_bounds__(256) backward_pixel_map_kernel(PixelBwdParams P) {
    const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= 2 * P.total) return;
    const long i = tid >> 1;
    const int axis = (int)(tid & 1);
    const int bn = (int)(i / P.nf);
    const int fn = (int)(i % P.nf);
    const int is = P.is;
    float face[9];
#pragma unroll
    for (int k = 0; k < 9; k++) face[k] = P.faces[i * 9 + k];
    if ((face[7] - face[1]) * (face[3] - face[0]) < (face[4] - face[1]) * (face[6] - face[0])) return;   // back side: untouched

    const float fis = (float)is;
    float grad[3] = {0.f, 0.f, 0.f};        // entries [vertex * 3 + (1 - axis)]
    const long map_base = (long)bn * is * is;
    const int map_offset = axis == 0 ? is : 1;      // step along d1
    const int d0_stride = axis == 0 ? 1 : is;       // step along d0

#pragma unroll
    for (int edge = 0; edge < 3; edge++) {
        int pi[3];
        float p[3][2];
#pragma unroll
        for (int num = 0; num < 3; num++) {
            pi[num] = (edge + num) % 3;
#pragma unroll
            for (int dim = 0; dim < 2; dim++) {
                const float f = face[3 * pi[num] + ((dim + axis) & 1)];
                p[num][dim] = 0.5f * (f * fis + fis - 1.f);
            }
        }
        const bool lt = p[0][0] < p[1][0];
        const int direction = (axis == 0) ? (lt ? -1 : 1) : (lt ? 1 : -1);
        // d0 range of the edge, clamped to the image (the float clamp keeps the int conversion in range)
        const float lo = fmaxf(ceilf(fminf(p[0][0], p[1][0])), 0.f);
        const float hi = fminf(fmaxf(p[0][0], p[1][0]), fis - 1.f);
        // int conversion truncates toward zero, so an edge wholly inside (-1, 0) still visits d0 = 0, as in the reference
        const int d0_from = cvt_i32(lo);
        const int d0_to = cvt_i32(hi);
        const float slope = (p[1][1] - p[0][1]) / (p[1][0] - p[0][0]);
        const float run = p[1][0] - p[0][0];
        for (int d0 = d0_from; d0 <= d0_to; d0++) {
            const float fd0 = (float)d0;
            const float d1_cross = slope * (fd0 - p[0][0]) + p[0][1];
            const int d1_in = cvt_i32(0 < direction ? floorf(d1_cross) : ceilf(d1_cross));
            if (d1_in < 0 || is <= d1_in) continue;
            const int d1_out = d1_in + direction;
            if (d1_out < 0 || is <= d1_out) continue;
            const long line = map_base + (long)d0 * d0_stride;
            const long idx_in = line + (long)d1_in * map_offset;
            const long idx_out = line + (long)d1_out * map_offset;
            float alpha_in = 0.f, alpha_out = 0.f, rgb_in[3] = {0.f, 0.f, 0.f}, rgb_out[3] = {0.f, 0.f, 0.f};
            if (P.return_alpha) { alpha_in = P.alpha_map[idx_in]; alpha_out = P.alpha_map[idx_out]; }
            if (P.return_rgb) {
#pragma unroll
                for (int k = 0; k < 3; k++) { rgb_in[k] = P.rgb_map[idx_in * 3 + k]; rgb_out[k] = P.rgb_map[idx_out * 3 + k]; }
            }
            const bool upd0 = p[1][0] != fd0, upd1 = p[0][0] != fd0;
            const float lever0 = run / (p[1][0] - fd0), lever1 = run / (fd0 - p[0][0]);

            auto accumulate = [&](int d1, float diff_grad) {
                const float off = (float)d1 - d1_cross;
                if (upd0) {
                    float dist = (lever0 * off) * 2.f / fis;
                    dist = (0 < dist) ? dist + P.eps : dist - P.eps;
                    grad[pi[0]] -= diff_grad / dist;
                }
                if (upd1) {
                    float dist = (lever1 * off) * 2.f / fis;
                    dist = (0 < dist) ? dist + P.eps : dist - P.eps;
                    grad[pi[1]] -= diff_grad / dist;
                }
            };

            // pixels outside the face along the sweep direction, only when the inner pixel shows this face
            if (P.face_index_map[idx_in] == fn) {
                const int d1_limit = 0 < direction ? is - 1 : 0;
                const int d1_from = max(min(d1_out, d1_limit), 0);
                const int d1_to = min(max(d1_out, d1_limit), is - 1);
                for (int d1 = d1_from; d1 <= d1_to; d1++) {
                    const long m = line + (long)d1 * map_offset;
                    float diff_grad = 0.f;
                    if (P.return_alpha) diff_grad += (P.alpha_map[m] - alpha_in) * P.grad_alpha_map[m];
                    if (P.return_rgb) {
#pragma unroll
                        for (int k = 0; k < 3; k++) diff_grad += (P.rgb_map[m * 3 + k] - rgb_in[k]) * P.grad_rgb_map[m * 3 + k];
                    }
                    if (diff_grad <= 0) continue;
                    accumulate(d1, diff_grad);
                }
            }
            // pixels inside the face, up to the opposite edge
            {
                float d0_cross2;
                if ((fd0 - p[0][0]) * (fd0 - p[2][0]) < 0)
                    d0_cross2 = (p[2][1] - p[0][1]) / (p[2][0] - p[0][0]) * (fd0 - p[0][0]) + p[0][1];
                else
                    d0_cross2 = (p[1][1] - p[2][1]) / (p[1][0] - p[2][0]) * (fd0 - p[2][0]) + p[2][1];
                const int d1_limit = cvt_i32(0 < direction ? ceilf(d0_cross2) : floorf(d0_cross2));
                const int d1_from = max(min(d1_in, d1_limit), 0);
                const int d1_to = min(max(d1_in, d1_limit), is - 1);
                for (int d1 = d1_from; d1 <= d1_to; d1++) {
                    const long m = line + (long)d1 * map_offset;
                    if (P.face_index_map[m] != fn) continue;
                    float diff_grad = 0.f;
                    if (P.return_alpha) diff_grad += (P.alpha_map[m] - alpha_out) * P.grad_alpha_map[m];
                    if (P.return_rgb) {
#pragma unroll
                        for (int k = 0; k < 3; k++) diff_grad += (P.rgb_map[m * 3 + k] - rgb_out[k]) * P.grad_rgb_map[m * 3 + k];
                    }
                    if (diff_grad <= 0) continue;
                    accumulate(d1, diff_grad);
                }
            }
        }
    }
    float* g = P.grad_faces + i * 9;
#pragma unroll
    for (int v = 0; v < 3; v++) {
        g[v * 3 + (1 - axis)] = grad[v];
        if (axis == 0) g[v * 3 + 2] = 0.f;
    }
}

// Sum over the lanes of `mask` (a set of lanes of this wave that all execute this call); every lane of the set
// receives nothing useful except the lowest one, which gets the total.
__device__ __forceinline__ float masked_wave_sum(float v, unsigned long long mask, bool member) {
    float x = member ? v : 0.f;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) x += __shfl_xor(x, off, 64);
    (void)mask;
    return x;
}

// backward_textures (rasterize_cuda_kernel.cu:500-535): grad_textures[b, face, texel, :] += w * grad_rgb, 8 texels a
// pixel.  Float atomics on HBM (global_atomic_add_f32); the summation order is unspecified in the reference as well.
__global__ void __launch_bounds__(256) backward_textures_kernel(const int32_t* __restrict__ face_index_map,
                                                                const float* __restrict__ sampling_weight_map,
                                                                const int32_t* __restrict__ sampling_index_map,
                                                                const float* __restrict__ grad_rgb_map,
                                                                float* grad_textures, long npix, int nf, int is, int ts) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix) return;
    const int face_index = face_index_map[i];
    if (face_index < 0) return;
    const long bn = i / ((long)is * is);
    float* grad_texture = grad_textures + (bn * nf + face_index) * (long)ts * ts * ts * 3;
    const float g0 = grad_rgb_map[i * 3], g1 = grad_rgb_map[i * 3 + 1], g2 = grad_rgb_map[i * 3 + 2];
    const float4* wp = reinterpret_cast<const float4*>(sampling_weight_map + i * 8);
    const int4* ip = reinterpret_cast<const int4*>(sampling_index_map + i * 8);
    const float4 w0 = wp[0], w1 = wp[1];
    const int4 i0 = ip[0], i1 = ip[1];
    const float w[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
    const int isc[8] = {i0.x, i0.y, i0.z, i0.w, i1.x, i1.y, i1.z, i1.w};
#pragma unroll
    for (int pn = 0; pn < 8; pn++) {
        float* t = grad_texture + (long)isc[pn] * 3;
        unsafeAtomicAdd(t, w[pn] * g0);
        unsafeAtomicAdd(t + 1, w[pn] * g1);
        unsafeAtomicAdd(t + 2, w[pn] * g2);
    }
}

// backward_depth_map (rasterize_cuda_kernel.cu:537-592): 9 atomics a pixel in the reference.  A wave covers 64
// consecutive pixels of a row, which mostly show a handful of faces: lanes that share a face are summed with wave
// shuffles first and one lane issues the 9 atomics (order unspecified in the reference too).
__global__ void __launch_bounds__(256) backward_depth_map_kernel(const float* __restrict__ faces,
                                                                 const float* __restrict__ depth_map,
                                                                 const int32_t* __restrict__ face_index_map,
                                                                 const float* __restrict__ face_inv_map,
                                                                 const float* __restrict__ weight_map,
                                                                 const float* __restrict__ grad_depth_map,
                                                                 float* grad_faces, long npix, int nf, int is) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    int fn = -1;
    long bn = 0;
    float g[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (i < npix) fn = face_index_map[i];
    if (fn >= 0) {
        bn = i / ((long)is * is);
        const float* face = faces + (bn * nf + fn) * 9;
        const float depth = depth_map[i];
        const float depth2 = depth * depth;
        const float* face_inv = face_inv_map + i * 9;
        const float* weight = weight_map + i * 3;
        const float grad_depth = grad_depth_map[i];
        float z[3], wgt[3];
#pragma unroll
        for (int k = 0; k < 3; k++) { z[k] = face[3 * k + 2]; wgt[k] = weight[k]; }
#pragma unroll
        for (int k = 0; k < 3; k++) g[3 * k + 2] = grad_depth * wgt[k] * depth2 / (z[k] * z[k]);
        float tmp[2] = {0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 2; k++)
#pragma unroll
            for (int l = 0; l < 3; l++) tmp[k] += -face_inv[3 * l + k] / z[l];
#pragma unroll
        for (int k = 0; k < 3; k++)
#pragma unroll
            for (int l = 0; l < 2; l++) g[3 * k + l] = -grad_depth * tmp[l] * wgt[k] * depth2 * (float)is / 2.f;
    }
    // segmented wave reduction keyed by (batch, face)
    const long key = fn >= 0 ? bn * nf + fn : -1;
    unsigned long long todo = __ballot(key >= 0);
    const int lane = threadIdx.x & 63;
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const long lkey = __shfl(key, leader, 64);
        const bool member = key == lkey;
        const unsigned long long grp = __ballot(member);
        float s[9];
#pragma unroll
        for (int k = 0; k < 9; k++) s[k] = masked_wave_sum(g[k], grp, member);
        if (lane == leader) {
            float* gf = grad_faces + lkey * 9;
#pragma unroll
            for (int k = 0; k < 9; k++) unsafeAtomicAdd(gf + k, s[k]);
        }
        todo &= ~grp;
    }
}

}  // namespace rnr

using namespace rnr;

extern "C" int rnr_backward_pixel_map(const float* faces, const int32_t* face_index_map, const float* rgb_map,
                                      const float* alpha_map, const float* grad_rgb_map, const float* grad_alpha_map,
                                      float* grad_faces, int batch_size, int num_faces, int image_size, float eps,
                                      int return_rgb, int return_alpha, void* stream) {
    RNR_REQUIRE(faces && face_index_map && grad_faces, "rnr_backward_pixel_map: null pointer argument");
    RNR_REQUIRE(!return_rgb || (rgb_map && grad_rgb_map), "rnr_backward_pixel_map: return_rgb needs rgb_map and grad_rgb_map");
    RNR_REQUIRE(!return_alpha || (alpha_map && grad_alpha_map),
                "rnr_backward_pixel_map: return_alpha needs alpha_map and grad_alpha_map");
    RNR_REQUIRE(batch_size > 0 && num_faces > 0 && image_size > 0 && image_size <= 16384,
                "rnr_backward_pixel_map: bad sizes B=%d nf=%d is=%d", batch_size, num_faces, image_size);
    if (!return_rgb && !return_alpha) return 0;       // rasterize.py:203-204
    PixelBwdParams P;
    P.faces = faces; P.face_index_map = face_index_map; P.rgb_map = rgb_map; P.alpha_map = alpha_map;
    P.grad_rgb_map = grad_rgb_map; P.grad_alpha_map = grad_alpha_map; P.grad_faces = grad_faces;
    P.total = (long)batch_size * num_faces; P.nf = num_faces; P.is = image_size; P.eps = eps;
    P.return_rgb = return_rgb; P.return_alpha = return_alpha;
    const long threads = 2 * P.total;
    hipLaunchKernelGGL(backward_pixel_map_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, as_stream(stream), P);
    return check_launch("backward_pixel_map_kernel");
}

extern "C" int rnr_backward_textures(const int32_t* face_index_map, const float* sampling_weight_map,
                                     const int32_t* sampling_index_map, const float* grad_rgb_map, float* grad_textures,
                                     int batch_size, int num_faces, int image_size, int texture_size, void* stream) {
    RNR_REQUIRE(face_index_map && sampling_weight_map && sampling_index_map && grad_rgb_map && grad_textures,
                "rnr_backward_textures: null pointer argument");
    RNR_REQUIRE(batch_size > 0 && num_faces > 0 && image_size > 0 && texture_size > 0, "rnr_backward_textures: bad sizes");
    const long npix = (long)batch_size * image_size * image_size;
    hipLaunchKernelGGL(backward_textures_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, as_stream(stream),
                       face_index_map, sampling_weight_map, sampling_index_map, grad_rgb_map, grad_textures, npix,
                       num_faces, image_size, texture_size);
    return check_launch("backward_textures_kernel");
}

extern "C" int rnr_backward_depth_map(const float* faces, const float* depth_map, const int32_t* face_index_map,
                                      const float* face_inv_map, const float* weight_map, const float* grad_depth_map,
                                      float* grad_faces, int batch_size, int num_faces, int image_size, void* stream) {
    RNR_REQUIRE(faces && depth_map && face_index_map && face_inv_map && weight_map && grad_depth_map && grad_faces,
                "rnr_backward_depth_map: null pointer argument");
    RNR_REQUIRE(batch_size > 0 && num_faces > 0 && image_size > 0, "rnr_backward_depth_map: bad sizes");
    const long npix = (long)batch_size * image_size * image_size;
    hipLaunchKernelGGL(backward_depth_map_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, as_stream(stream),
                       faces, depth_map, face_index_map, face_inv_map, weight_map, grad_depth_map, grad_faces, npix,
                       num_faces, image_size);
    return check_launch("backward_depth_map_kernel");
}
