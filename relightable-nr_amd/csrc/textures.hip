// Per-face texture cubes <-> texture images for gfx950: the two remaining extension modules of neural_renderer
// (load_textures_cuda_kernel.cu:1-152, create_texture_image_cuda_kernel.cu:1-163).  Off the per-view hot path
// (network.py:108 loads meshes without textures); provided so that the package boundary is complete.
// Built with -ffp-contract=off: one rounding per source operation, as the reference kernels evaluate in binary32.
#include "rnr_internal.h"

namespace rnr {

// mod() of load_textures_cuda_kernel.cu:7-15
__device__ __forceinline__ float wrap_mod(float x, float y) { return x > 0 ? fmodf(x, y) : y + fmodf(x, y); }

// The reference wraps face[0..5] in place from EVERY texel thread of the face (cu:54-77): a data race whose outcome
// depends on thread timing when a coordinate is an exact integer (mod(1) -> 0 -> 1 -> ...).  Here the wrap is applied
// exactly once per coordinate, in its own pass; for non-integer coordinates the result is the reference's.
__global__ void __launch_bounds__(256) wrap_face_uv_kernel(float* faces, const int32_t* __restrict__ is_update, int total,
                                                           int wrapping) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    if (is_update[i / 6] == 0) return;
    float f = faces[i];
    if (wrapping == 0) f = wrap_mod(f, 1.f);                                                      // REPEAT
    else if (wrapping == 1) f = wrap_mod(f, 2.f) < 1 ? wrap_mod(f, 1.f) : 1 - wrap_mod(f, 1.f);   // MIRRORED_REPEAT
    else if (wrapping == 2) f = fmaxf(fminf(f, 1.f), 0.f);                                        // CLAMP_TO_EDGE
    faces[i] = f;
}

// One lane per texel of a face cube (cu:25-120): barycentric position of the texel -> uv -> bilinear / nearest fetch.
__global__ void __launch_bounds__(256) load_textures_kernel(const float* __restrict__ image,
                                                            const int32_t* __restrict__ is_update,
                                                            const float* __restrict__ faces, float* __restrict__ textures,
                                                            long ntexels, int ts, int ih, int iw, int wrapping,
                                                            int use_bilinear) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ntexels) return;
    const long cube = (long)ts * ts * ts;
    const int fn = (int)(i / cube);
    if (is_update[fn] == 0) return;
    const float den = (float)(ts - 1);
    float dim0 = (float)((i / (ts * ts)) % ts) / den;
    float dim1 = (float)((i / ts) % ts) / den;
    float dim2 = (float)(i % ts) / den;
    if (0 < dim0 + dim1 + dim2) {
        const float sum = dim0 + dim1 + dim2;
        dim0 /= sum; dim1 /= sum; dim2 /= sum;
    }
    const float* face = faces + (long)fn * 6;
    float* texel = textures + i * 3;
    if (wrapping == 3) {                               // CLAMP_TO_BORDER: the reference writes zeros (cu:97-99, 111-113)
        texel[0] = 0.f; texel[1] = 0.f; texel[2] = 0.f;
        return;
    }
    const float pos_x = (face[0] * dim0 + face[2] * dim1 + face[4] * dim2) * (float)(iw - 1);
    const float pos_y = (face[1] * dim0 + face[3] * dim1 + face[5] * dim2) * (float)(ih - 1);
    if (use_bilinear) {
        const int xi = (int)pos_x, yi = (int)pos_y;
        const float wx1 = pos_x - (float)xi, wx0 = 1 - wx1;
        const float wy1 = pos_y - (float)yi, wy0 = 1 - wy1;
        const int y1 = min((int)(pos_y + 1), ih - 1), x1 = min(xi + 1, iw - 1);
#pragma unroll
        for (int k = 0; k < 3; k++) {
            float c = 0;
            c += image[((long)yi * iw + xi) * 3 + k] * (wx0 * wy0);
            c += image[((long)y1 * iw + xi) * 3 + k] * (wx0 * wy1);
            c += image[((long)yi * iw + x1) * 3 + k] * (wx1 * wy0);
            c += image[((long)y1 * iw + x1) * 3 + k] * (wx1 * wy1);
            texel[k] = c;
        }
    } else {
        const int xi = (int)roundf(pos_x), yi = (int)roundf(pos_y);
#pragma unroll
        for (int k = 0; k < 3; k++) texel[k] = image[((long)yi * iw + xi) * 3 + k];
    }
}

// One lane per image pixel (cu:9-100): the tile of face fn holds a right triangle; barycentrics of the pixel in that
// triangle index the face's cube trilinearly.  Tiles beyond the last face are left untouched (the reference reads
// past the end of `textures` there).
__global__ void __launch_bounds__(256) create_texture_image_kernel(const float* __restrict__ vertices_all,
                                                                   const float* __restrict__ textures,
                                                                   float* __restrict__ image, long npix, int nf, int tsi,
                                                                   int tso, int tile_width, float eps) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix) return;
    const int x = (int)(i % ((long)tile_width * tso));
    const int y = (int)(i / ((long)tile_width * tso));
    const int fn = x / tso + (y / tso) * tile_width;
    if (fn >= nf) return;
    const float* texture = textures + (long)fn * tsi * tsi * tsi * 3;
    const float* p0 = vertices_all + (long)fn * 6;
    const float* p1 = p0 + 2;
    const float* p2 = p0 + 4;
    float face_inv[9] = {p1[1] - p2[1], p2[0] - p1[0], p1[0] * p2[1] - p2[0] * p1[1],
                         p2[1] - p0[1], p0[0] - p2[0], p2[0] * p0[1] - p0[0] * p2[1],
                         p0[1] - p1[1], p1[0] - p0[0], p0[0] * p1[1] - p1[0] * p0[1]};
    const float den = p2[0] * (p0[1] - p1[1]) + p0[0] * (p1[1] - p2[1]) + p1[0] * (p2[1] - p0[1]);
#pragma unroll
    for (int k = 0; k < 9; k++) face_inv[k] /= den;
    float weight[3], weight_sum = 0;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        weight[k] = face_inv[3 * k + 0] * (float)x + face_inv[3 * k + 1] * (float)y + face_inv[3 * k + 2];
        weight_sum += weight[k];
    }
    float tif[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        weight[k] /= (weight_sum + eps);
        float t = weight[k] * (float)(tsi - 1);
        t = fmaxf(t, 0.f);
        t = fminf(t, (float)(tsi - 1) - eps);
        tif[k] = t;
    }
    float px[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int pn = 0; pn < 8; pn++) {
        float w = 1;
        int ti[3];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const int fl = (int)tif[k];
            if (((pn >> k) & 1) == 0) { w *= 1 - (tif[k] - (float)fl); ti[k] = fl; }
            else { w *= tif[k] - (float)fl; ti[k] = fl + 1; }
        }
        const int isc = ti[0] * tsi * tsi + ti[1] * tsi + ti[2];
#pragma unroll
        for (int k = 0; k < 3; k++) px[k] += w * texture[isc * 3 + k];
    }
#pragma unroll
    for (int k = 0; k < 3; k++) image[i * 3 + k] = px[k];
}

// Second pass (cu:104-122): the first pixel right of each tile's diagonal copies its left neighbour.
__global__ void __launch_bounds__(256) texture_image_boundary_kernel(float* image, long npix, int tso, int tile_width) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix) return;
    const long width = (long)tile_width * tso;
    const int x = (int)(i % width), y = (int)(i / width);
    if ((y % tso + 1) == (x % tso)) {
#pragma unroll
        for (int k = 0; k < 3; k++) image[i * 3 + k] = image[(y * width + (x - 1)) * 3 + k];
    }
}

}  // namespace rnr

using namespace rnr;

extern "C" int rnr_load_textures(const float* image, float* faces, float* textures, const int32_t* is_update,
                                 int num_faces, int texture_size, int image_height, int image_width,
                                 int texture_wrapping, int use_bilinear, void* stream) {
    RNR_REQUIRE(image && faces && textures && is_update, "rnr_load_textures: null pointer argument");
    RNR_REQUIRE(num_faces > 0 && texture_size > 1 && image_height > 0 && image_width > 0,
                "rnr_load_textures: bad sizes nf=%d ts=%d image %dx%d", num_faces, texture_size, image_height, image_width);
    RNR_REQUIRE(texture_wrapping >= 0 && texture_wrapping <= 3, "rnr_load_textures: unknown wrapping mode %d", texture_wrapping);
    hipStream_t st = as_stream(stream);
    const int ncoord = num_faces * 6;
    if (texture_wrapping != 3) {
        hipLaunchKernelGGL(wrap_face_uv_kernel, dim3((ncoord + 255) / 256), dim3(256), 0, st, faces, is_update, ncoord,
                           texture_wrapping);
        if (int e = check_launch("wrap_face_uv_kernel")) return e;
    }
    const long ntexels = (long)num_faces * texture_size * texture_size * texture_size;
    hipLaunchKernelGGL(load_textures_kernel, dim3((unsigned)((ntexels + 255) / 256)), dim3(256), 0, st, image, is_update,
                       faces, textures, ntexels, texture_size, image_height, image_width, texture_wrapping, use_bilinear);
    return check_launch("load_textures_kernel");
}

extern "C" int rnr_create_texture_image(const float* vertices_all, const float* textures, float* image, int num_faces,
                                        int texture_size_in, int image_height, int image_width, float eps, void* stream) {
    RNR_REQUIRE(vertices_all && textures && image, "rnr_create_texture_image: null pointer argument");
    RNR_REQUIRE(num_faces > 0 && texture_size_in > 1 && image_height > 0 && image_width > 0,
                "rnr_create_texture_image: bad sizes");
    int tile_width = 1;                                   // int(sqrt(nf - 1)) + 1   (create_texture_image_cuda_kernel.cu:133)
    while ((long)tile_width * tile_width <= (long)num_faces - 1) tile_width++;
    RNR_REQUIRE(image_width % tile_width == 0, "rnr_create_texture_image: image width %d is not a multiple of the tile count %d",
                image_width, tile_width);
    const int tso = image_width / tile_width;
    hipStream_t st = as_stream(stream);
    const long npix = (long)image_height * image_width;
    hipLaunchKernelGGL(create_texture_image_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, st, vertices_all,
                       textures, image, npix, num_faces, texture_size_in, tso, tile_width, eps);
    if (int e = check_launch("create_texture_image_kernel")) return e;
    hipLaunchKernelGGL(texture_image_boundary_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, st, image, npix,
                       tso, tile_width);
    return check_launch("texture_image_boundary_kernel");
}
