"""Runs the reference's load_textures / create_texture_image KERNEL BODIES on the CPU (build container only).

Same technique as raster_kernel_harness.py: the __global__ templates of
  /root/reference/neural_renderer/neural_renderer/cuda/load_textures_cuda_kernel.cu        (lines 6-115)
  /root/reference/neural_renderer/neural_renderer/cuda/create_texture_image_cuda_kernel.cu (lines 8-116)
are copied into a TEMP directory behind a thread-index shim and driven by a serial loop over thread ids.  Nothing of
the reference is written into the repository; make_golden.py commits only the produced (input, output) vectors.
"""
import ctypes
import os
import subprocess
import tempfile

import numpy as np

D = '/root/reference/neural_renderer/neural_renderer/cuda/'
PARTS = [(D + 'load_textures_cuda_kernel.cu', 6, 115), (D + 'create_texture_image_cuda_kernel.cu', 8, 116)]

_HEAD = r'''
#include <cstdint>
#include <cstddef>
#include <cmath>
#define __global__
#define __device__
#define __inline__ inline
#define __restrict__
struct idx3 { unsigned x, y, z; };
static thread_local idx3 blockIdx, blockDim, threadIdx;
static inline double max(float a, double b) { return fmax((double)a, b); }
static inline float  max(float a, float b)  { return fmaxf(a, b); }
static inline float  min(float a, float b)  { return fminf(a, b); }
static inline double min(float a, double b) { return fmin((double)a, b); }
static inline int    min(int a, int b) { return a < b ? a : b; }
static inline int    max(int a, int b) { return a > b ? a : b; }
'''

_TAIL = r'''
extern "C" {
void ref_load_textures(const float* image, const int32_t* is_update, float* faces, float* textures, int textures_size,
                       int texture_size, int image_height, int image_width, int wrapping, int use_bilinear) {
    blockDim.x = 1; threadIdx.x = 0;
    for (long i = 0; i < textures_size / 3; i++) {
        blockIdx.x = (unsigned)i;
        load_textures_cuda_kernel<float>(image, is_update, faces, textures, textures_size, texture_size, image_height,
                                         image_width, wrapping, (bool)use_bilinear);
    }
}
void ref_create_texture_image(const float* vertices_all, const float* textures, float* image, int image_size, int num_faces,
                              int tsi, int tso, int tile_width, float eps) {
    blockDim.x = 1; threadIdx.x = 0;
    for (long i = 0; i < image_size / 3; i++) {
        blockIdx.x = (unsigned)i;
        create_texture_image_cuda_kernel<float>(vertices_all, textures, image, (size_t)image_size, (size_t)num_faces,
                                                (size_t)tsi, (size_t)tso, (size_t)tile_width, eps);
    }
    for (long i = 0; i < image_size / 3; i++) {
        blockIdx.x = (unsigned)i;
        create_texture_image_boundary_cuda_kernel<float>(image, (size_t)image_size, (size_t)tso, (size_t)tile_width);
    }
}
}
'''

_lib = None


def build():
    global _lib
    if _lib is not None:
        return _lib
    tmp = tempfile.mkdtemp(prefix='rnr_ref_texkernels_')
    src = os.path.join(tmp, 'ref_tex_kernels.cpp')
    with open(src, 'w') as out:
        out.write(_HEAD)
        for path, first, last in PARTS:
            with open(path) as fh:
                out.writelines(fh.readlines()[first - 1:last])
        out.write(_TAIL)
    so = os.path.join(tmp, 'ref_tex_kernels.so')
    subprocess.check_call(['g++', '-O2', '-ffp-contract=off', '-shared', '-fPIC', '-w', src, '-o', so])
    _lib = ctypes.CDLL(so)
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _c(a, dt=np.float32):
    return np.ascontiguousarray(a, dt)


def load_textures(image, faces_uv, textures, is_update, wrapping, use_bilinear):
    lib = build()
    image, f, t, upd = _c(image), _c(faces_uv).copy(), _c(textures).copy(), _c(is_update, np.int32)
    lib.ref_load_textures(_p(image), _p(upd), _p(f), _p(t), t.size, t.shape[1], image.shape[0], image.shape[1],
                          int(wrapping), int(use_bilinear))
    return t, f


def create_texture_image(vertices_all, textures, image_hw, eps):
    lib = build()
    v, t = _c(vertices_all), _c(textures)
    nf = t.shape[0]
    tw = int((nf - 1) ** 0.5) + 1
    img = np.zeros((int(image_hw[0]), int(image_hw[1]), 3), np.float32)
    lib.ref_create_texture_image(_p(v), _p(t), _p(img), img.size, nf, t.shape[1], img.shape[1] // tw, tw,
                                 ctypes.c_float(eps))
    return img


if __name__ == '__main__':
    build()
    print('ok')
