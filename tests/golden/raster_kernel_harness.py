"""Runs the reference's forward rasterizer KERNEL BODIES on the CPU (build container only).

The reference rasterizer has no CPU path (neural_renderer/rasterize.py:17-19) and its extension needs
CUDA headers / nvcc that this image lacks, so a reference build (`oracle/_ref`) is not possible.
To still pin the oracle to the reference's own arithmetic, this harness copies
lines 23-592 of /root/reference/neural_renderer/neural_renderer/cuda/rasterize_cuda_kernel.cu
(the three forward and three backward __global__ templates) into a TEMP directory, prepends a thread-index shim
(blockIdx/blockDim/threadIdx as thread-locals, CUDA's fmin/fmax-semantics min/max overloads), and
drives the kernels with a serial loop over thread ids.  Nothing of the reference is written into
the repository: only the produced (input, output) vectors are committed by make_golden.py.
"""
import ctypes
import os
import subprocess
import tempfile

import numpy as np

REF_CU = '/root/reference/neural_renderer/neural_renderer/cuda/rasterize_cuda_kernel.cu'
FIRST, LAST = 23, 592

_SHIM_HEAD = r'''
#include <cstdint>
#include <cstddef>
#include <cmath>
#define __global__
#define __restrict__
struct idx3 { unsigned x, y, z; };
static thread_local idx3 blockIdx, blockDim, threadIdx;
/* CUDA device min/max: fmin/fmax semantics (the non-NaN operand is returned); mixed
   float/double calls promote to double (rasterize_cuda_kernel.cu:128, 210-211). */
static inline double max(float a, double b) { return fmax((double)a, b); }
static inline double max(double a, double b) { return fmax(a, b); }
static inline float  max(float a, float b)  { return fmaxf(a, b); }
static inline double min(double a, double b) { return fmin(a, b); }
static inline double min(float a, double b) { return fmin((double)a, b); }
static inline float  min(float a, float b)  { return fminf(a, b); }
static inline int    max(int a, int b) { return a > b ? a : b; }
static inline int    min(int a, int b) { return a < b ? a : b; }
/* the driver below is serial, so a plain add has atomicAdd's semantics */
template <typename T> static inline void atomicAdd(T* p, T v) { *p += v; }
namespace {
'''

_SHIM_TAIL = r'''
}  // namespace
extern "C" {
void ref_face_index_map(const float* faces, float* faces_inv, int32_t* face_index_map, float* weight_map,
                        float* depth_map, float* face_inv_map, int batch_size, int num_faces, int image_size,
                        float near, float far, int return_rgb, int return_alpha, int return_depth) {
    blockDim.x = 1; threadIdx.x = 0;
    for (long i = 0; i < (long)batch_size * num_faces; i++) {
        blockIdx.x = (unsigned)i;
        forward_face_index_map_cuda_kernel_1<float>(faces, faces_inv, batch_size, num_faces, image_size);
    }
    #pragma omp parallel for schedule(dynamic, 256)
    for (long i = 0; i < (long)batch_size * image_size * image_size; i++) {
        blockDim.x = 1; threadIdx.x = 0; blockIdx.x = (unsigned)i;
        forward_face_index_map_cuda_kernel_2<float>(faces, faces_inv, face_index_map, weight_map, depth_map,
            face_inv_map, batch_size, num_faces, image_size, near, far, return_rgb, return_alpha, return_depth);
    }
}
void ref_texture_sampling(const float* faces, const float* textures, const int32_t* face_index_map,
                          const float* weight_map, const float* depth_map, float* rgb_map,
                          int32_t* sampling_index_map, float* sampling_weight_map, int batch_size,
                          int num_faces, int image_size, int texture_size, float eps) {
    blockDim.x = 1; threadIdx.x = 0;
    for (long i = 0; i < (long)batch_size * image_size * image_size; i++) {
        blockIdx.x = (unsigned)i;
        forward_texture_sampling_cuda_kernel<float>(faces, textures, face_index_map, weight_map, depth_map,
            rgb_map, sampling_index_map, sampling_weight_map, (size_t)batch_size, num_faces, image_size,
            texture_size, eps);
    }
}
void ref_backward_pixel_map(const float* faces, int32_t* face_index_map, float* rgb_map, float* alpha_map,
                            float* grad_rgb_map, float* grad_alpha_map, float* grad_faces, int batch_size,
                            int num_faces, int image_size, float eps, int return_rgb, int return_alpha) {
    blockDim.x = 1; threadIdx.x = 0;
    for (long i = 0; i < (long)batch_size * num_faces; i++) {
        blockIdx.x = (unsigned)i;
        backward_pixel_map_cuda_kernel<float>(faces, face_index_map, rgb_map, alpha_map, grad_rgb_map, grad_alpha_map,
            grad_faces, (size_t)batch_size, (size_t)num_faces, image_size, eps, return_rgb, return_alpha);
    }
}
void ref_backward_textures(const int32_t* face_index_map, float* sampling_weight_map, int32_t* sampling_index_map,
                           float* grad_rgb_map, float* grad_textures, int batch_size, int num_faces, int image_size,
                           int texture_size) {
    blockDim.x = 1; threadIdx.x = 0;
    for (long i = 0; i < (long)batch_size * image_size * image_size; i++) {
        blockIdx.x = (unsigned)i;
        backward_textures_cuda_kernel<float>(face_index_map, sampling_weight_map, sampling_index_map, grad_rgb_map,
            grad_textures, (size_t)batch_size, (size_t)num_faces, image_size, (size_t)texture_size);
    }
}
void ref_backward_depth_map(const float* faces, const float* depth_map, const int32_t* face_index_map,
                            const float* face_inv_map, const float* weight_map, float* grad_depth_map,
                            float* grad_faces, int batch_size, int num_faces, int image_size) {
    blockDim.x = 1; threadIdx.x = 0;
    for (long i = 0; i < (long)batch_size * image_size * image_size; i++) {
        blockIdx.x = (unsigned)i;
        backward_depth_map_cuda_kernel<float>(faces, depth_map, face_index_map, face_inv_map, weight_map,
            grad_depth_map, grad_faces, (size_t)batch_size, (size_t)num_faces, image_size);
    }
}
}
'''

_libs = {}

# How the reference itself is compiled: neural_renderer/setup.py:14-27 passes no flags to nvcc, whose default is
# --fmad=true (a*b+c contracted into one fused multiply-add wherever the compiler sees it).  The committed fixtures
# are generated WITHOUT contraction (bit-reproducible on any host); the `fma=True` build contracts like nvcc does
# (`-ffp-contract=fast -mfma`) and is used to show that the integer outputs do not depend on that choice
# (make_golden.gen_raster_fma).
FLAGS = {False: ['-ffp-contract=off'], True: ['-ffp-contract=fast', '-mfma']}


CLANGXX = '/opt/rocm/lib/llvm/bin/clang++'     # LLVM, like nvcc's NVVM back end: its contraction pattern is the closer stand-in


def build(fma=False, cxx='g++'):
    fma = bool(fma)
    key = fma if cxx == 'g++' else (fma, cxx)
    if key in _libs:
        return _libs[key]
    tmp = tempfile.mkdtemp(prefix='rnr_ref_kernels_')
    with open(REF_CU) as fh:
        body = fh.readlines()[FIRST - 1:LAST]
    src = os.path.join(tmp, 'ref_kernels.cpp')
    with open(src, 'w') as fh:
        fh.write(_SHIM_HEAD)
        fh.writelines(body)
        fh.write(_SHIM_TAIL)
    so = os.path.join(tmp, 'ref_kernels.so')
    subprocess.check_call([cxx, '-O2'] + FLAGS[fma] + ['-fopenmp', '-shared', '-fPIC', '-w', src, '-o', so])
    if fma:
        # the build must really contain fused multiply-adds, otherwise the comparison proves nothing
        dis = subprocess.run(['objdump', '-d', so], capture_output=True, text=True).stdout
        assert 'vfmadd' in dis or 'vfnmadd' in dis or 'vfmsub' in dis, 'FMA build holds no fused instruction'
    _libs[key] = ctypes.CDLL(so)
    return _libs[key]


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def face_index_map(faces, image_size, near, far, return_depth=1, fma=False, cxx='g++'):
    """faces [B,nf,3,3] float32 -> dict of the kernel outputs (UNFLIPPED, as the extension returns)."""
    lib = build(fma, cxx)
    faces = np.ascontiguousarray(faces, np.float32)
    B, nf = faces.shape[:2]
    S = image_size
    faces_inv = np.zeros_like(faces).reshape(B, nf, 9)
    fim = np.full((B, S, S), -1, np.int32)
    wm = np.zeros((B, S, S, 3), np.float32)
    dm = np.full((B, S, S), far, np.float32)
    fivm = np.zeros((B, S, S, 3, 3), np.float32)
    lib.ref_face_index_map(_p(faces), _p(faces_inv), _p(fim), _p(wm), _p(dm), _p(fivm),
                           B, nf, S, ctypes.c_float(near), ctypes.c_float(far), 1, 1, int(return_depth))
    return {'faces_inv': faces_inv, 'face_index_map': fim, 'weight_map': wm, 'depth_map': dm,
            'face_inv_map': fivm}


def texture_sampling(faces, textures, fim, wm, dm, image_size, eps):
    lib = build()
    faces = np.ascontiguousarray(faces, np.float32)
    textures = np.ascontiguousarray(textures, np.float32)
    B, nf = faces.shape[:2]
    S = image_size
    ts = textures.shape[2]
    rgb = np.zeros((B, S, S, 3), np.float32)
    sim = np.zeros((B, S, S, 8), np.int32)
    swm = np.zeros((B, S, S, 8), np.float32)
    lib.ref_texture_sampling(_p(faces), _p(textures), _p(np.ascontiguousarray(fim)), _p(np.ascontiguousarray(wm)),
                             _p(np.ascontiguousarray(dm)), _p(rgb), _p(sim), _p(swm), B, nf, S, ts,
                             ctypes.c_float(eps))
    return {'rgb_map': rgb, 'sampling_index_map': sim, 'sampling_weight_map': swm}


def _c(a, dt=np.float32):
    return np.ascontiguousarray(a, dt)


def backward_pixel_map(faces, fim, rgb_map, alpha_map, grad_rgb_map, grad_alpha_map, image_size, eps,
                       return_rgb=1, return_alpha=1):
    lib = build()
    faces = _c(faces)
    B, nf = faces.shape[:2]
    gf = np.zeros((B, nf, 3, 3), np.float32)
    a = [_c(fim, np.int32), _c(rgb_map), _c(alpha_map), _c(grad_rgb_map), _c(grad_alpha_map)]
    lib.ref_backward_pixel_map(_p(faces), _p(a[0]), _p(a[1]), _p(a[2]), _p(a[3]), _p(a[4]), _p(gf), B, nf,
                               int(image_size), ctypes.c_float(eps), int(return_rgb), int(return_alpha))
    return gf


def backward_textures(fim, swm, sim, grad_rgb_map, num_faces, texture_size):
    lib = build()
    fim = _c(fim, np.int32)
    B, S = fim.shape[:2]
    ts = int(texture_size)
    gt = np.zeros((B, num_faces, ts, ts, ts, 3), np.float32)
    a = [_c(swm), _c(sim, np.int32), _c(grad_rgb_map)]
    lib.ref_backward_textures(_p(fim), _p(a[0]), _p(a[1]), _p(a[2]), _p(gt), B, int(num_faces), S, ts)
    return gt


def backward_depth_map(faces, depth_map, fim, face_inv_map, weight_map, grad_depth_map, image_size):
    lib = build()
    faces = _c(faces)
    B, nf = faces.shape[:2]
    gf = np.zeros((B, nf, 3, 3), np.float32)
    a = [_c(depth_map), _c(fim, np.int32), _c(face_inv_map), _c(weight_map), _c(grad_depth_map)]
    lib.ref_backward_depth_map(_p(faces), _p(a[0]), _p(a[1]), _p(a[2]), _p(a[3]), _p(a[4]), _p(gf), B, nf,
                               int(image_size))
    return gf


if __name__ == '__main__':
    build()
    print('ok')
