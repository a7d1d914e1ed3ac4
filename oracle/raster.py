"""ctypes front-end of oracle/raster_oracle.c (ORACLE, test infrastructure).

numpy in / numpy out; buffers are allocated and pre-filled exactly as RasterizeFunction.forward does
(rasterize.py:50-69: face_index -1, weight 0, depth far, faces_inv zeros per rasterize.py:163).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, '_build', 'libraster_oracle.so')
_lib = None


def build(force=False):
    """Compile the C restatement (gcc).  Building the checker is not using it."""
    if force or not os.path.isfile(_SO) or os.path.getmtime(_SO) < os.path.getmtime(os.path.join(_HERE, 'raster_oracle.c')):
        subprocess.check_call(['make', '-C', _HERE, '-s', '-B'])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def face_setup(faces, image_size):
    faces = np.ascontiguousarray(faces, np.float32)
    B, nf = faces.shape[:2]
    faces_inv = np.zeros((B, nf, 9), np.float32)
    lib().oracle_face_setup(_p(faces), _p(faces_inv), B, nf, int(image_size))
    return faces_inv


def face_index_map(faces, image_size, near, far, return_depth=True):
    """faces [B,nf,3,3] (NDC x,y + camera z) -> dict of UNFLIPPED maps, as the extension returns them."""
    faces = np.ascontiguousarray(faces, np.float32)
    B, nf = faces.shape[:2]
    S = int(image_size)
    faces_inv = face_setup(faces, S)
    fim = np.full((B, S, S), -1, np.int32)
    wm = np.zeros((B, S, S, 3), np.float32)
    dm = np.full((B, S, S), far, np.float32)
    fivm = np.zeros((B, S, S, 3, 3), np.float32)
    lib().oracle_face_index_map(_p(faces), _p(faces_inv), _p(fim), _p(wm), _p(dm), _p(fivm), B, nf, S,
                                ctypes.c_float(near), ctypes.c_float(far), int(bool(return_depth)))
    return {'faces_inv': faces_inv, 'face_index_map': fim, 'weight_map': wm, 'depth_map': dm,
            'face_inv_map': fivm}


def texture_sampling(faces, textures, fim, wm, dm, image_size, eps):
    faces = np.ascontiguousarray(faces, np.float32)
    textures = np.ascontiguousarray(textures, np.float32)
    B, nf = faces.shape[:2]
    S = int(image_size)
    ts = textures.shape[2]
    rgb = np.zeros((B, S, S, 3), np.float32)
    sim = np.zeros((B, S, S, 8), np.int32)
    swm = np.zeros((B, S, S, 8), np.float32)
    lib().oracle_texture_sampling(_p(faces), _p(textures), _p(np.ascontiguousarray(fim, np.int32)),
                                  _p(np.ascontiguousarray(wm, np.float32)),
                                  _p(np.ascontiguousarray(dm, np.float32)), _p(rgb), _p(sim), _p(swm),
                                  B, nf, S, ts, ctypes.c_float(eps))
    return {'rgb_map': rgb, 'sampling_index_map': sim, 'sampling_weight_map': swm}


def _c(a, dt=np.float32):
    return np.ascontiguousarray(a, dt)


def backward_pixel_map(faces, fim, rgb_map, alpha_map, grad_rgb_map, grad_alpha_map, image_size, eps,
                       return_rgb=True, return_alpha=True):
    """-> grad_faces [B,nf,3,3] (zeros where the reference leaves the caller's zeros: back faces)."""
    faces = _c(faces)
    B, nf = faces.shape[:2]
    gf = np.zeros((B, nf, 3, 3), np.float32)
    args = [_c(rgb_map), _c(alpha_map), _c(grad_rgb_map), _c(grad_alpha_map)]
    fim = _c(fim, np.int32)
    lib().oracle_backward_pixel_map(_p(faces), _p(fim), _p(args[0]), _p(args[1]), _p(args[2]), _p(args[3]), _p(gf),
                                    B, nf, int(image_size), ctypes.c_float(eps), int(return_rgb), int(return_alpha))
    return gf


def backward_textures(fim, sampling_weight_map, sampling_index_map, grad_rgb_map, num_faces, texture_size):
    fim = _c(fim, np.int32)
    B, S = fim.shape[:2]
    ts = int(texture_size)
    gt = np.zeros((B, num_faces, ts, ts, ts, 3), np.float32)
    swm, sim, g = _c(sampling_weight_map), _c(sampling_index_map, np.int32), _c(grad_rgb_map)
    lib().oracle_backward_textures(_p(fim), _p(swm), _p(sim), _p(g), _p(gt), B, int(num_faces), S, ts)
    return gt


def backward_depth_map(faces, depth_map, fim, face_inv_map, weight_map, grad_depth_map, image_size, grad_faces=None):
    faces = _c(faces)
    B, nf = faces.shape[:2]
    gf = np.zeros((B, nf, 3, 3), np.float32) if grad_faces is None else _c(grad_faces).copy()
    dm, fim, fivm, wm, gd = _c(depth_map), _c(fim, np.int32), _c(face_inv_map), _c(weight_map), _c(grad_depth_map)
    lib().oracle_backward_depth_map(_p(faces), _p(dm), _p(fim), _p(fivm), _p(wm), _p(gd), _p(gf), B, nf, int(image_size))
    return gf


def load_textures(image, faces_uv, textures, is_update, wrapping, use_bilinear):
    """-> (textures, wrapped faces_uv); inputs are not modified."""
    image, f, t = _c(image), _c(faces_uv).copy(), _c(textures).copy()
    upd = _c(is_update, np.int32)
    lib().oracle_load_textures(_p(image), _p(f), _p(t), _p(upd), f.shape[0], t.shape[1], image.shape[0], image.shape[1],
                               int(wrapping), int(bool(use_bilinear)))
    return t, f


def create_texture_image(vertices_all, textures, image_hw, eps):
    v, t = _c(vertices_all), _c(textures)
    img = np.zeros((int(image_hw[0]), int(image_hw[1]), 3), np.float32)
    lib().oracle_create_texture_image(_p(v), _p(t), _p(img), t.shape[0], t.shape[1], img.shape[0], img.shape[1],
                                      ctypes.c_float(eps))
    return img
