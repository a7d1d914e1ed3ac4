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
