"""ctypes binding of librnr_hip.so (the C ABI declared in include/rnr_hip.h).

The HIP library is the product: there is no CPU fallback.  If the shared object is missing or a symbol
cannot be resolved this module raises — callers never silently route elsewhere.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# RNR_HIP_LIB: load another build of the same library (kernel ablation experiments); default is the in-tree build
LIB_PATH = os.environ.get('RNR_HIP_LIB') or os.path.join(os.path.dirname(_HERE), 'librnr_hip.so')

c_void_p, c_int, c_float, c_size_t, c_double = (ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_size_t,
                                                ctypes.c_double)


class RnrMesh(ctypes.Structure):
    _fields_ = [('v', c_void_p), ('vt', c_void_p), ('vn', c_void_p), ('f_v_idx', c_void_p),
                ('f_vt_idx', c_void_p), ('f_vn_idx', c_void_p), ('num_vertices', c_int),
                ('num_texcoords', c_int), ('num_normals', c_int), ('num_faces', c_int)]


class RnrGbuffer(ctypes.Structure):
    _fields_ = [(n, c_void_p) for n in ['face_index_map', 'alpha', 'depth', 'weight_map', 'raw_weight_map',
                                        'uv_map', 'normal_map', 'normal_map_cam', 'position_map',
                                        'position_map_cam']]


class RnrObjCounts(ctypes.Structure):
    _fields_ = [(n, ctypes.c_long) for n in ['num_vertices', 'num_normals', 'num_texcoords', 'num_faces']]


class RnrRays(ctypes.Structure):
    _fields_ = [('pivots_spec_host', c_void_p), ('pivots_diff_host', c_void_p), ('num_spec', c_int),
                ('num_diff', c_int)]


class RnrConvSrc(ctypes.Structure):
    _fields_ = [('data', c_void_p), ('scale', c_void_p), ('shift', c_void_p), ('channels', c_int), ('act', c_int)]


class RnrConvDesc(ctypes.Structure):
    _fields_ = [('kind', c_int), ('c_in0', c_int), ('c_in0_pad', c_int), ('c_in1', c_int), ('c_in1_pad', c_int),
                ('c_out', c_int), ('c_out_pad', c_int), ('flags', c_int)]


class RnrConvBn(ctypes.Structure):
    _fields_ = [('gamma', c_void_p), ('beta', c_void_p), ('scale', c_void_p), ('shift', c_void_p), ('eps', c_float),
                ('running_mean', c_void_p), ('running_var', c_void_p), ('momentum', c_float)]      # NULL / 0: no running-statistics update


ACT_NONE, ACT_LRELU02, ACT_RELU = 0, 1, 2
CONV_STATS_PREZEROED = 1
CONV_F32_EMU_BF16X6 = 2
CONV_F32_EMU_F16X3 = 4
CONV_WINOGRAD = 8        # Winograd F(2x2, 3x3) for the 3x3 convolutions (exact-fp32 operands, include/rnr_hip.h)
CONV_WINOGRAD4 = 16      # with CONV_WINOGRAD: F(4x4, 3x3) where the shape allows (set by UNetPlan's default conv_algo 'winograd4'; include/rnr_hip.h)
EMU_FLAGS = {'f32': 0, 'bf16x6': CONV_F32_EMU_BF16X6, 'f16x3': CONV_F32_EMU_F16X3}
CONV3x3_REFLECT, CONV4x4S2_REFLECT, CONVT4x4S2 = 0, 1, 2

P = ctypes.POINTER
# name -> (restype, argtypes); must list every symbol of include/rnr_hip.h (tests/test_abi.py checks it)
SIGNATURES = {
    'rnr_abi_version': (c_int, []),
    'rnr_last_error': (ctypes.c_char_p, []),
    'rnr_raster_workspace_bytes': (c_size_t, [c_int, c_int, c_int]),
    'rnr_forward_face_index_map': (c_int, [c_void_p] * 6 + [c_int, c_int, c_int, c_float, c_float, c_int, c_int,
                                                            c_int, c_void_p, c_void_p]),
    'rnr_forward_texture_sampling': (c_int, [c_void_p] * 8 + [c_int, c_int, c_int, c_int, c_float, c_void_p]),
    'rnr_backward_pixel_map': (c_int, [c_void_p] * 7 + [c_int, c_int, c_int, c_float, c_int, c_int, c_void_p]),
    'rnr_backward_textures': (c_int, [c_void_p] * 5 + [c_int, c_int, c_int, c_int, c_void_p]),
    'rnr_backward_depth_map': (c_int, [c_void_p] * 7 + [c_int, c_int, c_int, c_void_p]),
    'rnr_load_textures': (c_int, [c_void_p] * 4 + [c_int] * 6 + [c_void_p]),
    'rnr_create_texture_image': (c_int, [c_void_p] * 3 + [c_int] * 4 + [c_float, c_void_p]),
    'rnr_project_vertices': (c_int, [c_void_p] * 8 + [c_int, c_int, c_float, c_float, c_void_p]),
    'rnr_gbuffer_workspace_bytes': (c_size_t, [c_int, c_int, c_int]),
    'rnr_rasterize_gbuffer': (c_int, [P(RnrMesh), c_void_p, c_void_p, c_int, c_int, c_float, c_float, P(RnrGbuffer),
                                      c_void_p, c_void_p]),
    'rnr_rasterize_gbuffer_prepared': (c_int, [P(RnrMesh), c_void_p, c_void_p, c_int, c_int, c_float, c_float, P(RnrGbuffer),
                                               c_void_p, c_void_p]),
    'rnr_frame_prepare': (c_int, [P(RnrMesh), c_void_p, c_void_p, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p, c_void_p,
                                  c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    'rnr_face_tangents': (c_int, [P(RnrMesh), c_void_p, c_void_p]),
    'rnr_shade_inputs': (c_int, [c_void_p] * 5 + [c_int, c_void_p, c_void_p, P(c_void_p), P(c_int), c_int, c_int,
                                                  c_int, P(RnrRays), c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                                                  c_int, c_int, c_int, c_void_p]),
    'rnr_packed_weight_floats': (c_size_t, [P(RnrConvDesc)]),
    'rnr_pack_conv_weight': (c_int, [P(RnrConvDesc), c_void_p, c_void_p, c_void_p]),
    'rnr_conv_workspace_bytes': (c_size_t, [P(RnrConvDesc), c_int, c_int, c_int]),
    'rnr_conv2d': (c_int, [P(RnrConvDesc), P(RnrConvSrc), P(RnrConvSrc), c_void_p, c_void_p, c_void_p, c_int, c_int,
                           c_int, c_void_p, c_size_t, c_void_p]),
    'rnr_conv_tile_count': (c_size_t, [P(RnrConvDesc), c_int, c_int, c_int]),
    'rnr_conv_algorithm': (c_int, [P(RnrConvDesc), c_int, c_int, c_int]),
    'rnr_conv_active_tiles': (c_int, [P(RnrConvDesc), c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    'rnr_conv2d_masked': (c_int, [P(RnrConvDesc), P(RnrConvSrc), P(RnrConvSrc), c_void_p, c_void_p, c_void_p, c_int, c_int,
                                  c_int, c_void_p, c_size_t, c_void_p, c_void_p]),
    'rnr_conv_sync_bytes': (c_size_t, [P(RnrConvDesc), c_int, c_int, c_int]),
    'rnr_conv2d_fused': (c_int, [P(RnrConvDesc), P(RnrConvSrc), P(RnrConvSrc), c_void_p, c_void_p, P(RnrConvBn), c_int, c_int,
                                 c_int, c_void_p, c_size_t, c_void_p, c_size_t, c_void_p, c_void_p]),
    'rnr_ray_weights': (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_int,
                                c_int, c_int, c_int, c_void_p]),
    'rnr_conv2d_ray': (c_int, [P(RnrConvDesc), P(RnrConvSrc), P(RnrConvSrc), c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
                               c_int, c_void_p, c_void_p]),
    'rnr_bn_finalize': (c_int, [c_void_p] * 5 + [c_int, c_int, c_int, c_double, c_float, c_void_p]),
    'rnr_bn_finalize_reset': (c_int, [c_void_p] * 5 + [c_int, c_int, c_int, c_double, c_float, c_void_p]),
    'rnr_bn_finalize_batch': (c_int, [c_void_p] * 7 + [c_float, c_int, c_int, c_int, c_double, c_float, c_void_p]),
    'rnr_nchw_to_nhwc': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    'rnr_nhwc_to_nchw': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    'rnr_ray_render': (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int,
                               c_int, c_int, c_int, c_void_p, c_int, c_int, c_int, c_void_p]),
    'rnr_sh_basis': (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p]),
    'rnr_sh_reconstruct': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    'rnr_sh_fit': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    'rnr_interpolate_bilinear': (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                         c_int, c_void_p]),
    'rnr_resize_area': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    'rnr_calibrate_mfma_f32': (c_int, [c_int, c_int, c_void_p, P(ctypes.c_double), P(ctypes.c_double), c_void_p]),
    'rnr_obj_scan': (c_int, [ctypes.c_char_p, c_size_t, P(RnrObjCounts)]),
    'rnr_obj_parse': (c_int, [ctypes.c_char_p, c_size_t, P(RnrObjCounts)] + [c_void_p] * 6),
    'rnr_view_dir_map': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    'rnr_tbn_map': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_void_p]),
    'rnr_tbn_matvec': (c_int, [c_void_p, c_void_p, c_void_p, ctypes.c_long, c_int, c_void_p]),
    'rnr_ray_sampler': (c_int, [c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                ctypes.c_long, c_void_p]),
    'rnr_texture_mapper': (c_int, [c_void_p, c_void_p, P(c_void_p), P(c_int), c_int, c_int, c_int, c_void_p, c_int,
                                   c_int, c_int, c_void_p]),
    'rnr_ray_renderer': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_int,
                                 c_int, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                 c_void_p, c_int, c_int, c_int, c_void_p]),
}

_lib = None


class RnrError(RuntimeError):
    pass


def _build_from_source():
    """A source checkout without the built library: compile it with hipcc if the ROCm toolchain is there (still the
    HIP path — there is nothing else to fall back to).  Safe under torchrun: an exclusive file lock serialises the
    ranks (the first one builds, the others find the library once they get the lock), the Makefile links to a
    temporary name and renames it into place.  Returns the captured build log on failure, None otherwise."""
    import fcntl
    import shutil
    import subprocess
    hipcc = shutil.which('hipcc') or ('/opt/rocm/bin/hipcc' if os.path.isfile('/opt/rocm/bin/hipcc') else None)
    if hipcc is None:
        return 'hipcc not found (PATH, /opt/rocm/bin)'
    csrc = os.path.join(os.path.dirname(_HERE), 'csrc')
    try:
        lock = open(os.path.join(csrc, '.build.lock'), 'w')
    except OSError as e:
        return 'cannot create build lock in %s: %s' % (csrc, e)
    with lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if os.path.isfile(LIB_PATH):        # another rank built it while we waited
                return None
            p = subprocess.run(['make', '-C', csrc, '-s', '-j4', 'HIPCC=' + hipcc], capture_output=True, text=True)
            if p.returncode != 0:
                return 'make exited with %d\n%s\n%s' % (p.returncode, p.stdout[-2000:], p.stderr[-4000:])
            return None
        except OSError as e:
            return 'could not run make: %s' % e
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def load():
    """Load librnr_hip.so (does not need a GPU; launching kernels does)."""
    global _lib
    if _lib is not None:
        return _lib
    log = None
    if not os.path.isfile(LIB_PATH) and not os.environ.get('RNR_HIP_LIB'):
        log = _build_from_source()
    if not os.path.isfile(LIB_PATH):
        raise RnrError('librnr_hip.so not found at %s - build it first: `python -c "import __graft_entry__ as g; '
                       'g.build()"` or `make -C relightable-nr_amd/csrc`. There is no CPU fallback.%s'
                       % (LIB_PATH, ('\nautomatic build failed: ' + log) if log else ''))
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)     # AttributeError if the symbol is missing: fail loudly
        fn.restype = res
        fn.argtypes = args
    ver = lib.rnr_abi_version()
    if ver != 1:
        raise RnrError('librnr_hip.so ABI version %d, python binding expects 1' % ver)
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        raise RnrError(load().rnr_last_error().decode('utf-8', 'replace'))
