"""`neural_renderer.cuda.rasterize` (reference: rasterize_cuda.cpp:124-191, 5 functions).

Every function keeps the extension's contract: the caller allocates and pre-fills every buffer, kernels write in
place, the same tensors are returned.  All five run HIP kernels of librnr_hip.so (include/rnr_hip.h section 1).

Scalar types.  The reference dispatches on `faces.type()` with AT_DISPATCH_FLOATING_TYPES (rasterize_cuda_kernel.cu:614,
628, 669, 712, 749, 784): float32 or float64 kernels.  Its own Python layer can only reach the float32 instances
(`RasterizeFunction.forward` allocates `torch.cuda.FloatTensor` buffers, rasterize.py:50-69; a float64 `faces` would
fail the `data<double>()` access on them), so float64 is reachable only by calling the extension directly.  The HIP
kernels are float32; float64 arguments are ACCEPTED here by running the float32 kernels on converted copies and writing
the results back into the caller's float64 buffers (in place, same tensors returned) — with a warning, because the
arithmetic is then float32: maps equal the float32 kernels' results, not what a float64 instantiation would give
(index flips are possible on edge pixels of faces whose float64 coordinates do not survive the narrowing).
Mixed float32 / float64 arguments raise, as the reference's typed accessors would."""
import functools
import warnings

import torch

from rnr_amd import ops

_warned = [False]


def _accept_float64(fn, outputs):
    """outputs: positions of the caller-allocated float buffers `fn` writes (only those are copied back)."""
    @functools.wraps(fn)
    def wrapper(*args):
        kinds = {a.dtype for a in args if isinstance(a, torch.Tensor) and a.is_floating_point() and a.numel() > 1}
        if torch.float64 not in kinds:
            return fn(*args)
        if kinds != {torch.float64}:
            raise RuntimeError('%s: float32 and float64 tensors mixed (the extension dispatches on ONE scalar type, '
                               'rasterize_cuda_kernel.cu:614)' % fn.__name__)
        if not _warned[0]:
            warnings.warn('neural_renderer.cuda.rasterize: float64 tensors are computed by the float32 HIP kernels on '
                          'converted copies (results are written back in place as float64)', RuntimeWarning, stacklevel=2)
            _warned[0] = True
        conv = [a.float().contiguous() if (isinstance(a, torch.Tensor) and a.dtype == torch.float64) else a for a in args]
        out = fn(*conv)
        back = {}
        for i, (a, c) in enumerate(zip(args, conv)):
            if isinstance(a, torch.Tensor) and a.dtype == torch.float64:
                if i in outputs:
                    a.copy_(c)                   # caller-allocated buffers are updated in place; inputs stay untouched
                back[id(c)] = a
        if isinstance(out, (list, tuple)):
            return type(out)(back.get(id(o), o) for o in out)
        return back.get(id(out), out)
    return wrapper


# argument positions as in rasterize_cuda.cpp:66-189
forward_face_index_map = _accept_float64(ops.forward_face_index_map, outputs=(2, 3, 4, 5))     # weight, depth, face_inv_map, faces_inv
forward_texture_sampling = _accept_float64(ops.forward_texture_sampling, outputs=(5, 7))      # rgb_map, sampling_weight_map
backward_pixel_map = _accept_float64(ops.backward_pixel_map, outputs=(6,))                    # grad_faces
backward_textures = _accept_float64(ops.backward_textures, outputs=(4,))                      # grad_textures
backward_depth_map = _accept_float64(ops.backward_depth_map, outputs=(6,))                    # grad_faces
