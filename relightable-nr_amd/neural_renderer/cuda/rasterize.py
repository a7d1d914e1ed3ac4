"""`neural_renderer.cuda.rasterize` (reference: rasterize_cuda.cpp:124-191, 5 functions).

forward_face_index_map / forward_texture_sampling keep the extension's contract: the caller allocates and pre-fills
every buffer, kernels write in place, the same tensors are returned.  The three backward functions belong to
training of geometry, which is out of scope; they raise."""
from rnr_amd.ops import forward_face_index_map, forward_texture_sampling  # noqa: F401


def _no_backward(name):
    def f(*a, **k):
        raise NotImplementedError('%s: rasterizer backward kernels are out of scope of the inference hot path' % name)
    f.__name__ = name
    return f


backward_pixel_map = _no_backward('backward_pixel_map')
backward_textures = _no_backward('backward_textures')
backward_depth_map = _no_backward('backward_depth_map')
