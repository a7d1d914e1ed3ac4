"""`neural_renderer.cuda.rasterize` (reference: rasterize_cuda.cpp:124-191, 5 functions).

Every function keeps the extension's contract: the caller allocates and pre-fills every buffer, kernels write in
place, the same tensors are returned.  All five run HIP kernels of librnr_hip.so (include/rnr_hip.h section 1)."""
from rnr_amd.ops import (forward_face_index_map, forward_texture_sampling, backward_pixel_map,  # noqa: F401
                         backward_textures, backward_depth_map)
