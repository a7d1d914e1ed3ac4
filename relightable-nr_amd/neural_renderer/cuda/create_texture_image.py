"""`neural_renderer.cuda.create_texture_image` (reference: create_texture_image_cuda.cpp:17-33) on the HIP kernels."""
from rnr_amd.ops import create_texture_image  # noqa: F401
