"""`neural_renderer.cuda.load_textures` (reference: load_textures_cuda.cpp:20-38) on the HIP kernel of librnr_hip.so."""
from rnr_amd.ops import load_textures  # noqa: F401
