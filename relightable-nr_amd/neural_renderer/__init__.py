"""Drop-in `neural_renderer` for MI355X: same names as the reference package
(neural_renderer/neural_renderer/__init__.py:1-13) for the parts the relightable-nr scripts use
(`nr.load_obj`, `nr.Renderer`, `nr.projection`, `nr.lighting`, `nr.vertices_to_faces`,
`nr.vertex_attrs_to_faces`, `nr.rasterize_rgbad`, `nr.Rasterize`, ...), backed by librnr_hip.so.

The rasterizer is differentiable (RasterizeFunction: HIP backward kernels for the rgb/alpha/depth maps and textures).
Textured OBJ loading / saving (load_obj(load_texture=True), save_obj) run the load_textures / create_texture_image
HIP kernels; look / look_at / perspective / get_points_from_angles / Mesh are plain tensor utilities.
"""
from .lighting import lighting
from .load_obj import load_obj
from .save_obj import save_obj
from .projection import projection
from .rasterize import (rasterize_rgbad, rasterize, rasterize_silhouettes, rasterize_depth, Rasterize)
from .renderer import Renderer
from .vertices_to_faces import vertices_to_faces
from .vertices_to_faces import vertex_attrs_to_faces


from .camera_modes import look_at, look, perspective, get_points_from_angles
from .mesh import Mesh

__version__ = '1.1.3+rnr_hip'
name = 'neural_renderer'
