"""nr.Mesh: a triangle mesh with an optional learnable per-face texture cube (reference: neural_renderer/mesh.py:7-43)."""
import torch
from torch.nn import Parameter

import neural_renderer as nr


class Mesh(object):
    def __init__(self, vertices, faces, textures=None, texture_size=4):
        """vertices [nv, 3], faces [nf, 3]; textures [nf, ts, ts, ts, 3] or None for a small random Parameter."""
        self.vertices, self.faces = vertices, faces
        self.num_vertices, self.num_faces = len(vertices), len(faces)
        if textures is not None:
            # kept (the reference drops a texture that is passed in, mesh.py:26-27)
            self.textures, self.texture_size = textures, textures.shape[1]
        else:
            cube = (self.num_faces,) + (texture_size,) * 3 + (3,)
            self.textures, self.texture_size = Parameter(torch.randn(cube) * 0.05), texture_size

    @classmethod
    def fromobj(cls, filename_obj, normalization=True, load_texture=False, texture_size=4):
        """Mesh from a Wavefront OBJ; nr.load_obj of this code base returns attribute dictionaries
        (load_obj.py:203-209), optionally followed by the baked textures."""
        loaded = nr.load_obj(filename_obj, normalization=normalization, texture_size=texture_size,
                             load_texture=load_texture)
        v_attr, f_attr = loaded[0], loaded[1]
        return cls(v_attr['v'], f_attr['f_v_idx'], loaded[2] if load_texture else None, texture_size)
