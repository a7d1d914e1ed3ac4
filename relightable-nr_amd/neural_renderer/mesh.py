"""nr.Mesh (reference: neural_renderer/mesh.py:7-43): vertices + faces (+ a learnable per-face texture cube)."""
import torch
import torch.nn as nn

import neural_renderer as nr


class Mesh(object):
    def __init__(self, vertices, faces, textures=None, texture_size=4):
        self.vertices, self.faces = vertices, faces
        self.num_vertices, self.num_faces = vertices.shape[0], faces.shape[0]
        if textures is None:
            self.textures = nn.Parameter(0.05 * torch.randn(self.num_faces, texture_size, texture_size, texture_size, 3))
            self.texture_size = texture_size
        else:
            self.textures = textures            # the reference forgets to keep a given texture (mesh.py:26-27)
            self.texture_size = textures.shape[1]

    @classmethod
    def fromobj(cls, filename_obj, normalization=True, load_texture=False, texture_size=4):
        """From a Wavefront OBJ.  nr.load_obj of this code base returns attribute dictionaries (load_obj.py:203-209)."""
        if load_texture:
            v_attr, f_attr, textures = nr.load_obj(filename_obj, normalization=normalization, texture_size=texture_size,
                                                   load_texture=True)
        else:
            v_attr, f_attr = nr.load_obj(filename_obj, normalization=normalization, texture_size=texture_size)
            textures = None
        return cls(v_attr['v'], f_attr['f_v_idx'], textures, texture_size)
