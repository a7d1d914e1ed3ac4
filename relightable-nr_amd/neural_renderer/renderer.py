"""nr.Renderer (reference: neural_renderer/renderer.py:12-257), camera_mode='projection' only — the one mode the
relightable-nr scripts use (network.py:145-149).  render() returns the same 8-tuple (renderer.py:257)."""
import numpy
import torch
import torch.nn as nn

import neural_renderer as nr


class Renderer(nn.Module):
    def __init__(self, image_size=256, anti_aliasing=True, background_color=[0, 0, 0], fill_back=True,
                 camera_mode='projection', K=None, R=None, t=None, dist_coeffs=None, orig_size=1024, offset=None,
                 scale=None, perspective=True, viewing_angle=30, camera_direction=[0, 0, 1], near=0.1, far=100,
                 light_intensity_ambient=0.5, light_intensity_directional=0.5, light_color_ambient=[1, 1, 1],
                 light_color_directional=[1, 1, 1], light_direction=[0, 1, 0]):
        super().__init__()
        if camera_mode != 'projection':
            if camera_mode in ('look', 'look_at'):
                raise NotImplementedError("camera_mode '%s' is outside the hot-path build; use 'projection'" % camera_mode)
            raise ValueError('Camera mode has to be one of projection, look or look_at')
        self.image_size, self.anti_aliasing = image_size, anti_aliasing
        self.background_color, self.fill_back = background_color, fill_back
        self.camera_mode = camera_mode
        as_t = lambda x: torch.as_tensor(x, dtype=torch.float32).cuda() if isinstance(x, numpy.ndarray) else x
        self.K, self.R, self.t = as_t(K), as_t(R), as_t(t)
        self.dist_coeffs = dist_coeffs     # None -> zeros at render time (renderer.py:41-42)
        self.orig_size, self.offset, self.scale = orig_size, offset, scale
        self.near, self.far = near, far
        self.light_intensity_ambient = light_intensity_ambient
        self.light_intensity_directional = light_intensity_directional
        self.light_color_ambient = light_color_ambient
        self.light_color_directional = light_color_directional
        self.light_direction = light_direction
        self.rasterizer_eps = 1e-3

    def forward(self, vertices, faces, textures=None, mode=None, K=None, R=None, t=None, dist_coeffs=None,
                orig_size=None, offset=None, scale=None):
        if mode is None:
            return self.render(vertices, faces, textures, K, R, t, dist_coeffs, orig_size, offset=offset, scale=scale)
        raise NotImplementedError("Renderer mode '%s' is outside the hot-path build (only mode=None is used)" % mode)

    def render(self, vertices, faces, textures, K=None, R=None, t=None, dist_coeffs=None, orig_size=None, offset=None,
               scale=None):
        if self.fill_back:      # renderer.py:209-211
            faces = torch.cat((faces, faces.flip(-1)), dim=1).detach()
            textures = torch.cat((textures, textures.permute((0, 1, 4, 3, 2, 5))), dim=1)
        faces_lighting = nr.vertices_to_faces(vertices, faces)
        textures = nr.lighting(faces_lighting, textures, self.light_intensity_ambient, self.light_intensity_directional,
                               self.light_color_ambient, self.light_color_directional, self.light_direction)
        K = self.K if K is None else K
        R = self.R if R is None else R
        t = self.t if t is None else t
        dist_coeffs = self.dist_coeffs if dist_coeffs is None else dist_coeffs
        if dist_coeffs is None:
            dist_coeffs = torch.zeros(1, 5, device=vertices.device)
        orig_size = self.orig_size if orig_size is None else orig_size
        offset = self.offset if offset is None else offset
        scale = self.scale if scale is None else scale
        vertices = nr.projection(vertices, K, R, t, dist_coeffs, orig_size, offset=offset, scale=scale)
        faces_v = nr.vertices_to_faces(vertices, faces)
        if textures.shape[0] == 1 and faces_v.shape[0] != 1:
            textures = textures.expand(faces_v.shape[0], *textures.shape[1:])   # the reference reads out of bounds here
        out = nr.rasterize_rgbad(faces_v, textures, self.image_size, self.anti_aliasing, self.near, self.far,
                                 self.rasterizer_eps, self.background_color)
        return out['rgb'], out['depth'], out['alpha'], out['face_index_map'], out['weight_map'], vertices, faces_v, faces
