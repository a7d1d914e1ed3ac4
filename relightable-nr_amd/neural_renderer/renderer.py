"""nr.Renderer (reference: neural_renderer/renderer.py:12-257): camera modes 'projection' (the one relightable-nr uses,
network.py:145-149), 'look_at' and 'look'; render() returns the reference's 8-tuple (renderer.py:257), the modes
'rgb' / 'silhouettes' / 'depth' their single map.  Rasterization runs on the HIP kernels of librnr_hip.so."""
import math

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
        if camera_mode not in ('projection', 'look', 'look_at'):
            raise ValueError('Camera mode has to be one of projection, look or look_at')
        self.image_size, self.anti_aliasing = image_size, anti_aliasing
        self.background_color, self.fill_back = background_color, fill_back
        self.camera_mode = camera_mode
        if camera_mode == 'projection':
            as_t = lambda x: torch.as_tensor(x, dtype=torch.float32).cuda() if isinstance(x, numpy.ndarray) else x
            self.K, self.R, self.t = as_t(K), as_t(R), as_t(t)
            self.dist_coeffs = dist_coeffs     # None -> zeros at render time (renderer.py:41-42)
            self.orig_size, self.offset, self.scale = orig_size, offset, scale
        else:                                  # renderer.py:54-58: the eye sits on -z so that the unit cube fills the view
            self.perspective = perspective
            self.viewing_angle = viewing_angle
            self.eye = [0, 0, -(1. / math.tan(math.radians(viewing_angle)) + 1)]
            self.camera_direction = [0, 0, 1]
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
        if mode == 'rgb':
            return self.render_rgb(vertices, faces, textures, K, R, t, dist_coeffs, orig_size)
        if mode == 'silhouettes':
            return self.render_silhouettes(vertices, faces, K, R, t, dist_coeffs, orig_size)
        if mode == 'depth':
            return self.render_depth(vertices, faces, K, R, t, dist_coeffs, orig_size)
        raise ValueError("mode should be one of None, 'rgb', 'silhouettes' or 'depth'")

    # ---- shared pieces -------------------------------------------------------------------------------------------
    def _both_sides(self, faces, textures=None):
        if self.fill_back:      # renderer.py:209-211: every face once more with reversed winding
            faces = torch.cat((faces, faces.flip(-1)), dim=1).detach()
            if textures is not None:
                textures = torch.cat((textures, textures.permute((0, 1, 4, 3, 2, 5))), dim=1)
        return faces, textures

    def _lit(self, vertices, faces, textures):
        return nr.lighting(nr.vertices_to_faces(vertices, faces), textures, self.light_intensity_ambient,
                           self.light_intensity_directional, self.light_color_ambient, self.light_color_directional,
                           self.light_direction)

    def _to_view(self, vertices, K, R, t, dist_coeffs, orig_size, offset=None, scale=None):
        if self.camera_mode == 'look_at':
            vertices = nr.look_at(vertices, self.eye)
        elif self.camera_mode == 'look':
            vertices = nr.look(vertices, self.eye, self.camera_direction)
        else:
            K = self.K if K is None else K
            R = self.R if R is None else R
            t = self.t if t is None else t
            dist_coeffs = self.dist_coeffs if dist_coeffs is None else dist_coeffs
            if dist_coeffs is None:
                dist_coeffs = torch.zeros(1, 5, device=vertices.device)
            orig_size = self.orig_size if orig_size is None else orig_size
            offset = self.offset if offset is None else offset
            scale = self.scale if scale is None else scale
            return nr.projection(vertices, K, R, t, dist_coeffs, orig_size, offset=offset, scale=scale)
        return nr.perspective(vertices, angle=self.viewing_angle) if self.perspective else vertices

    # ---- render modes --------------------------------------------------------------------------------------------
    def render_silhouettes(self, vertices, faces, K=None, R=None, t=None, dist_coeffs=None, orig_size=None):
        faces, _ = self._both_sides(faces)
        faces_v = nr.vertices_to_faces(self._to_view(vertices, K, R, t, dist_coeffs, orig_size), faces)
        return nr.rasterize_silhouettes(faces_v, self.image_size, self.anti_aliasing)

    def render_depth(self, vertices, faces, K=None, R=None, t=None, dist_coeffs=None, orig_size=None):
        faces, _ = self._both_sides(faces)
        faces_v = nr.vertices_to_faces(self._to_view(vertices, K, R, t, dist_coeffs, orig_size), faces)
        return nr.rasterize_depth(faces_v, self.image_size, self.anti_aliasing)

    def render_rgb(self, vertices, faces, textures, K=None, R=None, t=None, dist_coeffs=None, orig_size=None):
        faces, textures = self._both_sides(faces, textures)
        textures = self._lit(vertices, faces, textures)
        faces_v = nr.vertices_to_faces(self._to_view(vertices, K, R, t, dist_coeffs, orig_size), faces)
        return nr.rasterize(faces_v, textures, self.image_size, self.anti_aliasing, self.near, self.far,
                            self.rasterizer_eps, self.background_color)

    def render(self, vertices, faces, textures, K=None, R=None, t=None, dist_coeffs=None, orig_size=None, offset=None,
               scale=None):
        faces, textures = self._both_sides(faces, textures)
        textures = self._lit(vertices, faces, textures)
        vertices = self._to_view(vertices, K, R, t, dist_coeffs, orig_size, offset, scale)
        faces_v = nr.vertices_to_faces(vertices, faces)
        if textures.shape[0] == 1 and faces_v.shape[0] != 1:
            textures = textures.expand(faces_v.shape[0], *textures.shape[1:])   # the reference reads out of bounds here
        out = nr.rasterize_rgbad(faces_v, textures, self.image_size, self.anti_aliasing, self.near, self.far,
                                 self.rasterizer_eps, self.background_color)
        return out['rgb'], out['depth'], out['alpha'], out['face_index_map'], out['weight_map'], vertices, faces_v, faces
