"""nr.rasterize_rgbad / nr.Rasterize / RasterizeFunction (reference: neural_renderer/rasterize.py:15-340).

Buffer protocol identical to RasterizeFunction.forward (rasterize.py:50-100): face_index -1, weights 0, depth = far,
alpha = (face_index >= 0), background blend, then the vertical flip and optional 2x SSAA of rasterize_rgbad
(rasterize.py:296-330).  The flip is a tensor.flip() here instead of index-list gathers."""
import torch
import torch.nn as nn
import torch.nn.functional as F

import neural_renderer.cuda.rasterize as rasterize_cuda

DEFAULT_IMAGE_SIZE = 256
DEFAULT_ANTI_ALIASING = True
DEFAULT_NEAR = 0.1
DEFAULT_FAR = 100
DEFAULT_EPS = 1e-4
DEFAULT_BACKGROUND_COLOR = (0, 0, 0)


class RasterizeFunction(torch.autograd.Function):
    """Forward and backward of the rasterizer on the HIP kernels (rasterize.py:15-231).  Differentiable with respect
    to `faces` (x, y through the silhouette sweep, x, y, z through the depth map) and `textures`."""

    @staticmethod
    def forward(ctx, faces, textures, image_size, near, far, eps, background_color, return_rgb=False,
                return_alpha=False, return_depth=False):
        if not faces.is_cuda or (textures is not None and not textures.is_cuda):
            raise TypeError('Rasterize module supports only cuda Tensors')
        ctx.image_size, ctx.near, ctx.far, ctx.eps = image_size, near, far, eps
        ctx.return_rgb, ctx.return_alpha, ctx.return_depth = return_rgb, return_alpha, return_depth
        faces = faces.detach().float().contiguous().clone()
        dev = faces.device
        B, nf = faces.shape[:2]
        ctx.num_faces = nf
        S = image_size
        face_index_map = torch.full((B, S, S), -1, dtype=torch.int32, device=dev)
        weight_map = torch.zeros(B, S, S, 3, device=dev)
        depth_map = torch.full((B, S, S), float(far), dtype=torch.float32, device=dev)
        face_inv_map = torch.zeros(B, S, S, 3, 3, device=dev) if return_depth else torch.zeros(1, device=dev)
        faces_inv = torch.zeros_like(faces)
        rasterize_cuda.forward_face_index_map(faces, face_index_map, weight_map, depth_map, face_inv_map, faces_inv, S,
                                              near, far, return_rgb, return_alpha, return_depth)
        empty = torch.tensor([])
        rgb_map = sidx = swgt = torch.zeros(1, device=dev)
        alpha_map = torch.zeros(1, device=dev)
        if return_rgb:
            textures = textures.detach().float().contiguous()
            rgb_map = torch.zeros(B, S, S, 3, device=dev)
            sidx = torch.zeros(B, S, S, 8, dtype=torch.int32, device=dev)
            swgt = torch.zeros(B, S, S, 8, device=dev)
            rasterize_cuda.forward_texture_sampling(faces, textures, face_index_map, weight_map, depth_map, rgb_map,
                                                    sidx, swgt, S, eps)
            bg = torch.tensor(background_color, dtype=torch.float32, device=dev)
            mask = (face_index_map >= 0).float()[..., None]
            bg = bg[None, None, None, :] if bg.ndimension() == 1 else bg[:, None, None, :]
            rgb_map = rgb_map * mask + (1 - mask) * bg
        else:
            textures = torch.zeros(1, device=dev)
        if return_alpha:
            alpha_map = (face_index_map >= 0).float()
        ctx.save_for_backward(faces, textures, face_index_map, weight_map, depth_map, rgb_map, alpha_map, face_inv_map,
                              sidx, swgt)
        ctx.mark_non_differentiable(face_index_map)
        return (rgb_map if return_rgb else empty, alpha_map.clone() if return_alpha else empty,
                depth_map.clone() if return_depth else empty, face_index_map.clone(), weight_map.clone())

    @staticmethod
    def backward(ctx, grad_rgb_map, grad_alpha_map, grad_depth_map, grad_face_index_map, grad_weight_map):
        (faces, textures, face_index_map, weight_map, depth_map, rgb_map, alpha_map, face_inv_map, sidx,
         swgt) = ctx.saved_tensors
        grad_faces = torch.zeros_like(faces)
        grad_textures = None

        def grad_or_zeros(g, like):
            return g.float().contiguous() if g is not None else torch.zeros_like(like)
        if ctx.return_rgb:
            grad_rgb_map = grad_or_zeros(grad_rgb_map, rgb_map)
        if ctx.return_alpha:
            grad_alpha_map = grad_or_zeros(grad_alpha_map, alpha_map)
        if ctx.return_rgb or ctx.return_alpha:
            rasterize_cuda.backward_pixel_map(faces, face_index_map, rgb_map, alpha_map, grad_rgb_map, grad_alpha_map,
                                              grad_faces, ctx.image_size, ctx.eps, ctx.return_rgb, ctx.return_alpha)
        if ctx.return_rgb and ctx.needs_input_grad[1]:
            grad_textures = torch.zeros_like(textures)
            rasterize_cuda.backward_textures(face_index_map, swgt, sidx, grad_rgb_map, grad_textures, ctx.num_faces)
        if ctx.return_depth:
            grad_depth_map = grad_or_zeros(grad_depth_map, depth_map)
            rasterize_cuda.backward_depth_map(faces, depth_map, face_index_map, face_inv_map, weight_map,
                                              grad_depth_map, grad_faces, ctx.image_size)
        return grad_faces, grad_textures, None, None, None, None, None, None, None, None


class Rasterize(nn.Module):
    def __init__(self, image_size, near, far, eps, background_color, return_rgb=False, return_alpha=False,
                 return_depth=False):
        super().__init__()
        self.image_size, self.near, self.far, self.eps = image_size, near, far, eps
        self.background_color = background_color
        self.return_rgb, self.return_alpha, self.return_depth = return_rgb, return_alpha, return_depth

    def forward(self, faces, textures):
        return RasterizeFunction.apply(faces, textures, self.image_size, self.near, self.far, self.eps,
                                       self.background_color, self.return_rgb, self.return_alpha, self.return_depth)


def rasterize_rgbad(faces, textures=None, image_size=DEFAULT_IMAGE_SIZE, anti_aliasing=DEFAULT_ANTI_ALIASING,
                    near=DEFAULT_NEAR, far=DEFAULT_FAR, eps=DEFAULT_EPS, background_color=DEFAULT_BACKGROUND_COLOR,
                    return_rgb=True, return_alpha=True, return_depth=True):
    size = image_size * 2 if anti_aliasing else image_size
    rgb, alpha, depth, face_index_map, weight_map = Rasterize(size, near, far, eps, background_color, return_rgb,
                                                              return_alpha, return_depth)(faces, textures)
    if return_rgb:
        rgb = rgb.permute(0, 3, 1, 2).flip(2)
    if return_alpha:
        alpha = alpha.flip(1)
    if return_depth:
        depth = depth.flip(1)
    face_index_map = face_index_map.flip(1)
    weight_map = weight_map.flip(1)
    if anti_aliasing:
        if return_rgb:
            rgb = F.avg_pool2d(rgb, kernel_size=(2, 2))
        if return_alpha:
            alpha = F.avg_pool2d(alpha[:, None], kernel_size=(2, 2))[:, 0]
        if return_depth:
            depth = F.avg_pool2d(depth[:, None], kernel_size=(2, 2))[:, 0]
    return {'rgb': rgb if return_rgb else None, 'alpha': alpha if return_alpha else None,
            'depth': depth if return_depth else None, 'face_index_map': face_index_map, 'weight_map': weight_map}


def rasterize(faces, textures, image_size=DEFAULT_IMAGE_SIZE, anti_aliasing=DEFAULT_ANTI_ALIASING, near=DEFAULT_NEAR,
              far=DEFAULT_FAR, eps=DEFAULT_EPS, background_color=DEFAULT_BACKGROUND_COLOR):
    return rasterize_rgbad(faces, textures, image_size, anti_aliasing, near, far, eps, background_color, True, False,
                           False)['rgb']


def rasterize_silhouettes(faces, image_size=DEFAULT_IMAGE_SIZE, anti_aliasing=DEFAULT_ANTI_ALIASING, near=DEFAULT_NEAR,
                          far=DEFAULT_FAR, eps=DEFAULT_EPS):
    return rasterize_rgbad(faces, None, image_size, anti_aliasing, near, far, eps, None, False, True, False)['alpha']


def rasterize_depth(faces, image_size=DEFAULT_IMAGE_SIZE, anti_aliasing=DEFAULT_ANTI_ALIASING, near=DEFAULT_NEAR,
                    far=DEFAULT_FAR, eps=DEFAULT_EPS):
    return rasterize_rgbad(faces, None, image_size, anti_aliasing, near, far, eps, None, False, False, True)['depth']
