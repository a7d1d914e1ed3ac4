"""Drop-in `camera` (reference: camera.py)."""
import numpy as np
import torch

from rnr_amd import ops, scene


def get_view_dir_map(img_size, proj_inv, R_inv):
    """camera.py:5-32 -> (view_dir_map world [N,H,W,3], view_dir_map_cam [N,H,W,3]) on the HIP kernel."""
    return ops.view_dir_map(img_size, proj_inv.float().contiguous(), R_inv.float().contiguous())


def get_reflect_dir(orig_dir, pivot_dir, dim=-1):
    """camera.py:35-45 (generic-shape helper; the ray sampler kernels have it fused in)."""
    return torch.nn.functional.normalize((pivot_dir * orig_dir).sum(dim=dim, keepdim=True) * 2.0 * pivot_dir - orig_dir,
                                         dim=dim)


def RT_from_pos_lookat(cam_pos, cam_lookat=np.array([0., 0., 0.]), cam_up=np.array([0., 1., 0.])):
    """camera.py:48-69."""
    return scene.rt_from_pos_lookat(cam_pos, cam_lookat, cam_up).astype(np.asarray(cam_pos).dtype)


def get_spiral(step_azi=-2, step_ele=90.0 / 720):
    """camera.py:72-75."""
    return scene.spiral_angles(step_azi, step_ele)
