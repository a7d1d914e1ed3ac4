"""Drop-in `render` (reference: render.py): the functions the inference path uses."""
import numpy as np
import torch

from rnr_amd import ops


def spherical_mapping(l_dir):
    """render.py:87-93.  [3,...] -> [2,...] equirect uv (y-up)."""
    return torch.stack((torch.atan2(l_dir[2], l_dir[0]) * 0.5 / np.pi + 0.5, torch.acos(l_dir[1]) * 1.0 / np.pi), dim=0)


def spherical_mapping_batch(l_dir):
    """render.py:96-102.  [N,3,...] -> [N,2,...]."""
    return torch.stack((torch.atan2(l_dir[:, 2], l_dir[:, 0]) * 0.5 / np.pi + 0.5, torch.acos(l_dir[:, 1]) * 1.0 / np.pi),
                       dim=1)


def spherical_mapping_inv(lp_samples_uv):
    """render.py:105-121.  uv [2,n] -> unit directions [3,n] (init-time helper of LightingSH)."""
    y = torch.cos(lp_samples_uv[1] * np.pi)
    s = (1 - y ** 2).sqrt()
    a = lp_samples_uv[0] * 2 - 1
    x = s * torch.cos(a * np.pi)
    z = s * torch.sin(a * np.pi)
    z = z * ((~(a == 1.0)).to(s.dtype) * 2 - 1)
    z = z * ((~(a == -1.0)).to(s.dtype) * 2 - 1)
    return torch.nn.functional.normalize(torch.stack((x, y, z), dim=0), dim=0)


def get_TBN_map(normal_map, face_index_map, faces_v=None, faces_texcoord=None, tangent=None, check_nan=False):
    """render.py:124-168 -> [N,H,W,3,3].  `check_nan=True` restores the reference's NaN guards (3 host syncs per call,
    ValueError('nan value detected')); they are off by default to keep the stream asynchronous."""
    if tangent is None:
        assert faces_v is not None and faces_texcoord is not None
        tangent = ops.face_tangents(faces_v, faces_texcoord)
    else:
        tangent = torch.nn.functional.normalize(tangent.float(), dim=-1).contiguous()
    if check_nan and torch.isnan(tangent).sum() > 0:
        raise ValueError('nan value detected')
    tbn = ops.tbn_map(normal_map.float().contiguous(), face_index_map.int().contiguous(), tangent)
    if check_nan and torch.isnan(tbn).sum() > 0:
        raise ValueError('nan value detected')
    return tbn


def _out_of_scope(name):
    def f(*a, **k):
        raise NotImplementedError('render.%s is not used by the inference hot path (SURVEY.md §2.1)' % name)
    f.__name__ = name
    return f


interp_vertex_attr = _out_of_scope('interp_vertex_attr')
texture_mapping = _out_of_scope('texture_mapping')
lp_mapping = _out_of_scope('lp_mapping')
sample_light_dir = _out_of_scope('sample_light_dir')
get_TBN_map_perpixel = _out_of_scope('get_TBN_map_perpixel')
