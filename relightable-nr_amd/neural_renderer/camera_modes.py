"""nr.look_at / nr.look / nr.perspective / nr.get_points_from_angles (reference: neural_renderer/look_at.py:6-62,
look.py:6-53, perspective.py:6-21, get_points_from_angles.py:6-24).  Plain tensor algebra on the vertices' device; not on
the relightable-nr hot path (its cameras come from calib.mat as K, R, t), provided so the package is whole."""
import math

import numpy as np
import torch
import torch.nn.functional as F


def _vec(x, device, batch):
    """list / tuple / ndarray / tensor -> float tensor [batch or 1, 3] on `device`."""
    if isinstance(x, (list, tuple)):
        x = torch.tensor(x, dtype=torch.float32, device=device)
    elif isinstance(x, np.ndarray):
        x = torch.from_numpy(x).to(device)
    else:
        x = x.to(device)
    if x.ndimension() == 1:
        x = x[None, :].repeat(batch, 1) if batch else x[None, :]
    return x


def _rotate(vertices, eye, z_axis, up):
    """Rows of the camera rotation: x = up x z, y = z x x, z (all normalised with eps 1e-5, look_at.py:49-52)."""
    z_axis = F.normalize(z_axis, eps=1e-5)
    x_axis = F.normalize(torch.cross(up, z_axis, dim=-1), eps=1e-5)
    y_axis = F.normalize(torch.cross(z_axis, x_axis, dim=-1), eps=1e-5)
    r = torch.stack((x_axis, y_axis, z_axis), dim=1)
    if vertices.shape != eye.shape:
        eye = eye[:, None, :]
    return torch.matmul(vertices - eye, r.transpose(1, 2))


def look_at(vertices, eye, at=[0, 0, 0], up=[0, 1, 0]):
    """Camera at `eye` looking at the point `at`: vertices [B, nv, 3] -> camera coordinates."""
    if vertices.ndimension() != 3:
        raise ValueError('vertices Tensor should have 3 dimensions')
    B, dev = vertices.shape[0], vertices.device
    eye, at, up = _vec(eye, dev, B), _vec(at, dev, B), _vec(up, dev, B)
    return _rotate(vertices, eye, at - eye, up)


def look(vertices, eye, direction=[0, 1, 0], up=None):
    """Camera at `eye` looking along `direction`."""
    if vertices.ndimension() != 3:
        raise ValueError('vertices Tensor should have 3 dimensions')
    dev = vertices.device
    up = [0, 1, 0] if up is None else up
    eye, direction, up = _vec(eye, dev, 0), _vec(direction, dev, 0), _vec(up, dev, 0)
    return _rotate(vertices, eye, direction, up)


def perspective(vertices, angle=30.):
    """x, y divided by z * tan(angle): the unit-square frustum of half-angle `angle` degrees."""
    if vertices.ndimension() != 3:
        raise ValueError('vertices Tensor should have 3 dimensions')
    width = torch.tan(torch.tensor(angle / 180 * math.pi, dtype=torch.float32, device=vertices.device))[None][:, None]
    z = vertices[:, :, 2]
    return torch.stack((vertices[:, :, 0] / z / width, vertices[:, :, 1] / z / width, z), dim=2)


def get_points_from_angles(distance, elevation, azimuth, degrees=True):
    """Point on a sphere around the origin, y up, azimuth measured from -z (the package's camera convention)."""
    if isinstance(distance, (float, int)):
        if degrees:
            elevation, azimuth = math.radians(elevation), math.radians(azimuth)
        return (distance * math.cos(elevation) * math.sin(azimuth), distance * math.sin(elevation),
                -distance * math.cos(elevation) * math.cos(azimuth))
    if degrees:
        elevation, azimuth = math.pi / 180. * elevation, math.pi / 180. * azimuth
    return torch.stack([distance * torch.cos(elevation) * torch.sin(azimuth), distance * torch.sin(elevation),
                        -distance * torch.cos(elevation) * torch.cos(azimuth)]).transpose(1, 0)
