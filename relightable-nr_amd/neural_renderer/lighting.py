"""nr.lighting (reference: neural_renderer/lighting.py:5-57): ambient + directional shading of per-face texture
cubes.  On the hot path it multiplies an all-zero texture by 1 (network.py:150-151); kept for API completeness."""
import numpy as np
import torch
import torch.nn.functional as F


def _as_row(x, device):
    if isinstance(x, (tuple, list)):
        x = torch.tensor(x, dtype=torch.float32, device=device)
    elif isinstance(x, np.ndarray):
        x = torch.from_numpy(x).float().to(device)
    return x[None, :] if x.ndimension() == 1 else x


def lighting(faces, textures, intensity_ambient=0.5, intensity_directional=0.5, color_ambient=(1, 1, 1),
             color_directional=(1, 1, 1), direction=(0, 1, 0)):
    bs, nf = faces.shape[:2]
    dev = faces.device
    ca, cd, di = _as_row(color_ambient, dev), _as_row(color_directional, dev), _as_row(direction, dev)
    light = torch.zeros(bs, nf, 3, dtype=torch.float32, device=dev)
    if intensity_ambient != 0:
        light = light + intensity_ambient * ca[:, None, :]
    if intensity_directional != 0:
        f = faces.reshape(bs * nf, 3, 3)
        n = F.normalize(torch.cross(f[:, 0] - f[:, 1], f[:, 2] - f[:, 1], dim=-1), eps=1e-5).reshape(bs, nf, 3)
        cos = F.relu((n * di[:, None, :]).sum(2))
        light = light + intensity_directional * (cd[:, None, :] * cos[:, :, None])
    textures *= light[:, :, None, None, None, :]     # in place, like the reference
    return textures
