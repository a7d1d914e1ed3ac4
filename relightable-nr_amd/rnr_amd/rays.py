"""Ray pivot directions of network.RaySampler (network.py:418-443): the 13 + 13 directions the light-transport rays of a
pixel are built from (train_rnr.py:344-354 defaults: 6 azimuths x 2 polar rings at 5 / 10 degrees, plus the axis)."""
import math

import numpy as np
import torch


def ray_pivots(num_azi, num_polar, interval_polar):
    """RaySampler.__init__ (network.py:418-443; data_util.euler_to_rot, data_util.py:175-191): pivots_dir [3,R]."""
    pol = np.arange(1, num_polar + 1) * interval_polar * np.pi / 180.0
    azi = np.arange(num_azi) * 2 * np.pi / num_azi
    pol, azi = np.meshgrid(pol, azi)
    pol, azi = pol.flatten(), azi.flatten()
    Rs = np.zeros((pol.shape[0] + 1, 3, 3), np.float32)
    Rs[0] = np.eye(3)
    for i in range(pol.shape[0]):
        cy, sy, cz, sz = math.cos(pol[i]), math.sin(pol[i]), math.cos(azi[i]), math.sin(azi[i])
        ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
        rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
        Rs[i + 1] = rz.dot(ry)
    Rs = torch.from_numpy(Rs)
    return torch.matmul(Rs, torch.tensor([0.0, 0.0, 1.0])[:, None])[..., 0].permute(1, 0).contiguous()
