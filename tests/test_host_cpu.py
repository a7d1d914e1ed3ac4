"""CPU: host-side utilities (synthetic scene / weights) agree with the oracle's reading of the reference."""
import numpy as np
import torch

from oracle import rnr_oracle as orc
from rnr_amd import scene, testing


def test_ray_pivots_match_oracle():
    for n_azi, n_pol, step in [(6, 2, 5), (6, 2, 10), (4, 3, 7)]:
        _, piv = orc.ray_sampler_pivots(n_azi, n_pol, step)
        assert torch.allclose(testing.ray_pivots(n_azi, n_pol, step), piv, atol=1e-7)


def test_synthetic_state_dict_runs_through_oracle_unet():
    sd = testing.unet_state_dict(20, 6, 4, out_channels_gcn=16)
    y = orc.unet_forward(sd, torch.randn(1, 20, 32, 32))
    assert y.shape == (1, 6, 32, 32) and torch.isfinite(y).all() and y.abs().max() <= 1.0
    sd5 = testing.unet_state_dict(20, 6, 4, num_down=3, use_gcn=False)
    assert orc.unet_forward(sd5, torch.randn(1, 20, 16, 16), num_down=3).shape == (1, 6, 16, 16)


def test_spiral_views_consistent():
    v = scene.spiral_views(128, [0, 10, 719])
    assert v['pose'].shape == (3, 4, 4)
    for i in range(3):
        R = v['pose'][i, :3, :3]
        assert np.allclose(R.dot(R.T), np.eye(3), atol=1e-5)
        assert np.allclose(v['proj'][i].dot(v['proj_inv'][i]), np.eye(3), atol=1e-4)
        cam_pos = -R.T.dot(v['pose'][i, :3, 3])
        assert abs(np.linalg.norm(cam_pos) - 3.0) < 1e-4


def test_obj_roundtrip(tmp_path):
    m = scene.uv_sphere(4, 8)
    p = tmp_path / 's.obj'
    scene.write_obj(str(p), m)
    txt = p.read_text().splitlines()
    assert sum(l.startswith('f ') for l in txt) == m['f_v_idx'].shape[0]
    assert sum(l.startswith('v ') for l in txt) == m['v'].shape[0]
