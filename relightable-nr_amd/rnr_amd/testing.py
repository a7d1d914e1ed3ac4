"""Seeded synthetic weights / textures / lighting for smoke(), bench.py and the tests:
the reference ships neither checkpoints nor data.  Pure torch-CPU / numpy; no reference code involved."""
import math

import numpy as np
import torch

from . import scene


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


def unet_state_dict(in_channels, out_channels, nf0, num_down=5, seed=0, use_gcn=True, out_channels_gcn=512):
    """Random-init weights with the reference's RenderingNet key names / shapes, PyTorch-default-like
    scales (uniform +-1/sqrt(fan_in)), BN gamma ~ U(0.75,1.25), beta ~ U(-0.25,0.25)."""
    g = torch.Generator().manual_seed(seed)
    sd = {}

    def conv(key, cout, cin, k, bias):
        bound = 1.0 / math.sqrt(cin * k * k)
        sd[key + '.weight'] = (torch.rand(cout, cin, k, k, generator=g) * 2 - 1) * bound
        if bias:
            sd[key + '.bias'] = (torch.rand(cout, generator=g) * 2 - 1) * bound

    def convT(key, cin, cout, bias):
        bound = 1.0 / math.sqrt(cout * 16)
        sd[key + '.weight'] = (torch.rand(cin, cout, 4, 4, generator=g) * 2 - 1) * bound
        if bias:
            sd[key + '.bias'] = (torch.rand(cout, generator=g) * 2 - 1) * bound

    def bn(key, c):
        sd[key + '.weight'] = 0.75 + 0.5 * torch.rand(c, generator=g)
        sd[key + '.bias'] = 0.5 * torch.rand(c, generator=g) - 0.25

    p = 'net.'
    conv(p + 'in_layer.0.net.1', nf0, in_channels, 3, False)
    bn(p + 'in_layer.1', nf0)
    max_c = 8 * nf0

    def block(path, outer, depth):
        d, u = path + 'down.net.', path + 'up.net.'
        if depth == num_down - 1:
            conv(d + '1', outer, outer, 3, True)
            conv(d + '5', outer, outer, 4, True)
            convT(u + '0', outer, outer, True)
            conv(u + '3.net.1', outer, outer, 3, True)
        else:
            inner = min(2 ** (depth + 1) * nf0, max_c)
            conv(d + '1', outer, outer, 3, False); bn(d + '2', outer)
            conv(d + '6', inner, outer, 4, False); bn(d + '7', inner)
            block(path + 'submodule.', inner, depth + 1)
            convT(u + '0', 2 * inner, outer, False); bn(u + '1', outer)
            conv(u + '4.net.1', outer, outer, 3, False); bn(u + '5', outer)

    block(p + 'unet_block.', nf0, 0)
    if use_gcn:      # dead at the output (pytorch_prototyping.py:407-419); present so that strict loading matches
        inner = min(2 * nf0, max_c)
        conv(p + 'unet_block.fuse.net.1', inner + out_channels_gcn, inner + out_channels_gcn, 3, False)
        bn(p + 'unet_block.fuse.net.2', inner + out_channels_gcn)
        conv(p + 'unet_block.fuse.net.6', inner, inner + out_channels_gcn, 3, False)
        bn(p + 'unet_block.fuse.net.7', inner)
    conv(p + 'out_layer.0.net.1', out_channels, 2 * nf0, 3, True)
    return sd


def synthetic_textures(tex_size, tex_ch, levels=4, seed=0):
    g = torch.Generator().manual_seed(seed + 1000)
    out = []
    for l in range(levels):
        s = int(np.round(tex_size / (2.0 ** l)))
        out.append(torch.rand(1, s, s, tex_ch, generator=g) * (1.0 if l == 0 else 0.01))
    return out


def synthetic_light_probe(h=100, w=200, seed=2):
    """Smooth positive env map (sum of Gaussians on the equirect grid) — stands in for LightingSH.reconstruct_lp."""
    rng = np.random.RandomState(seed)
    vv, uu = np.meshgrid(np.arange(h) / (h - 1.0), np.arange(w) / (w - 1.0), indexing='ij')
    lp = np.full((h, w, 3), 0.2)
    for _ in range(8):
        cu, cv, s = rng.rand(), rng.rand(), 0.05 + 0.15 * rng.rand()
        col = 0.3 + rng.rand(3)
        du = np.minimum(np.abs(uu - cu), 1 - np.abs(uu - cu))
        lp += np.exp(-(du ** 2 + (vv - cv) ** 2) / (2 * s * s))[..., None] * col
    return torch.from_numpy(lp.astype(np.float32))[None]


def tiny_scene(img_size=64, nf0=4, tex_size=32, tex_ch=16, nlat=16, nlon=32, seed=0, num_down=5):
    ps, pd = ray_pivots(6, 2, 5), ray_pivots(6, 2, 10)
    n_rays = ps.shape[1] + pd.shape[1]
    return {
        'mesh': scene.uv_sphere(nlat, nlon),
        'textures': synthetic_textures(tex_size, tex_ch, 4, seed),
        'unet_sd': unet_state_dict(3 * n_rays + 6 + tex_ch, 3 * n_rays, nf0, num_down, seed, out_channels_gcn=16),
        'pivots_spec': ps, 'pivots_diff': pd,
        'lp': synthetic_light_probe(20, 40, seed + 2),
    }
