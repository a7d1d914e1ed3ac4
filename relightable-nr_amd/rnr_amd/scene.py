"""Synthetic material_sphere-like scene: the reference ships no data
(/root/reference/.gitignore:2 `data/*`), so bench.py, smoke() and the tests use this generator.

Pure numpy; deterministic for a given argument set.

  * mesh: unit UV-sphere, written/consumed in the OBJ conventions of
    neural_renderer/load_obj.py:108-209 (`f v/vt/vn`, 1-based in the file, 0-based int32 in memory),
    vt = (phi/2pi, 1 - theta/pi) with a duplicated seam column, vn = v.
  * cameras: `spiral_step720` restated from camera.get_spiral (camera.py:72-75) and
    camera.RT_from_pos_lookat (camera.py:48-69), radius 3, looking at the origin,
    K = [[1.2 S, 0, S/2], [0, 1.2 S, S/2], [0, 0, 1]] (614.4 / 256 at S = 512).
  * per-view tensors as produced by dataio.ViewDataset.read_view (dataio.py:176-211):
    proj, pose, proj_inv, R_inv.
  * seeded random-init RenderingNet weights (reference key names / shapes), neural textures, light probe: the reference
    ships no checkpoints either.
"""
import math

import numpy as np
import torch

from .rays import ray_pivots


def uv_sphere(nlat=128, nlon=256, radius=1.0):
    """Returns dict v [nv,3] f32, vt [nv,2] f32, vn [nv,3] f32, f_v_idx/f_vt_idx/f_vn_idx [nf,3] i32.

    (nlat+1) x (nlon+1) lattice, two triangles per cell => nf = 2*nlat*nlon (65 536 at the defaults).
    The pole rows give zero-area triangles on purpose: the reference rasterizer's degenerate-face
    semantics (rasterize_cuda_kernel.cu:56-62) stay exercised by the headline scene.
    """
    theta = np.linspace(0.0, np.pi, nlat + 1, dtype=np.float64)          # colatitude from +y
    phi = np.linspace(0.0, 2.0 * np.pi, nlon + 1, dtype=np.float64)
    th, ph = np.meshgrid(theta, phi, indexing='ij')
    st = np.sin(th)
    st[0, :] = 0.0
    st[-1, :] = 0.0
    x = st * np.cos(ph)
    y = np.cos(th)
    z = st * np.sin(ph)
    vn = np.stack([x, y, z], -1).reshape(-1, 3)
    v = (vn * radius).astype(np.float32)
    vt = np.stack([ph / (2 * np.pi), 1.0 - th / np.pi], -1).reshape(-1, 2).astype(np.float32)
    i, j = np.meshgrid(np.arange(nlat), np.arange(nlon), indexing='ij')
    a = i * (nlon + 1) + j
    b = a + 1
    c = a + (nlon + 1)
    d = c + 1
    # counter-clockwise seen from outside (outward normal = v)
    f = np.stack([np.stack([a, b, c], -1), np.stack([b, d, c], -1)], 2).reshape(-1, 3).astype(np.int32)
    return {'v': v, 'vt': vt, 'vn': vn.astype(np.float32),
            'f_v_idx': f, 'f_vt_idx': f.copy(), 'f_vn_idx': f.copy()}


def write_obj(path, mesh):
    """Wavefront OBJ with `f v/vt/vn` triplets (the only form load_obj.py:168-175 accepts)."""
    with open(path, 'w') as fh:
        for p in mesh['v']:
            fh.write('v %.9g %.9g %.9g\n' % tuple(p))
        for p in mesh['vt']:
            fh.write('vt %.9g %.9g\n' % tuple(p))
        for p in mesh['vn']:
            fh.write('vn %.9g %.9g %.9g\n' % tuple(p))
        fv, ft, fn = mesh['f_v_idx'] + 1, mesh['f_vt_idx'] + 1, mesh['f_vn_idx'] + 1
        for k in range(fv.shape[0]):
            fh.write('f %d/%d/%d %d/%d/%d %d/%d/%d\n' % (
                fv[k, 0], ft[k, 0], fn[k, 0], fv[k, 1], ft[k, 1], fn[k, 1], fv[k, 2], ft[k, 2], fn[k, 2]))


def rt_from_pos_lookat(cam_pos, cam_lookat=(0.0, 0.0, 0.0), cam_up=(0.0, 1.0, 0.0)):
    """camera.py:48-69: rows of R = right, -up, forward; T = -R pos."""
    cam_pos = np.asarray(cam_pos, np.float64)
    fwd = np.asarray(cam_lookat, np.float64) - cam_pos
    fwd = fwd / np.linalg.norm(fwd)
    right = np.cross(fwd, np.asarray(cam_up, np.float64))
    right = right / np.linalg.norm(right)
    up = np.cross(right, fwd)
    R = np.stack([right, -up, fwd], 0)
    T = -R.dot(cam_pos[:, None])
    RT = np.vstack([np.hstack([R, T]), np.array([[0.0, 0.0, 0.0, 1.0]])])
    return RT


def spiral_angles(step_azi=-2.0, step_ele=90.0 / 720):
    """camera.py:72-75 (spiral_step720: 720 views, azimuth -2 deg/step, elevation 0.125 deg/step)."""
    num_step = int(np.floor(90.0 / step_ele))
    azi = np.arange(0, step_azi * num_step, step=step_azi)
    ele = np.arange(0, step_ele * num_step, step=step_ele)
    return azi, ele


def spiral_views(img_size=512, view_ids=None, radius=3.0, focal_scale=1.2, global_RT=None):
    """Per-view camera tensors, float32, shaped like ViewDataset.read_view stacked over views.

    Returns dict proj [V,3,3], pose [V,4,4], proj_inv [V,3,3], R_inv [V,3,3].
    pose = RT . global_RT^-1 (dataio.py:184-185); proj_inv = inv(proj), R_inv = R^T (dataio.py:200-201).
    """
    azi, ele = spiral_angles()
    if view_ids is None:
        view_ids = np.arange(azi.shape[0])
    view_ids = np.asarray(view_ids)
    S = float(img_size)
    K = np.array([[focal_scale * S, 0.0, S / 2], [0.0, focal_scale * S, S / 2], [0.0, 0.0, 1.0]])
    g_inv = np.eye(4) if global_RT is None else np.linalg.inv(np.asarray(global_RT, np.float64))
    proj, pose, proj_inv, r_inv = [], [], [], []
    for i in view_ids:
        a = np.deg2rad(azi[i])
        e = np.deg2rad(ele[i])
        pos = radius * np.array([np.cos(e) * np.sin(a), np.sin(e), np.cos(e) * np.cos(a)])
        RT = rt_from_pos_lookat(pos).dot(g_inv)
        proj.append(K)
        pose.append(RT)
        proj_inv.append(np.linalg.inv(K))
        r_inv.append(RT[:3, :3].T)
    # C-contiguous whatever the pieces are (R^T is a transposed view: stacked as it stands, every R_inv[i:i+1] a caller hands to a
    # pipeline costs a 9-float .contiguous() launch, 4 us of a 2.2 ms one-view frame)
    f32 = lambda x: np.ascontiguousarray(np.stack(x), dtype=np.float32)
    return {'proj': f32(proj), 'pose': f32(pose), 'proj_inv': f32(proj_inv), 'R_inv': f32(r_inv)}


def sphere_samples(n=4096):
    """Stand-in for sphere_samples_4096.mat ([n,3] float32 unit vectors, spiral starting at +z):
    a Fibonacci/spiral lattice — the same role (uniform quadrature nodes for sph_harm.fit_sh_coeff,
    sph_harm.py:80-86), not the same points."""
    k = np.arange(n, dtype=np.float64) + 0.5
    z = 1.0 - 2.0 * k / n
    r = np.sqrt(np.maximum(0.0, 1.0 - z * z))
    ph = k * (np.pi * (3.0 - np.sqrt(5.0)))
    return np.stack([r * np.cos(ph), r * np.sin(ph), z], -1).astype(np.float32)


def synthetic_sh_coeff(num_lighting=2, lmax=10, seed=1):
    """[L, (lmax+1)^2, 3]: l = 0 term 1.0, others N(0, 0.05)."""
    rng = np.random.RandomState(seed)
    c = rng.normal(0.0, 0.05, size=(num_lighting, (lmax + 1) ** 2, 3))
    c[:, 0, :] = 1.0
    return c.astype(np.float32)


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
        'mesh': uv_sphere(nlat, nlon),
        'textures': synthetic_textures(tex_size, tex_ch, 4, seed),
        'unet_sd': unet_state_dict(3 * n_rays + 6 + tex_ch, 3 * n_rays, nf0, num_down, seed, out_channels_gcn=16),
        'pivots_spec': ps, 'pivots_diff': pd,
        'lp': synthetic_light_probe(20, 40, seed + 2),
    }
