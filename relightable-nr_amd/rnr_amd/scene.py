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
"""
import numpy as np


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
    f32 = lambda x: np.stack(x).astype(np.float32)
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
