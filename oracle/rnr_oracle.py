"""ORACLE (test infrastructure, not product): CPU restatement of the reference's per-view
rasterize -> shade -> U-Net -> ray-render forward pass, in torch-CPU float32 / numpy.

Every function cites the reference lines it follows.  What pins each function to the reference is
listed in DESIGN.md §Oracle; in short:
  * pinned by golden vectors generated from the reference's own Python (tests/golden/make_golden.py):
    projection, rasterizer_forward, interpolate_bilinear, texture_mapper, tbn_map, view_dir_map,
    ray_sampler (both modes), spherical mappings, unet_forward, ray_renderer, reconstruct/fit SH,
    the test_rnr.py frame assembly.
  * PARITY UNPINNED: sh_basis — the reference delegates to pyshtools==4.5 (sph_harm.py:66-68,
    environment.yml:143), which is neither vendored nor installable here and has no test vectors in
    the reference.  Restated from the published real-SH definition (orthonormal, no Condon-Shortley
    phase) and cross-checked against scipy.special and the quadrature orthonormality the reference
    itself relies on (sph_harm.py:80-86).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import raster as _raster


# ------------------------------------------------------------------------------------------------
# camera / projection
# ------------------------------------------------------------------------------------------------
def projection(vertices, K, R, t, dist_coeffs, orig_size, offset=None, scale=None, eps=1e-9):
    """neural_renderer/projection.py:6-53.  vertices [B,nv,3] world -> [B,nv,3] (u_ndc, v_ndc, z_cam)."""
    cam = torch.matmul(vertices, R.transpose(2, 1)) + t
    x, y, z = cam[..., 0], cam[..., 1], cam[..., 2]
    xn = x / (z + eps)
    yn = y / (z + eps)
    k1, k2, p1, p2, k3 = [dist_coeffs[:, None, i] for i in range(5)]
    r = torch.sqrt(xn ** 2 + yn ** 2)
    radial = 1 + k1 * (r ** 2) + k2 * (r ** 4) + k3 * (r ** 6)
    xd = xn * radial + 2 * p1 * xn * yn + p2 * (r ** 2 + 2 * xn ** 2)
    yd = yn * radial + p1 * (r ** 2 + 2 * yn ** 2) + 2 * p2 * xn * yn
    pix = torch.matmul(torch.stack([xd, yd, torch.ones_like(z)], dim=-1), K.transpose(1, 2))
    u, v = pix[..., 0], pix[..., 1]
    if offset is not None and scale is not None:
        u = (u + offset[:, None, 1]) * scale[:, None, 1]
        v = (v + offset[:, None, 0]) * scale[:, None, 0]
    v = orig_size - v
    u = 2 * (u - orig_size / 2.) / orig_size
    v = 2 * (v - orig_size / 2.) / orig_size
    return torch.stack([u, v, z], dim=-1)


def gather_faces(attrs, faces):
    """vertices_to_faces.py:4-46.  attrs [B,nv,A], faces [B or 1,nf,3] int -> [B,nf,3,A]."""
    if faces.shape[0] == 1 and attrs.shape[0] != 1:
        faces = faces.expand(attrs.shape[0], -1, -1)
    idx = faces.long()
    return torch.stack([attrs[b][idx[b]] for b in range(attrs.shape[0])])


def view_dir_map(img_hw, proj_inv, R_inv):
    """camera.py:5-32 -> (world-space [N,H,W,3], camera-space [N,H,W,3])."""
    H, W = int(img_hw[0]), int(img_hw[1])
    vv, uu = torch.meshgrid(torch.arange(H, dtype=torch.float32) + 0.5,
                            torch.arange(W, dtype=torch.float32) + 0.5, indexing='ij')
    pix = torch.stack([uu, vv, torch.ones_like(uu)], 0).reshape(3, -1)
    world, cam = [], []
    for i in range(proj_inv.shape[0]):
        d = F.normalize(-torch.matmul(proj_inv[i], pix), dim=0)
        cam.append(d.reshape(3, H, W).permute(1, 2, 0))
        world.append(torch.matmul(R_inv[i], d).reshape(3, H, W).permute(1, 2, 0))
    return F.normalize(torch.stack(world), dim=-1), torch.stack(cam)


# ------------------------------------------------------------------------------------------------
# rasterizer (python glue around the C restatement)
# ------------------------------------------------------------------------------------------------
def rasterize_rgbad(faces_v, image_size, near, far):
    """rasterize.py:255-340 with anti_aliasing=False and an all-zero texture (hot-path use,
    network.py:140-153): kernel maps + alpha + the vertical flip (rows reversed, rasterize.py:307-321)."""
    r = _raster.face_index_map(faces_v.detach().cpu().numpy(), image_size, near, far, return_depth=True)
    fim = torch.from_numpy(r['face_index_map'])
    out = {
        'face_index_map': fim.flip(1),
        'weight_map': torch.from_numpy(r['weight_map']).flip(1),
        'depth': torch.from_numpy(r['depth_map']).flip(1),
        'alpha': (fim >= 0).float().flip(1),
        'rgb': torch.zeros(fim.shape[0], 3, image_size, image_size),
        'faces_inv': torch.from_numpy(r['faces_inv']),
        'face_inv_map': torch.from_numpy(r['face_inv_map']).flip(1),
    }
    return out


def rasterizer_forward(mesh, proj, pose, img_size, dist_coeffs=None, offset=None, scale=None,
                       near=0.0, far=1e5, v_uvz_ndc=None):
    """network.Rasterizer.forward, network.py:156-216 (+ renderer.py:207-257).

    mesh: dict of torch tensors v [nv,3], vt [nvt,2], vn [nvn,3], f_v_idx/f_vt_idx/f_vn_idx [nf,3] int32
          (global_RT already applied, network.py:126-128).
    Returns the 14 outputs by name.
    """
    N = proj.shape[0]
    S = img_size
    if dist_coeffs is None:
        dist_coeffs = torch.zeros(1, 5)  # renderer.py:41-42
    verts = mesh['v'][None]
    R = pose[:, :3, :3]
    t = pose[:, :3, 3][:, None, :]
    # v_uvz_ndc: projected vertices computed elsewhere (e.g. by the kernel under test), so that the integer maps can be
    # compared exactly without this host's matmul rounding in between
    v_uvz = projection(verts.expand(N, -1, -1), proj, R, t, dist_coeffs, S, offset, scale) if v_uvz_ndc is None else v_uvz_ndc
    faces_v_uvz = gather_faces(v_uvz, mesh['f_v_idx'][None])
    ras = rasterize_rgbad(faces_v_uvz, S, near, far)
    fim, depth, alpha = ras['face_index_map'], ras['depth'], ras['alpha']
    fl = fim.long()  # -1 (background) wraps to the last face, as torch indexing does in the reference

    # vertices on the frontal surface (network.py:170-173; batch element 0 only)
    v_uvz = v_uvz.clone()
    v_uvz[..., 0] = (v_uvz[..., 0] * 0.5 + 0.5) * S
    v_uvz[..., 1] = (1 - (v_uvz[..., 1] * 0.5 + 0.5)) * S
    v_depth = interpolate_bilinear(depth[0, :, :, None], v_uvz[..., 0], v_uvz[..., 1])
    mesh_span = (mesh['v'].max(0)[0] - mesh['v'].min(0)[0]).max()
    v_front_mask = ((v_uvz[0, :, 2] - v_depth[0, :, 0]) < mesh_span * 5e-3)[None]

    # perspective-correct weights (network.py:176-180)
    z_inv = torch.stack([1 / faces_v_uvz[i, fl[i]][..., -1] for i in range(N)])
    depth = depth[..., None]
    w = ((z_inv * ras['weight_map']) * depth)[..., None]  # [N,S,S,3,1]

    def interp(attr, fidx):
        per_face = gather_faces(attr[None], fidx[None])[0]  # [nf,3,A]
        return (per_face[fl] * w).sum(-2), per_face

    uv_map, faces_vt = interp(mesh['vt'], mesh['f_vt_idx'])
    uv_map = uv_map - uv_map.floor()
    normal_map, _ = interp(mesh['vn'], mesh['f_vn_idx'])
    normal_map = F.normalize(normal_map, dim=-1)
    normal_cam = F.normalize(torch.einsum('nij,nhwj->nhwi', R, normal_map), dim=-1)
    position_map, faces_v = interp(mesh['v'], mesh['f_v_idx'])
    position_cam = torch.einsum('nij,nhwj->nhwi', R, position_map) + pose[:, :3, 3][:, None, None, :]
    return {
        'uv_map': uv_map, 'alpha': alpha, 'face_index_map': fim, 'weight_map': w,
        'faces_v_idx': mesh['f_v_idx'][None], 'normal_map': normal_map, 'normal_map_cam': normal_cam,
        'faces_v': faces_v[None], 'faces_vt': faces_vt[None], 'position_map': position_map,
        'position_map_cam': position_cam, 'depth': depth, 'v_uvz': v_uvz, 'v_front_mask': v_front_mask,
        'raw_weight_map': ras['weight_map'], 'faces_v_uvz': faces_v_uvz,
    }


# ------------------------------------------------------------------------------------------------
# samplers
# ------------------------------------------------------------------------------------------------
def bilinear_taps(H, W, x, y):
    """The integer tap indices and float weights of misc.interpolate_bilinear (misc.py:14-40).
    Returns (x0, y0, x1, y1) int64 fetch indices and (w00, w10, w01, w11)."""
    valid = ((x >= 0) & (x <= W - 1) & (y >= 0) & (y <= H - 1)).to(torch.float32)
    x0 = torch.floor(x).long()
    y0 = torch.floor(y).long()
    x1 = (x0 + 1).clamp(0, W - 1)
    y1 = (y0 + 1).clamp(0, H - 1)
    x0 = x0.clamp(0, W - 1)
    y0 = y0.clamp(0, H - 1)
    # weights use x0 - 1 where the clamp collapsed the pair (misc.py:32-35); fetches do not
    x0w = (x0 - (x0 == x1).long()).to(torch.float32)
    y0w = (y0 - (y0 == y1).long()).to(torch.float32)
    x1f, y1f = x1.to(torch.float32), y1.to(torch.float32)
    w00 = (x1f - x) * (y1f - y) * valid
    w10 = (x1f - x) * (y - y0w) * valid
    w01 = (x - x0w) * (y1f - y) * valid
    w11 = (x - x0w) * (y - y0w) * valid
    return (x0, y0, x1, y1), (w00, w10, w01, w11)


def interpolate_bilinear(data, x, y):
    """misc.py:5-42.  data [H,W,C]; x, y [...] -> [...,C]."""
    (x0, y0, x1, y1), (w00, w10, w01, w11) = bilinear_taps(data.shape[0], data.shape[1], x, y)
    return (data[y0, x0] * w00[..., None] + data[y1, x0] * w10[..., None] +
            data[y0, x1] * w01[..., None] + data[y1, x1] * w11[..., None])


def texture_mapper(textures, uv_map, sh_basis_map=None, sh_start_ch=3, apply_sh=True):
    """network.TextureMapper.forward, network.py:67-91.  textures: list of [1,S_i,S_i,C]; uv [N,H,W,2]
    -> [N,C,H,W] (sum over levels of bilinear fetches, then 9 channels x SH basis)."""
    out = None
    for tex in textures:
        s = tex.shape[1]
        xy = uv_map * (s - 1)
        lvl = interpolate_bilinear(tex[0], xy[..., 0], (s - 1) - xy[..., 1]).permute(0, 3, 1, 2)
        out = lvl if out is None else out + lvl
    if apply_sh and sh_basis_map is not None:
        out = out.clone()
        out[:, sh_start_ch:sh_start_ch + 9] = out[:, sh_start_ch:sh_start_ch + 9] * sh_basis_map.permute(0, 3, 1, 2)
    return out


# ------------------------------------------------------------------------------------------------
# tangent frames, rays
# ------------------------------------------------------------------------------------------------
def face_tangents(faces_v, faces_vt):
    """render.py:135-147: per-face tangent from UV deltas; det clamped at 1e-8 (negative dets too)."""
    e1 = faces_v[:, 1] - faces_v[:, 0]
    e2 = faces_v[:, 2] - faces_v[:, 0]
    d1 = faces_vt[:, 1] - faces_vt[:, 0]
    d2 = faces_vt[:, 2] - faces_vt[:, 0]
    f = 1.0 / (d1[:, 0] * d2[:, 1] - d2[:, 0] * d1[:, 1]).clamp(min=1e-8)
    tan = f[:, None] * (d2[:, 1:2] * e1 - d1[:, 1:2] * e2)
    return F.normalize(tan, dim=-1)


def tbn_map(normal_map, face_index_map, faces_v, faces_vt):
    """render.get_TBN_map, render.py:124-168 -> [N,H,W,3,3] with columns (T, B, N)."""
    tan = face_tangents(faces_v, faces_vt)[face_index_map.long()]
    n = F.normalize(normal_map, dim=-1)
    b = F.normalize(torch.cross(n, tan, dim=-1), dim=-1)
    t = F.normalize(torch.cross(b, n, dim=-1), dim=-1)
    return torch.stack([t, b, n], dim=4)


def euler_to_rot(theta):
    """data_util.py:175-191 (R = Rz Ry Rx), float64."""
    cx, sx = math.cos(theta[0]), math.sin(theta[0])
    cy, sy = math.cos(theta[1]), math.sin(theta[1])
    cz, sz = math.cos(theta[2]), math.sin(theta[2])
    rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return rz.dot(ry.dot(rx))


def ray_sampler_pivots(num_azi, num_polar, interval_polar):
    """network.RaySampler.__init__, network.py:418-443 -> (Rs [R,3,3] f32, pivots_dir [3,R] f32), R = 1 + azi*polar."""
    pol = np.arange(1, num_polar + 1) * interval_polar * np.pi / 180.0
    azi = np.arange(num_azi) * 2 * np.pi / num_azi
    pol, azi = np.meshgrid(pol, azi)
    pol, azi = pol.flatten(), azi.flatten()
    Rs = np.zeros((pol.shape[0] + 1, 3, 3), np.float32)
    Rs[0] = np.eye(3)
    for i in range(pol.shape[0]):
        Rs[i + 1] = euler_to_rot([0.0, pol[i], azi[i]])
    Rs = torch.from_numpy(Rs)
    pivots = torch.matmul(Rs, torch.tensor([0.0, 0.0, 1.0])[:, None])[..., 0].permute(1, 0)
    return Rs, pivots


def spherical_mapping(d, dim=0):
    """render.py:87-102: y-up equirect uv = (atan2(z,x)/2pi + 1/2, acos(y)/pi); `dim` is the xyz axis."""
    x, y, z = d.select(dim, 0), d.select(dim, 1), d.select(dim, 2)
    return torch.stack((torch.atan2(z, x) * 0.5 / np.pi + 0.5, torch.acos(y) * 1.0 / np.pi), dim=dim)


def spherical_mapping_inv(uv):
    """render.py:105-121.  uv [2,n] -> unit dirs [3,n], including the +-1 sign fixes at the seam."""
    y = torch.cos(uv[1] * np.pi)
    s = (1 - y ** 2).sqrt()
    a = uv[0] * 2 - 1
    x = s * torch.cos(a * np.pi)
    z = s * torch.sin(a * np.pi)
    z = z * ((~(a == 1.0)).to(s.dtype) * 2 - 1)
    z = z * ((~(a == -1.0)).to(s.dtype) * 2 - 1)
    return F.normalize(torch.stack((x, y, z), 0), dim=0)


def ray_sampler(mode, pivots, tbn, view_tangent, alpha):
    """network.RaySampler.forward, network.py:445-472.
    tbn [N,H,W,3,3], view_tangent [N,H,W,3], alpha [N,H,W,1] -> rays_dir [N,H,W,3,R], rays_uv [N,H,W,2,R],
    rays_dir_tangent."""
    if mode == 'reflect':
        v = view_tangent[..., None]                                   # [N,H,W,3,1]
        p = pivots                                                    # [3,R]
        refl = F.normalize((p * v).sum(-2, keepdim=True) * 2.0 * p - v, dim=-2)   # camera.py:35-45
        dirs_t = refl * alpha[..., None]
        dirs = torch.matmul(tbn, dirs_t)
    else:
        dirs_t = pivots
        dirs = torch.matmul(tbn, pivots)
    dirs = F.normalize(dirs, dim=-2)
    uv = spherical_mapping(dirs, dim=-2)
    uv = uv * alpha[..., None] - (alpha[..., None] == 0).to(dirs.dtype)
    return dirs, uv, dirs_t


# ------------------------------------------------------------------------------------------------
# spherical harmonics
# ------------------------------------------------------------------------------------------------
def sh_basis(lmax, directions):
    """sph_harm.evaluate_sh_basis, sph_harm.py:41-71 — PARITY UNPINNED (pyshtools absent).

    Real, orthonormal (integral of Y^2 over the sphere = 1) spherical harmonics WITHOUT the
    Condon-Shortley phase; azimuth = atan2(y, x), colatitude measured from +z (sph_harm.py:14-15,54-57);
    column order l = 0..lmax, m = -l..l with m < 0 <-> sin|m|phi, m >= 0 <-> cos m phi (sph_harm.py:64-69).
    float64 in, float64 out, like the reference's numpy path.
    """
    d = np.asarray(directions, np.float64)
    x, y, z = d[:, 0], d[:, 1], d[:, 2]
    phi = np.arctan2(y, x)
    ele = np.arctan2(z, np.sqrt(x * x + y * y))
    ct = np.cos(np.pi / 2.0 - ele)
    st = np.sqrt(np.maximum(0.0, 1.0 - ct * ct))
    n = d.shape[0]
    # associated Legendre P_l^m(ct) without the (-1)^m factor
    P = np.zeros((lmax + 1, lmax + 1, n))
    P[0, 0] = 1.0
    for m in range(1, lmax + 1):
        P[m, m] = P[m - 1, m - 1] * (2 * m - 1) * st
    for m in range(0, lmax):
        P[m + 1, m] = ct * (2 * m + 1) * P[m, m]
    for m in range(0, lmax + 1):
        for l in range(m + 2, lmax + 1):
            P[l, m] = ((2 * l - 1) * ct * P[l - 1, m] - (l + m - 1) * P[l - 2, m]) / (l - m)
    out = np.zeros((n, (lmax + 1) ** 2))
    col = 0
    for l in range(lmax + 1):
        for m in range(-l, l + 1):
            am = abs(m)
            norm = math.sqrt((2.0 - (1.0 if m == 0 else 0.0)) * (2 * l + 1) / (4.0 * math.pi) *
                             math.factorial(l - am) / math.factorial(l + am))
            ang = np.cos(am * phi) if m >= 0 else np.sin(am * phi)
            out[:, col] = norm * P[l, am] * ang
            col += 1
    return out


def fit_sh_coeff(samples, basis):
    """sph_harm.py:74-88: uniform quadrature weight 4pi/num_sample."""
    w = 4.0 * np.pi / samples.shape[-2]
    if samples.dim() == 2:
        return (samples[:, None, :] * basis[:, :, None]).sum(-3) * w
    return (samples[:, :, None, :] * basis[None, :, :, None]).sum(-3) * w


def reconstruct_sh(coeff, basis):
    """sph_harm.py:91-102."""
    if coeff.dim() == 2:
        return (basis[..., None] * coeff[None]).sum(-2)
    return (basis[None, :, :, None] * coeff[:, None]).sum(-2)


def lp_recon_dirs(h=100, w=200):
    """LightingSH.__init__, network.py:574-579: directions of the h x w equirect reconstruction grid."""
    vv, uu = torch.meshgrid(torch.arange(h, dtype=torch.float32) / (h - 1),
                            torch.arange(w, dtype=torch.float32) / (w - 1), indexing='ij')
    uv = torch.stack([uu, vv]).flatten(1)
    return spherical_mapping_inv(uv).permute(1, 0)


def reconstruct_lp(coeff, basis_recon, h=100, w=200):
    """LightingSH.reconstruct_lp, network.py:622-627.  coeff [nb,3] -> [h,w,3]."""
    return reconstruct_sh(coeff, basis_recon).reshape(h, w, -1)


def _area_tab(ssize, dsize):
    """OpenCV computeResizeAreaTab (imgproc/resize.cpp): list of (dst index, src index, weight) for one axis when
    shrinking (scale = ssize/dsize >= 1)."""
    scale = ssize / dsize
    tab = []
    for d in range(dsize):
        f1 = d * scale
        f2 = f1 + scale
        cell = min(scale, ssize - f1)
        s1 = int(math.ceil(f1))
        s2 = min(int(math.floor(f2)), ssize - 1)
        s1 = min(s1, s2)
        if s1 - f1 > 1e-3:
            tab.append((d, s1 - 1, np.float32((s1 - f1) / cell)))
        for sx in range(s1, s2):
            tab.append((d, sx, np.float32(1.0 / cell)))
        if f2 - s2 > 1e-3:
            tab.append((d, s2, np.float32(min(min(f2 - s2, 1.0), cell) / cell)))
    return tab


def resize_area(img, out_h, out_w):
    """cv2.resize(img, (out_w, out_h), interpolation=cv2.INTER_AREA) as LightingLP.__init__ calls it (network.py:667),
    restated from OpenCV's published algorithm — cv2 is absent here, so this restatement is PARITY-UNPINNED against
    OpenCV itself; tests pin its integer-ratio case against the plain box mean.  img [H,W,C] float32 numpy."""
    img = np.asarray(img, np.float32)
    H, W, C = img.shape
    if W >= out_w and H >= out_h:
        def mat(ssize, dsize):
            m = np.zeros((dsize, ssize), np.float32)
            for d, s_, a in _area_tab(ssize, dsize):
                m[d, s_] += a
            return m
        mx, my = mat(W, out_w), mat(H, out_h)
        tmp = np.einsum('dx,yxc->ydc', mx, img).astype(np.float32)          # rows first, like the kernel
        return np.einsum('ey,ydc->edc', my, tmp).astype(np.float32)

    def lin(ssize, dsize):
        scale = ssize / dsize
        d = np.arange(dsize)
        s0 = np.floor(d * scale).astype(np.int64)
        f = ((d + 1) - (s0 + 1) / scale).astype(np.float32)
        f = np.where(f <= 0, np.float32(0), f - np.floor(f)).astype(np.float32)
        s0 = np.minimum(s0, ssize - 1)
        return s0, np.minimum(s0 + 1, ssize - 1), f
    x0, x1, fx = lin(W, out_w)
    y0, y1, fy = lin(H, out_h)
    fx, fy = fx[None, :, None], fy[:, None, None]
    top = img[y0][:, x0] * (1 - fx) + img[y0][:, x1] * fx
    bot = img[y1][:, x0] * (1 - fx) + img[y1][:, x1] * fx
    return (top * (1 - fy) + bot * fy).astype(np.float32)


# ------------------------------------------------------------------------------------------------
# U-Net (RenderingNet)
# ------------------------------------------------------------------------------------------------
def _bn_batchstat(x, gamma, beta, eps=1e-5):
    """BatchNorm2d forced to train mode at inference (test_rnr.py:229-233): per-view mean / biased var."""
    mean = x.mean(dim=(2, 3), keepdim=True)
    var = x.var(dim=(2, 3), unbiased=False, keepdim=True)
    return (x - mean) / torch.sqrt(var + eps) * gamma[None, :, None, None] + beta[None, :, None, None]


def _conv_reflect(x, w, b=None, stride=1):
    return F.conv2d(F.pad(x, (1, 1, 1, 1), mode='reflect'), w, b, stride=stride)


def unet_forward(sd, x, num_down=5, prefix='net.'):
    """RenderingNet.forward (network.py:251-253) -> Unet.forward (pytorch_prototyping.py:532-536),
    live path only: the `if self.gcn:` branch of UnetSkipConnectionBlock.forward is overwritten by the
    `if self.flag_outer:` branch (pytorch_prototyping.py:407-419), so v_fea never reaches the output.
    sd: reference state-dict (keys of RenderingNet.state_dict()).  x [N,Cin,H,W].  Dropout2d = identity."""
    g = lambda k: sd[prefix + k]
    has = lambda k: (prefix + k) in sd

    def norm_act(y, key, act):
        if has(key + '.weight'):
            y = _bn_batchstat(y, g(key + '.weight'), g(key + '.bias'))
        return F.leaky_relu(y, 0.2) if act == 'lrelu' else F.relu(y)

    def block(y, path, depth):
        """UnetSkipConnectionBlock at nesting `depth` (0 = outermost); returns cat([x, up(sub(down(x)))])."""
        innermost = depth == num_down - 1
        d = path + 'down.net.'
        if innermost:   # DownBlock with norm=None: conv(+bias) at .1 and .5 (pytorch_prototyping.py:239-274)
            h = F.leaky_relu(_conv_reflect(y, g(d + '1.weight'), g(d + '1.bias')), 0.2)
            h = F.leaky_relu(_conv_reflect(h, g(d + '5.weight'), g(d + '5.bias'), stride=2), 0.2)
            u = path + 'up.net.'
            h = F.relu(F.conv_transpose2d(h, g(u + '0.weight'), g(u + '0.bias'), stride=2, padding=1))
            h = F.relu(_conv_reflect(h, g(u + '3.net.1.weight'), g(u + '3.net.1.bias')))
        else:
            h = norm_act(_conv_reflect(y, g(d + '1.weight')), d + '2', 'lrelu')
            h = norm_act(_conv_reflect(h, g(d + '6.weight'), stride=2), d + '7', 'lrelu')
            h = block(h, path + 'submodule.', depth + 1)
            u = path + 'up.net.'
            h = norm_act(F.conv_transpose2d(h, g(u + '0.weight'), None, stride=2, padding=1), u + '1', 'relu')
            h = norm_act(_conv_reflect(h, g(u + '4.net.1.weight')), u + '5', 'relu')
        return torch.cat([y, h], 1)

    outs = []
    for i in range(x.shape[0]):     # statistics stay per view (SURVEY.md §0 finding 2)
        h = _conv_reflect(x[i:i + 1], g('in_layer.0.net.1.weight'))
        h = norm_act(h, 'in_layer.1', 'lrelu')
        h = block(h, 'unet_block.', 0)
        h = _conv_reflect(h, g('out_layer.0.net.1.weight'), g('out_layer.0.net.1.bias'))
        outs.append(torch.tanh(h))
    return torch.cat(outs, 0)


# ------------------------------------------------------------------------------------------------
# ray renderer and the per-view frame
# ------------------------------------------------------------------------------------------------
def ray_renderer(albedo_specular, rays_uv, rays_lt, lp, albedo_diffuse=None, num_ray_diffuse=0,
                 seperate_albedo=False, lp_scale_factor=1):
    """network.RayRenderer.forward, network.py:481-527 (no_albedo=False).  lp [1 or N,H,W,C].
    Returns (out, out_specular, out_diffuse, ltt_specular, ltt_diffuse, rays_color)."""
    n_spec = rays_uv.shape[-1] - num_ray_diffuse
    lp = lp * lp_scale_factor
    Hl, Wl = lp.shape[1], lp.shape[2]
    sx = (rays_uv[..., 0, :] * float(Wl)).clamp(max=Wl - 1)
    sy = (rays_uv[..., 1, :] * float(Hl)).clamp(max=Hl - 1)
    if lp.shape[0] == 1:
        color = interpolate_bilinear(lp[0], sx, sy)
    else:
        color = torch.stack([interpolate_bilinear(lp[i], sx[i], sy[i]) for i in range(lp.shape[0])])
    color = color.permute(0, 3, 4, 1, 2)   # [N,R,C,H,W]
    lt_s = (rays_lt[:, :n_spec] * color[:, :n_spec]).sum(1) / n_spec
    out_s = albedo_specular * lt_s
    if num_ray_diffuse > 0:
        lt_d = (rays_lt[:, n_spec:] * color[:, n_spec:]).sum(1) / num_ray_diffuse
        out_d = (albedo_diffuse if seperate_albedo else albedo_specular) * lt_d
    else:
        lt_d = torch.zeros_like(lt_s)
        out_d = torch.zeros_like(out_s)
    return out_s + out_d, out_s, out_d, lt_s, lt_d, color


def shade_inputs(gb, proj_inv, R_inv, textures, pivots_spec, pivots_diff, sh_start_ch=6):
    """test_rnr.py:303-356: G-buffer -> (render_net_input [N,78+6+C,H,W], rays_uv [N,H,W,2,26], neural_img).
    gb: output dict of rasterizer_forward."""
    alpha = gb['alpha']
    N, H, W = alpha.shape
    tbn = tbn_map(gb['normal_map'], gb['face_index_map'], gb['faces_v'][0], gb['faces_vt'][0])
    vdir, _ = view_dir_map((H, W), proj_inv, R_inv)
    vtan = F.normalize(torch.einsum('nhwji,nhwj->nhwi', tbn, vdir), dim=-1)   # TBN^T v (test_rnr.py:314-315)
    sh = torch.from_numpy(sh_basis(2, vdir.reshape(-1, 3).numpy()).reshape(N, H, W, 9).astype(np.float32))
    neural = texture_mapper(textures, gb['uv_map'], sh, sh_start_ch=sh_start_ch)
    d_s, uv_s, _ = ray_sampler('reflect', pivots_spec, tbn, vtan, alpha[..., None])
    d_d, uv_d, _ = ray_sampler('diffuse', pivots_diff, tbn, vtan, alpha[..., None])
    dirs = torch.cat((d_s, d_d), -1)
    uvs = torch.cat((uv_s, uv_d), -1)
    net_in = torch.cat((dirs.permute(0, 4, 3, 1, 2).reshape(N, -1, H, W),     # ray-major, xyz inner
                        gb['normal_map'].permute(0, 3, 1, 2),
                        vdir.permute(0, 3, 1, 2),
                        neural), 1)
    return {'net_in': net_in, 'rays_uv': uvs, 'neural_img': neural, 'tbn': tbn, 'view_dir': vdir,
            'view_tangent': vtan, 'sh_basis_map': sh, 'rays_dir': dirs}


def render_frame(mesh, views, img_size, textures, unet_sd, lp, pivots_spec, pivots_diff,
                 num_down=5, sh_start_ch=6):
    """One pass of test_rnr.py:265-377 for a batch of views (each treated independently).
    views: dict proj, pose, proj_inv, R_inv (torch).  lp [1,Hl,Wl,3].  Returns dict incl. 'image' [N,3,S,S]."""
    gb = rasterizer_forward(mesh, views['proj'], views['pose'], img_size)
    sh_in = shade_inputs(gb, views['proj_inv'], views['R_inv'], textures, pivots_spec, pivots_diff, sh_start_ch)
    N, _, H, W = sh_in['net_in'].shape
    n_spec, n_diff = pivots_spec.shape[1], pivots_diff.shape[1]
    y = unet_forward(unet_sd, sh_in['net_in'], num_down)
    rays_lt = (y.reshape(N, n_spec + n_diff, -1, H, W) * 0.5 + 0.5) * 2.0      # test_rnr.py:357-359
    neural = sh_in['neural_img']
    img = ray_renderer(neural[:, 3:6], sh_in['rays_uv'], rays_lt, lp, albedo_diffuse=neural[:, :3],
                       num_ray_diffuse=n_diff, seperate_albedo=True)[0]
    out = dict(gb)
    out.update(sh_in)
    out.update({'unet_out': y, 'image': img})
    return out


def psnr(a, b, peak=None):
    """metric.py:7-16 style PSNR; peak defaults to the reference signal's max."""
    a = a.double()
    b = b.double()
    mse = ((a - b) ** 2).mean().item()
    if peak is None:
        peak = float(b.abs().max())
    if mse == 0:
        return float('inf')
    return 10.0 * math.log10(peak * peak / mse)
