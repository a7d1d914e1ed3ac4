"""Operators of the hot path on torch device tensors, each a thin shim over one C-ABI entry point of
librnr_hip.so.  torch is plumbing here (HBM allocation, the current HIP stream); every operator launches
hand-written gfx950 kernels and raises if the library or a GPU is missing — there is no fallback.

Argument checks mirror the reference extension's CHECK_INPUT (rasterize_cuda.cpp:66-68): device-resident,
contiguous, right dtype, else RuntimeError.
"""
import ctypes
import weakref

import torch

from . import _lib
from ._lib import (RnrConvDesc, RnrConvSrc, RnrGbuffer, RnrMesh, RnrRays, check)


def _stream():
    """The current HIP stream of the CURRENT device.  Operators run inside `on_device(...)`, which makes the device of
    their tensor arguments current first — the C side never calls hipSetDevice, and a kernel launched into another
    device's stream faults (the reference pins everything to device 0, rasterize.py:50-69; we do not)."""
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _cuda_tensors(x, depth=0):
    if isinstance(x, torch.Tensor):
        if x.is_cuda:
            yield x
    elif isinstance(x, DeviceMesh):
        yield x.v
    elif depth < 2 and isinstance(x, dict):
        for v in x.values():
            yield from _cuda_tensors(v, depth + 1)
    elif depth < 2 and isinstance(x, (list, tuple)):
        for v in x:
            yield from _cuda_tensors(v, depth + 1)


class on_device:
    """Context: make `device` the current HIP device (no-op when it already is)."""

    def __init__(self, device):
        device = torch.device(device) if device is not None else None
        idx = None if device is None else (device.index if device.index is not None else torch.cuda.current_device())
        self._ctx = None if (idx is None or idx == torch.cuda.current_device()) else torch.cuda.device(idx)

    def __enter__(self):
        if self._ctx is not None:
            self._ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self._ctx is not None:
            return self._ctx.__exit__(*exc)
        return False


def _device_op(fn):
    """Run `fn` with the device of its tensor arguments current; raise if the arguments span devices."""
    import functools

    @functools.wraps(fn)
    def wrapper(*args, **kwargs):
        dev = None
        for a in args:
            for t in _cuda_tensors(a):
                if dev is None:
                    dev = t.device
                elif t.device != dev:
                    raise RuntimeError('%s: arguments live on different devices (%s and %s)' % (fn.__name__, dev, t.device))
        for a in kwargs.values():
            for t in _cuda_tensors(a):
                if dev is None:
                    dev = t.device
                elif t.device != dev:
                    raise RuntimeError('%s: arguments live on different devices (%s and %s)' % (fn.__name__, dev, t.device))
        if dev is None or dev.index == torch.cuda.current_device():
            return fn(*args, **kwargs)
        with torch.cuda.device(dev):
            return fn(*args, **kwargs)
    return wrapper


def _chk(t, name, dtype=torch.float32):
    if not isinstance(t, torch.Tensor):
        raise RuntimeError('%s must be a torch.Tensor' % name)
    if not t.is_cuda:
        raise RuntimeError('%s must be a CUDA (HIP) tensor' % name)
    if not t.is_contiguous():
        raise RuntimeError('%s must be contiguous' % name)
    if t.dtype != dtype:
        raise RuntimeError('%s must be %s, got %s' % (name, dtype, t.dtype))
    return t


_SIDE_STREAMS = {}


def side_streams(device, n):
    """The first `n` side HIP streams of `device`, created once per process and shared by every pipeline (view-group lanes,
    calls in flight).  HIP multiplexes streams onto a handful of hardware queues (4 by default): a process that keeps creating
    streams ends up with two of them on ONE queue, where their kernels run strictly one after the other — measured: the two
    slots of RNRPipeline(inflight=2) delivered exactly the sequential rate after other pipelines had created six streams."""
    device = torch.device(device)
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
    pool = _SIDE_STREAMS.setdefault(key, [])
    while len(pool) < n:
        pool.append(torch.cuda.Stream(device=device))
    return pool[:n]


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


# ---------------------------------------------------------------------------------------------------
# neural_renderer.cuda.rasterize drop-ins
# ---------------------------------------------------------------------------------------------------
@_device_op
def forward_face_index_map(faces, face_index_map, weight_map, depth_map, face_inv_map, faces_inv, image_size, near,
                           far, return_rgb, return_alpha, return_depth):
    """rasterize_cuda.cpp:66-98.  In-place on the caller's pre-filled buffers; returns them."""
    L = _lib.load()
    _chk(faces, 'faces'); _chk(face_index_map, 'face_index_map', torch.int32); _chk(weight_map, 'weight_map')
    _chk(depth_map, 'depth_map'); _chk(faces_inv, 'faces_inv')
    if return_depth:
        _chk(face_inv_map, 'face_inv_map')
    B, nf = faces.shape[0], faces.shape[1]
    ws = torch.empty(L.rnr_raster_workspace_bytes(B, nf, int(image_size)), dtype=torch.uint8, device=faces.device)
    check(L.rnr_forward_face_index_map(_ptr(faces), _ptr(face_index_map), _ptr(weight_map), _ptr(depth_map),
                                       _ptr(face_inv_map if return_depth else None), _ptr(faces_inv), B, nf,
                                       int(image_size), float(near), float(far), int(return_rgb), int(return_alpha),
                                       int(return_depth), _ptr(ws), _stream()))
    return [face_index_map, weight_map, depth_map, face_inv_map]


@_device_op
def forward_texture_sampling(faces, textures, face_index_map, weight_map, depth_map, rgb_map, sampling_index_map,
                             sampling_weight_map, image_size, eps):
    """rasterize_cuda.cpp:100-122."""
    L = _lib.load()
    _chk(faces, 'faces'); _chk(textures, 'textures'); _chk(face_index_map, 'face_index_map', torch.int32)
    _chk(weight_map, 'weight_map'); _chk(depth_map, 'depth_map'); _chk(rgb_map, 'rgb_map')
    _chk(sampling_index_map, 'sampling_index_map', torch.int32); _chk(sampling_weight_map, 'sampling_weight_map')
    B, nf = faces.shape[0], faces.shape[1]
    check(L.rnr_forward_texture_sampling(_ptr(faces), _ptr(textures), _ptr(face_index_map), _ptr(weight_map),
                                         _ptr(depth_map), _ptr(rgb_map), _ptr(sampling_index_map),
                                         _ptr(sampling_weight_map), B, nf, int(image_size), int(textures.shape[2]),
                                         float(eps), _stream()))
    return [rgb_map, sampling_index_map, sampling_weight_map]


@_device_op
def backward_pixel_map(faces, face_index_map, rgb_map, alpha_map, grad_rgb_map, grad_alpha_map, grad_faces, image_size,
                       eps, return_rgb, return_alpha):
    """rasterize_cuda.cpp:124-147.  grad_faces [B,nf,3,3] (pre-filled 0) is written in place and returned; the maps a
    flag switches off are placeholders and are not read (rasterize.py:118-131)."""
    L = _lib.load()
    _chk(faces, 'faces'); _chk(face_index_map, 'face_index_map', torch.int32); _chk(grad_faces, 'grad_faces')
    if return_rgb:
        _chk(rgb_map, 'rgb_map'); _chk(grad_rgb_map, 'grad_rgb_map')
    if return_alpha:
        _chk(alpha_map, 'alpha_map'); _chk(grad_alpha_map, 'grad_alpha_map')
    B, nf = faces.shape[0], faces.shape[1]
    check(L.rnr_backward_pixel_map(_ptr(faces), _ptr(face_index_map), _ptr(rgb_map if return_rgb else None),
                                   _ptr(alpha_map if return_alpha else None),
                                   _ptr(grad_rgb_map if return_rgb else None),
                                   _ptr(grad_alpha_map if return_alpha else None), _ptr(grad_faces), B, nf,
                                   int(image_size), float(eps), int(return_rgb), int(return_alpha), _stream()))
    return grad_faces


@_device_op
def backward_textures(face_index_map, sampling_weight_map, sampling_index_map, grad_rgb_map, grad_textures, num_faces):
    """rasterize_cuda.cpp:149-165.  grad_textures [B,nf,ts,ts,ts,3] (pre-filled 0) accumulated in place."""
    L = _lib.load()
    _chk(face_index_map, 'face_index_map', torch.int32); _chk(sampling_weight_map, 'sampling_weight_map')
    _chk(sampling_index_map, 'sampling_index_map', torch.int32); _chk(grad_rgb_map, 'grad_rgb_map')
    _chk(grad_textures, 'grad_textures')
    B, S = face_index_map.shape[0], face_index_map.shape[1]
    check(L.rnr_backward_textures(_ptr(face_index_map), _ptr(sampling_weight_map), _ptr(sampling_index_map),
                                  _ptr(grad_rgb_map), _ptr(grad_textures), B, int(num_faces), S,
                                  int(grad_textures.shape[2]), _stream()))
    return grad_textures


@_device_op
def backward_depth_map(faces, depth_map, face_index_map, face_inv_map, weight_map, grad_depth_map, grad_faces,
                       image_size):
    """rasterize_cuda.cpp:167-189.  Adds the depth-map gradient to grad_faces in place."""
    L = _lib.load()
    _chk(faces, 'faces'); _chk(depth_map, 'depth_map'); _chk(face_index_map, 'face_index_map', torch.int32)
    _chk(face_inv_map, 'face_inv_map'); _chk(weight_map, 'weight_map'); _chk(grad_depth_map, 'grad_depth_map')
    _chk(grad_faces, 'grad_faces')
    B, nf = faces.shape[0], faces.shape[1]
    check(L.rnr_backward_depth_map(_ptr(faces), _ptr(depth_map), _ptr(face_index_map), _ptr(face_inv_map),
                                   _ptr(weight_map), _ptr(grad_depth_map), _ptr(grad_faces), B, nf, int(image_size),
                                   _stream()))
    return grad_faces


@_device_op
def load_textures(image, faces, textures, is_update, texture_wrapping, use_bilinear):
    """load_textures_cuda.cpp:20-34.  `faces` [nf,3,2] uv are wrapped in place, `textures` [nf,ts,ts,ts,3] updated in
    place for the faces flagged in is_update [nf] int32; returns textures."""
    L = _lib.load()
    _chk(image, 'image'); _chk(faces, 'faces'); _chk(textures, 'textures'); _chk(is_update, 'is_update', torch.int32)
    check(L.rnr_load_textures(_ptr(image), _ptr(faces), _ptr(textures), _ptr(is_update), textures.shape[0],
                              textures.shape[1], image.shape[0], image.shape[1], int(texture_wrapping),
                              int(bool(use_bilinear)), _stream()))
    return textures


@_device_op
def create_texture_image(vertices_all, textures, image, eps):
    """create_texture_image_cuda.cpp:17-29.  Fills `image` [H,W,3] in place and returns it."""
    L = _lib.load()
    _chk(vertices_all, 'vertices_all'); _chk(textures, 'textures'); _chk(image, 'image')
    check(L.rnr_create_texture_image(_ptr(vertices_all), _ptr(textures), _ptr(image), textures.shape[0],
                                     textures.shape[1], image.shape[0], image.shape[1], float(eps), _stream()))
    return image


# ---------------------------------------------------------------------------------------------------
# fused path
# ---------------------------------------------------------------------------------------------------
@_device_op
def project_vertices(vertices, K, R, t, orig_size, dist_coeffs=None, offset=None, scale=None, eps=1e-9):
    """nr.projection for a shared mesh: vertices [nv,3], K/R [N,3,3], t [N,3] -> [N,nv,3]."""
    L = _lib.load()
    _chk(vertices, 'vertices'); _chk(K, 'K'); _chk(R, 'R'); _chk(t, 't')
    N, nv = K.shape[0], vertices.shape[0]
    out = torch.empty(N, nv, 3, dtype=torch.float32, device=vertices.device)
    for x, n in ((dist_coeffs, 'dist_coeffs'), (offset, 'offset'), (scale, 'scale')):
        if x is not None:
            _chk(x, n)
    check(L.rnr_project_vertices(_ptr(vertices), _ptr(K), _ptr(R), _ptr(t), _ptr(dist_coeffs), _ptr(offset),
                                 _ptr(scale), _ptr(out), N, nv, float(orig_size), float(eps), _stream()))
    return out


class DeviceMesh:
    """Mesh tensors resident in HBM + the ctypes struct the kernels take."""

    def __init__(self, v, vt, vn, f_v_idx, f_vt_idx, f_vn_idx, device):
        f = lambda x: torch.as_tensor(x, dtype=torch.float32).contiguous().to(device)
        i = lambda x: torch.as_tensor(x, dtype=torch.int32).contiguous().to(device)
        self.v, self.vt, self.vn = f(v), f(vt), f(vn)
        self.f_v_idx, self.f_vt_idx, self.f_vn_idx = i(f_v_idx), i(f_vt_idx), i(f_vn_idx)
        self.c = RnrMesh(self.v.data_ptr(), self.vt.data_ptr(), self.vn.data_ptr(), self.f_v_idx.data_ptr(),
                         self.f_vt_idx.data_ptr(), self.f_vn_idx.data_ptr(), self.v.shape[0], self.vt.shape[0],
                         self.vn.shape[0], self.f_v_idx.shape[0])
        self.num_faces = self.f_v_idx.shape[0]
        self.num_vertices = self.v.shape[0]
        self._tangents = None

    def tangents(self):
        if self._tangents is None:
            L = _lib.load()
            out = torch.empty(self.num_faces, 3, dtype=torch.float32, device=self.v.device)
            with on_device(self.v.device):
                check(L.rnr_face_tangents(ctypes.byref(self.c), _ptr(out), _stream()))
            self._tangents = out
        return self._tangents


GBUFFER_MAPS = {'face_index_map': (torch.int32, ()), 'alpha': (torch.float32, ()), 'depth': (torch.float32, ()),
                'weight_map': (torch.float32, (3,)), 'raw_weight_map': (torch.float32, (3,)),
                'uv_map': (torch.float32, (2,)), 'normal_map': (torch.float32, (3,)),
                'normal_map_cam': (torch.float32, (3,)), 'position_map': (torch.float32, (3,)),
                'position_map_cam': (torch.float32, (3,))}


@_device_op
def rasterize_gbuffer(mesh, v_uvz, pose, image_size, near=0.0, far=1e5, maps=None, out=None, workspace=None, prepared=False):
    """network.Rasterizer.forward's per-pixel maps in one pass.  Returns dict name -> tensor [N,S,S(,k)].
    prepared: `workspace` was cleared by frame_prepare on this stream for exactly this call (one launch fewer)."""
    L = _lib.load()
    _chk(v_uvz, 'v_uvz')
    N, S = v_uvz.shape[0], int(image_size)
    maps = list(GBUFFER_MAPS) if maps is None else list(maps)
    out = {} if out is None else out
    for m in maps:
        if m not in out:
            dt, tail = GBUFFER_MAPS[m]
            out[m] = torch.empty((N, S, S) + tail, dtype=dt, device=v_uvz.device)
    gb = RnrGbuffer(*[out[m].data_ptr() if m in out else None for m in GBUFFER_MAPS])
    if pose is not None:
        _chk(pose, 'pose')
    if prepared and workspace is None:
        raise ValueError('rasterize_gbuffer(prepared=True) needs the workspace frame_prepare cleared')
    if workspace is None:
        workspace = torch.empty(L.rnr_gbuffer_workspace_bytes(N, mesh.num_faces, S), dtype=torch.uint8, device=v_uvz.device)
    fn = L.rnr_rasterize_gbuffer_prepared if prepared else L.rnr_rasterize_gbuffer
    check(fn(ctypes.byref(mesh.c), _ptr(v_uvz), _ptr(pose), N, S, float(near), float(far), ctypes.byref(gb), _ptr(workspace),
             _stream()))
    return out


@_device_op
def frame_prepare(mesh, K, pose, image_size, v_uvz=None, tangents=None, lp_basis=None, lp_coeff=None, light_probe=None,
                  workspace=None, eps=1e-9):
    """The per-call preliminaries of a frame batch in one launch (rnr_frame_prepare): vertex projection straight from the
    [N,4,4] poses into v_uvz [N,nv,3], per-face tangents [nf,3], the light probe [ns,nc] = lp_basis [ns,nb] @ lp_coeff [nb,nc],
    and the clearing of a rasterize_gbuffer workspace.  Outputs are caller-allocated; None skips a part."""
    L = _lib.load()
    _chk(K, 'K'); _chk(pose, 'pose')
    N = K.shape[0]
    # project_vertex<true> strides the poses by 16 floats and reads t from elements 3, 7, 11: anything but [N,4,4] is out of bounds
    if K.dim() != 3 or tuple(K.shape[1:]) != (3, 3) or tuple(pose.shape) != (N, 4, 4):
        raise ValueError('frame_prepare: K must be [N,3,3] and pose [N,4,4] (got %s, %s)' % (tuple(K.shape), tuple(pose.shape)))
    if light_probe is not None and (lp_basis is None or lp_coeff is None):
        raise ValueError('frame_prepare: light_probe needs lp_basis and lp_coeff')
    for x, n in ((v_uvz, 'v_uvz'), (tangents, 'tangents'), (lp_basis, 'lp_basis'), (lp_coeff, 'lp_coeff'), (light_probe, 'light_probe')):
        if x is not None:
            _chk(x, n)
    if v_uvz is not None and tuple(v_uvz.shape) != (N, mesh.num_vertices, 3):
        raise ValueError('v_uvz must be [%d, %d, 3]' % (N, mesh.num_vertices))
    if tangents is not None and tuple(tangents.shape) != (mesh.num_faces, 3):
        raise ValueError('tangents must be [%d, 3]' % mesh.num_faces)
    ns = nb = nc = 0
    if light_probe is not None:
        ns, nb = lp_basis.shape
        nc = lp_coeff.shape[1]
        if lp_coeff.shape[0] != nb or light_probe.numel() != ns * nc:
            raise ValueError('light_probe must hold lp_basis [%d,%d] @ lp_coeff [%d,%d]' % (ns, nb, lp_coeff.shape[0], nc))
    if workspace is not None and workspace.numel() < L.rnr_gbuffer_workspace_bytes(N, mesh.num_faces, int(image_size)):
        raise ValueError('workspace smaller than rnr_gbuffer_workspace_bytes(%d views)' % N)
    check(L.rnr_frame_prepare(ctypes.byref(mesh.c), _ptr(K), _ptr(pose), N, int(image_size), float(eps), _ptr(v_uvz), _ptr(tangents),
                              _ptr(lp_basis), _ptr(lp_coeff), _ptr(light_probe), int(ns), int(nb), int(nc), _ptr(workspace), _stream()))


@_device_op
def shade_inputs(gb, mesh, proj_inv, R_inv, textures, pivots_spec, pivots_diff, sh_start_ch, c_pad=None,
                 want_rays_uv=False, want_neural_img=False, want_sh=False, net_in=None, tangents=None):
    """G-buffer -> channel-last RenderingNet input [N,H,W,c_pad] (+ optional API copies).
    textures: list of [1,S_l,S_l,C] or [S_l,S_l,C] device tensors; pivots_*: [3,R] CPU float tensors."""
    L = _lib.load()
    fim, alpha, uv, nrm = gb['face_index_map'], gb['alpha'], gb['uv_map'], gb['normal_map']
    N, H, W = fim.shape
    C = textures[0].shape[-1]
    ns, nd = int(pivots_spec.shape[1]), int(pivots_diff.shape[1])
    c_in = 3 * (ns + nd) + 6 + C
    if c_pad is None:
        c_pad = (c_in + 15) // 16 * 16
    dev = fim.device
    if net_in is None:
        net_in = torch.empty(N, H, W, c_pad, dtype=torch.float32, device=dev)
    rays_uv = torch.empty(N, H, W, 2, ns + nd, dtype=torch.float32, device=dev) if want_rays_uv else None
    neural = torch.empty(N, C, H, W, dtype=torch.float32, device=dev) if want_neural_img else None
    sh = torch.empty(N, H, W, 9, dtype=torch.float32, device=dev) if want_sh else None
    nl = len(textures)
    tex_ptrs = (ctypes.c_void_p * nl)(*[_chk(t, 'texture').data_ptr() for t in textures])
    tex_sizes = (ctypes.c_int * nl)(*[int(t.shape[-2]) for t in textures])
    ps = pivots_spec.detach().cpu().contiguous().float()
    pd = pivots_diff.detach().cpu().contiguous().float()
    rays = RnrRays(ps.data_ptr(), pd.data_ptr(), ns, nd)
    check(L.rnr_shade_inputs(_ptr(_chk(fim, 'face_index_map', torch.int32)), _ptr(_chk(alpha, 'alpha')),
                             _ptr(_chk(uv, 'uv_map')), _ptr(_chk(nrm, 'normal_map')), _ptr(mesh.tangents() if tangents is None else _chk(tangents, 'tangents')),
                             mesh.num_faces, _ptr(_chk(proj_inv, 'proj_inv')), _ptr(_chk(R_inv, 'R_inv')), tex_ptrs,
                             tex_sizes, nl, C, int(sh_start_ch), ctypes.byref(rays), _ptr(net_in), c_pad,
                             _ptr(rays_uv), _ptr(neural), _ptr(sh), N, H, W, _stream()))
    return {'net_in': net_in, 'rays_uv': rays_uv, 'neural_img': neural, 'sh_basis_map': sh, 'c_pad': c_pad}


@_device_op
def ray_render(unet_raw, bias, net_in, alpha, lp, num_spec, num_diff, albedo_diff_ch=0, albedo_spec_ch=3, image=None):
    """bias+tanh + rays_lt scaling + RayRenderer.forward (seperate_albedo=True) -> [N,3,H,W]."""
    L = _lib.load()
    N, H, W, cop = unet_raw.shape
    if image is None:
        image = torch.empty(N, 3, H, W, dtype=torch.float32, device=unet_raw.device)
    lp3 = lp.reshape(lp.shape[-3], lp.shape[-2], 3)
    check(L.rnr_ray_render(_ptr(_chk(unet_raw, 'unet_raw')), cop, _ptr(_chk(bias, 'bias')), _ptr(_chk(net_in, 'net_in')),
                           net_in.shape[-1], _ptr(_chk(alpha, 'alpha')), _ptr(_chk(lp3, 'lp')), lp3.shape[0],
                           lp3.shape[1], int(num_spec), int(num_diff), int(albedo_diff_ch), int(albedo_spec_ch),
                           _ptr(image), N, H, W, _stream()))
    return image


@_device_op
def ray_weights(net_in, alpha, lp, num_spec, num_diff, c_w, albedo_diff_ch=0, albedo_spec_ch=3, out=None):
    """The U-Net-independent half of the ray renderer (rnr_ray_weights): [N,H,W,c_w] weights W with
    frame[c] = sum_r (tanh(y[3r+c] + b) + 1) * W[3r+c]; consumed by UNetPlan.forward(..., ray=...) (rnr_conv2d_ray)."""
    L = _lib.load()
    N, H, W, cp = net_in.shape
    if out is None:
        out = torch.empty(N, H, W, int(c_w), dtype=torch.float32, device=net_in.device)
    lp3 = lp.reshape(lp.shape[-3], lp.shape[-2], 3)
    check(L.rnr_ray_weights(_ptr(_chk(net_in, 'net_in')), cp, _ptr(_chk(alpha, 'alpha')), _ptr(_chk(lp3, 'lp')), lp3.shape[0],
                            lp3.shape[1], int(num_spec), int(num_diff), int(albedo_diff_ch), int(albedo_spec_ch), _ptr(out),
                            int(c_w), N, H, W, _stream()))
    return out


def calibrate_mfma_f32(device='cuda:0', seconds=0.1, waves_per_simd=2):
    """TFLOP/s this device sustains NOW in a register-resident v_mfma_f32_32x32x2_f32 loop on every SIMD
    (rnr_calibrate_mfma_f32; nominal 157.3): a probe call sizes the loop for about `seconds`.  Blocks."""
    L = _lib.load()
    with on_device(device):
        scratch = torch.zeros(16, dtype=torch.float32, device=device)
        tf, dt = ctypes.c_double(0.0), ctypes.c_double(0.0)
        check(L.rnr_calibrate_mfma_f32(256, int(waves_per_simd), _ptr(scratch), ctypes.byref(tf), ctypes.byref(dt), _stream()))
        per_iter = max(dt.value, 1e-6) / 256.0
        iters = int(min(max(seconds / per_iter, 256), 4e6))
        check(L.rnr_calibrate_mfma_f32(iters, int(waves_per_simd), _ptr(scratch), ctypes.byref(tf), ctypes.byref(dt), _stream()))
    return {'tflops': tf.value, 'seconds': dt.value, 'iters': iters, 'waves_per_simd': int(waves_per_simd)}


@_device_op
def sh_basis(dirs, lmax):
    """sph_harm.evaluate_sh_basis: dirs [n,3] -> [n,(lmax+1)^2] float32."""
    L = _lib.load()
    _chk(dirs, 'dirs')
    out = torch.empty(dirs.shape[0], (lmax + 1) ** 2, dtype=torch.float32, device=dirs.device)
    check(L.rnr_sh_basis(_ptr(dirs), _ptr(out), dirs.shape[0], int(lmax), _stream()))
    return out


@_device_op
def sh_reconstruct(basis, coeff):
    """sph_harm.reconstruct_sh for coeff [nb,C]: -> [ns,C]."""
    L = _lib.load()
    _chk(basis, 'basis'); _chk(coeff, 'coeff')
    out = torch.empty(basis.shape[0], coeff.shape[1], dtype=torch.float32, device=basis.device)
    check(L.rnr_sh_reconstruct(_ptr(basis), _ptr(coeff), _ptr(out), basis.shape[0], basis.shape[1], coeff.shape[1],
                               _stream()))
    return out


@_device_op
def sh_fit(samples, basis):
    """sph_harm.fit_sh_coeff for samples [ns,C]: -> [nb,C]."""
    L = _lib.load()
    _chk(samples, 'samples'); _chk(basis, 'basis')
    out = torch.empty(basis.shape[1], samples.shape[1], dtype=torch.float32, device=basis.device)
    check(L.rnr_sh_fit(_ptr(samples), _ptr(basis), _ptr(out), samples.shape[0], basis.shape[1], samples.shape[1],
                       _stream()))
    return out


@_device_op
def interpolate_bilinear(data, x, y, want_taps=False):
    """misc.interpolate_bilinear: data [H,W,C], x/y [...] -> [...,C] (+ int32 taps [...,4])."""
    L = _lib.load()
    _chk(data, 'data')
    shp = x.shape
    xf, yf = x.reshape(-1).contiguous(), y.reshape(-1).contiguous()
    _chk(xf, 'x'); _chk(yf, 'y')
    n, c = xf.shape[0], data.shape[2]
    out = torch.empty(n, c, dtype=torch.float32, device=data.device)
    taps = torch.empty(n, 4, dtype=torch.int32, device=data.device) if want_taps else None
    check(L.rnr_interpolate_bilinear(_ptr(data), data.shape[0], data.shape[1], c, _ptr(xf), _ptr(yf), _ptr(out),
                                     _ptr(taps), n, _stream()))
    out = out.reshape(*shp, c)
    return (out, taps.reshape(*shp, 4)) if want_taps else out


@_device_op
def resize_area(img, out_h, out_w):
    """cv2.resize(img, (out_w, out_h), interpolation=cv2.INTER_AREA) for a channel-last float image [H,W,C] -> [out_h,out_w,C]
    (network.py:667; restated from OpenCV's algorithm, see include/rnr_hip.h)."""
    L = _lib.load()
    _chk(img, 'img')
    if img.dim() != 3:
        raise RuntimeError('img must be [H,W,C]')
    out = torch.empty(int(out_h), int(out_w), img.shape[2], dtype=torch.float32, device=img.device)
    check(L.rnr_resize_area(_ptr(img), _ptr(out), img.shape[0], img.shape[1], int(out_h), int(out_w), img.shape[2], _stream()))
    return out


@_device_op
def nchw_to_nhwc(x, c_pad):
    L = _lib.load()
    _chk(x, 'x')
    n, c, h, w = x.shape
    out = torch.empty(n, h, w, c_pad, dtype=torch.float32, device=x.device)
    check(L.rnr_nchw_to_nhwc(_ptr(x), _ptr(out), n, c, h, w, c_pad, _stream()))
    return out


@_device_op
def nhwc_to_nchw(x, c, bias=None, apply_tanh=False):
    L = _lib.load()
    _chk(x, 'x')
    n, h, w, c_pad = x.shape
    out = torch.empty(n, c, h, w, dtype=torch.float32, device=x.device)
    check(L.rnr_nhwc_to_nchw(_ptr(x), _ptr(out), _ptr(bias), int(apply_tanh), n, c, h, w, c_pad, _stream()))
    return out


# ---------------------------------------------------------------------------------------------------
# stand-alone operators behind the drop-in Python API
# ---------------------------------------------------------------------------------------------------
@_device_op
def view_dir_map(img_hw, proj_inv, R_inv):
    """camera.get_view_dir_map -> (world [N,H,W,3], cam [N,H,W,3])."""
    L = _lib.load()
    _chk(proj_inv, 'proj_inv'); _chk(R_inv, 'R_inv')
    N, H, W = proj_inv.shape[0], int(img_hw[0]), int(img_hw[1])
    world = torch.empty(N, H, W, 3, dtype=torch.float32, device=proj_inv.device)
    cam = torch.empty_like(world)
    check(L.rnr_view_dir_map(_ptr(proj_inv), _ptr(R_inv), _ptr(world), _ptr(cam), N, H, W, _stream()))
    return world, cam


@_device_op
def face_tangents(faces_v, faces_vt):
    """Per-face unit tangents from gathered per-face positions [nf,3,3] and texcoords [nf,3,2] (render.py:135-150)."""
    L = _lib.load()
    nf = faces_v.shape[0]
    dev = faces_v.device
    v = faces_v.reshape(nf * 3, 3).float().contiguous()
    vt = faces_vt.reshape(nf * 3, 2).float().contiguous()
    idx = torch.arange(nf * 3, dtype=torch.int32, device=dev).reshape(nf, 3).contiguous()
    mesh = RnrMesh(v.data_ptr(), vt.data_ptr(), None, idx.data_ptr(), idx.data_ptr(), None, nf * 3, nf * 3, 0, nf)
    out = torch.empty(nf, 3, dtype=torch.float32, device=dev)
    check(L.rnr_face_tangents(ctypes.byref(mesh), _ptr(out), _stream()))
    return out


@_device_op
def tbn_map(normal_map, face_index_map, tangents):
    """render.get_TBN_map body given unit per-face tangents -> [N,H,W,3,3]."""
    L = _lib.load()
    _chk(normal_map, 'normal_map'); _chk(face_index_map, 'face_index_map', torch.int32); _chk(tangents, 'tangents')
    N, H, W = face_index_map.shape
    out = torch.empty(N, H, W, 3, 3, dtype=torch.float32, device=normal_map.device)
    check(L.rnr_tbn_map(_ptr(normal_map), _ptr(face_index_map), _ptr(tangents), tangents.shape[0], _ptr(out), N, H, W,
                        _stream()))
    return out


@_device_op
def tbn_matvec(tbn, vec, transposed=True):
    """out[p] = tbn[p]^T vec[p] (transposed) or tbn[p] vec[p]: tbn [P,3,3] contiguous, vec [P,3] contiguous -> [P,3]
    (test_rnr.py:314's batched product in one launch)."""
    L = _lib.load()
    _chk(tbn, 'tbn'); _chk(vec, 'vec')
    P = tbn.shape[0]
    out = torch.empty(P, 3, dtype=torch.float32, device=tbn.device)
    check(L.rnr_tbn_matvec(_ptr(tbn), _ptr(vec), _ptr(out), P, 1 if transposed else 0, _stream()))
    return out


_HOST_COPIES = {}


def _host_copy(t):
    """float32 CPU copy of a small constant tensor (ray pivots): `.cpu()` on a device buffer is a blocking copy — one stream
    drain per call in the reference's per-view loop (two per view for the two ray samplers).  The copy is cached per tensor
    OBJECT: the entry holds a weak reference to the tensor and is dropped when the tensor dies, and a hit is only accepted
    when it is the same live object at the same address, version, shape and dtype (an address recycled by the caching
    allocator for another tensor, a `.data =` / `set_()` swap or an in-place write all miss)."""
    if not t.is_cuda:
        return t.detach().contiguous().float()
    k = id(t)
    sig = (t.data_ptr(), t._version, tuple(t.shape), t.dtype, t.device.index)
    hit = _HOST_COPIES.get(k)
    if hit is not None and hit[0]() is t and hit[1] == sig:
        return hit[2]
    host = t.detach().cpu().contiguous().float()
    _HOST_COPIES[k] = (weakref.ref(t, lambda _r, k=k: _HOST_COPIES.pop(k, None)), sig, host)
    return host


@_device_op
def ray_sampler(reflect, pivots, tbn, view_tangent, alpha):
    """network.RaySampler.forward.  tbn [...,3,3], view_tangent [...,3], alpha [...,1] -> dirs [...,3,R], uv [...,2,R],
    dirs_tangent [...,3,R] (reflect) or the pivots (diffuse)."""
    L = _lib.load()
    lead = tbn.shape[:-2]
    npix = 1
    for d in lead:
        npix *= int(d)
    R = int(pivots.shape[1])
    dev = tbn.device
    tb = _chk(tbn.reshape(npix, 3, 3).contiguous(), 'tbn')
    al = _chk(alpha.reshape(npix).contiguous(), 'alpha')
    vt = _chk(view_tangent.reshape(npix, 3).contiguous(), 'view_tangent') if reflect else None
    piv = _host_copy(pivots)
    dirs = torch.empty(lead + (3, R), dtype=torch.float32, device=dev)
    uv = torch.empty(lead + (2, R), dtype=torch.float32, device=dev)
    dt = torch.empty(lead + (3, R), dtype=torch.float32, device=dev) if reflect else None
    check(L.rnr_ray_sampler(int(bool(reflect)), piv.data_ptr(), R, _ptr(tb), _ptr(vt), _ptr(al), _ptr(dirs), _ptr(uv),
                            _ptr(dt), npix, _stream()))
    return dirs, uv, dt


@_device_op
def texture_mapper(textures, uv_map, sh_basis_map=None, sh_start_ch=3):
    """network.TextureMapper.forward for any channel count -> [N,C,H,W]."""
    L = _lib.load()
    _chk(uv_map, 'uv_map')
    N, H, W = uv_map.shape[:3]
    tex = [_chk(t.reshape(t.shape[-3], t.shape[-2], t.shape[-1]).contiguous(), 'texture') for t in textures]
    C = tex[0].shape[-1]
    nl = len(tex)
    ptrs = (ctypes.c_void_p * nl)(*[t.data_ptr() for t in tex])
    sizes = (ctypes.c_int * nl)(*[int(t.shape[0]) for t in tex])
    out = torch.empty(N, C, H, W, dtype=torch.float32, device=uv_map.device)
    if sh_basis_map is not None:
        _chk(sh_basis_map, 'sh_basis_map')
    check(L.rnr_texture_mapper(_ptr(uv_map), _ptr(sh_basis_map), ptrs, sizes, nl, C, int(sh_start_ch), _ptr(out), N, H, W,
                               _stream()))
    return out


@_device_op
def ray_renderer(rays_uv, rays_lt, lp, albedo_specular, albedo_diffuse=None, num_ray_diffuse=0, no_albedo=False,
                 seperate_albedo=False, lp_scale_factor=1.0):
    """network.RayRenderer.forward on API-shaped tensors -> (out, out_spec, out_diff, ltt_spec, ltt_diff, rays_color)."""
    L = _lib.load()
    _chk(rays_uv, 'rays_uv'); _chk(rays_lt, 'rays_lt'); _chk(lp, 'lp'); _chk(albedo_specular, 'albedo_specular')
    N, R, C, H, W = rays_lt.shape
    if albedo_diffuse is not None:
        _chk(albedo_diffuse, 'albedo_diffuse')
    dev = rays_lt.device
    mk = lambda: torch.empty(N, C, H, W, dtype=torch.float32, device=dev)
    out, o_s, o_d, l_s, l_d = mk(), mk(), mk(), mk(), mk()
    color = torch.empty(N, R, C, H, W, dtype=torch.float32, device=dev)
    check(L.rnr_ray_renderer(_ptr(rays_uv), _ptr(rays_lt), _ptr(lp), lp.shape[0], lp.shape[1], lp.shape[2],
                             _ptr(albedo_specular), _ptr(albedo_diffuse), C, R, int(num_ray_diffuse), int(bool(no_albedo)),
                             int(bool(seperate_albedo)), float(lp_scale_factor), _ptr(out), _ptr(o_s), _ptr(o_d), _ptr(l_s),
                             _ptr(l_d), _ptr(color), N, H, W, _stream()))
    return out, o_s, o_d, l_s, l_d, color
