"""Drop-in `network` (reference: network.py): the nn.Module classes test_rnr.py / test_dnr.py construct, with the same
constructor / forward signatures, buffers and state-dict keys, computing on librnr_hip.so.

In scope: TextureMapper, Rasterizer, RenderingNet, Interpolater, RaySampler, RayRenderer, LightingSH.
In scope as "next" rows: LightingLP (HIP area resize instead of cv2), Mesh (host bookkeeping).
Out of scope (raise on construction): DenseDeepGCN (train-time only; its cached output `v_feature` is loaded from the
checkpoint and is dead at the output anyway), InterpolaterVertexAttr, RaysLTChromLoss (loss).
With these, test_dnr.py and test_rnr.py construct every module they need, provided the inference flags keep
DenseDeepGCN unused (test_rnr.py only builds it when the checkpoint lacks a cached `v_feature`).

Intentional deviations from the reference classes (all invisible to the scripts' outputs):
  * RenderingNet / Unet: the dead GCN pass is not executed (no second running-stat update from it); BatchNorm in train
    mode takes whole-batch statistics like torch (the fused RNRPipeline uses per-view statistics = N calls with N = 1).
  * LightingSH.reconstruct_lp detaches `coeff` (inference build: no gradient flows through the HIP kernels) and returns
    [..., lp_recon_h, lp_recon_w, C]; a 2-D `init_coeff` is expanded to [num_lighting, nb, C] (the reference assigns
    the 2-D tensor and breaks its own indexing); `l_samples` is computed from the coefficients at init.
  * render.get_TBN_map does not raise on NaN by default (the reference's check costs three host syncs per view).
"""
import numpy as np
import torch
import torch.nn as nn

import neural_renderer as nr
import camera
import misc
import render
import sph_harm
from pytorch_prototyping.pytorch_prototyping import *  # noqa: F401,F403  (the reference does the same, network.py:10)
from pytorch_prototyping.pytorch_prototyping import Unet
from rnr_amd import ops
from rnr_amd.rays import ray_pivots as _ray_pivots


class TextureMapper(nn.Module):
    """network.py:20-99."""

    def __init__(self, texture_size, texture_num_ch, mipmap_level, texture_init=None, fix_texture=False, apply_sh=False):
        super().__init__()
        self.register_buffer('texture_size', torch.tensor(texture_size))
        self.register_buffer('texture_num_ch', torch.tensor(texture_num_ch))
        self.register_buffer('mipmap_level', torch.tensor(mipmap_level))
        self.register_buffer('apply_sh', torch.tensor(apply_sh))
        self.textures = nn.ParameterList([])
        self.textures_size = []
        for lvl in range(int(mipmap_level)):
            s = int(np.round(int(texture_size) / (2.0 ** lvl)))
            t = torch.ones(1, s, s, int(texture_num_ch), dtype=torch.float32) * (1.0 if lvl == 0 else 0.01)
            if texture_init is not None and lvl == 0:
                k = texture_init.shape[-1]
                t[..., :k] = texture_init[None]
                t[..., k:2 * k] = texture_init[None]
            self.textures_size.append(s)
            self.textures.append(nn.Parameter(t))
        self.register_buffer('tex_flatten_mipmap_init', torch.nn.functional.relu(self.flatten_mipmap(0, 6)))
        if fix_texture:
            for p in self.textures:
                p.requires_grad = False

    def forward(self, uv_map, sh_basis_map=None, sh_start_ch=3):
        """uv_map [N,H,W,2], sh_basis_map [N,H,W,9] -> [N,C,H,W] (network.py:67-91)."""
        sh = sh_basis_map.float().contiguous() if (sh_basis_map is not None and self._apply_sh_flag()) else None
        return ops.texture_mapper([p.detach() for p in self.textures], uv_map.float().contiguous(), sh, sh_start_ch)

    def _apply_sh_flag(self):
        """bool(self.apply_sh) without a device -> host read per call: the buffer lives on the GPU after `.to(device)` and the
        reference's `if self.apply_sh` (network.py:84) drains the stream once per view.  Cached per (storage, version): an in-place
        change or a load_state_dict is seen."""
        t = self.apply_sh
        key = (id(t), t.data_ptr(), t._version)
        if getattr(self, '_apply_sh_key', None) != key:
            self._apply_sh_key, self._apply_sh_val = key, bool(t)
        return self._apply_sh_val

    def _apply(self, fn, *a, **k):
        self._apply_sh_key = None
        return super()._apply(fn, *a, **k)

    def _load_from_state_dict(self, *a, **k):
        self._apply_sh_key = None
        return super()._load_from_state_dict(*a, **k)

    def flatten_mipmap(self, start_ch, end_ch):
        """network.py:93-99 (init-time / visualisation helper)."""
        out = None
        for lvl, p in enumerate(self.textures):
            t = p[..., start_ch:end_ch]
            if lvl > 0:
                t = torch.nn.functional.interpolate(t.permute(0, 3, 1, 2), size=(self.textures_size[0],) * 2,
                                                    mode='bilinear').permute(0, 2, 3, 1)
            out = t if out is None else out + t
        return out


class Rasterizer(nn.Module):
    """network.py:102-216: same buffers and the same 14-tuple; one fused projection + raster + interpolation pass."""

    def __init__(self, obj_fp, img_size, global_RT=None):
        super().__init__()
        v_attr, f_attr = nr.load_obj(obj_fp, normalization=False, use_cuda=False)
        vertices, vn, vt = v_attr['v'], v_attr['vn'], v_attr['vt']
        self.num_vertex, self.num_face = vertices.shape[0], f_attr['f_v_idx'].shape[0]
        self.img_size = img_size
        if global_RT is not None:   # network.py:126-128
            g = global_RT.to(vertices.device).float()
            vertices = torch.matmul(g, torch.cat((vertices, torch.ones(self.num_vertex, 1)), dim=1).t()).t()[:, :3]
            vn = torch.nn.functional.normalize(torch.matmul(g[:3, :3], vn.t()).t(), dim=1)
        self.register_buffer('vertices', vertices[None].contiguous())
        self.register_buffer('faces', f_attr['f_v_idx'][None].contiguous())
        self.register_buffer('vertices_texcoords', vt[None].contiguous())
        self.register_buffer('faces_vt_idx', f_attr['f_vt_idx'][None].contiguous())
        self.register_buffer('vertices_normals', vn[None].contiguous())
        self.register_buffer('faces_vn_idx', f_attr['f_vn_idx'][None].contiguous())
        self.mesh_span = (self.vertices[0].max(dim=0)[0] - self.vertices[0].min(dim=0)[0]).max()
        self.textures = nn.Parameter(torch.zeros(1, self.num_face, 4, 4, 4, 3, dtype=torch.float32))  # API only (network.py:140-142)
        renderer = nr.Renderer(image_size=img_size, camera_mode='projection', orig_size=img_size, near=0.0, far=1e5)
        renderer.light_intensity_directional = 0.0
        renderer.light_intensity_ambient = 1.0
        renderer.anti_aliasing = False
        renderer.fill_back = False
        self.renderer = renderer
        self._mesh = {}
        self._static = {}

    def _apply(self, fn, *a, **k):
        self._mesh = {}
        self._static = {}
        return super()._apply(fn, *a, **k)

    def _load_from_state_dict(self, *a, **k):
        self._mesh = {}
        self._static = {}
        return super()._load_from_state_dict(*a, **k)

    def _static_outputs(self, fill_back, device):
        """The entries of the 14-tuple that do not depend on the camera (faces_v_idx, faces_v, faces_vt: gathers of the mesh
        buffers the reference redoes per call, network.py:183-198) and mesh_span on the device (a per-call `.to(device)` of a
        CPU scalar is a pageable copy that drains the stream): computed once per (fill_back, device), reset by .to() / .cuda() /
        load_state_dict.  The returned tensors are SHARED between calls and must be treated as read-only (the reference returns
        fresh gathers; a caller that edits them in place must clone first)."""
        # in-place edits of the mesh buffers (and of mesh_span) bump their version counters: a stale gather is never returned
        key = (fill_back, str(device)) + tuple((t.data_ptr(), t._version) for t in
                                                (self.vertices, self.faces, self.vertices_texcoords, self.faces_vt_idx,
                                                 self.mesh_span))
        if key not in self._static:
            self._static.clear()
            faces_v_idx = self._both_sides(self.faces) if fill_back else self.faces
            faces_v = nr.vertex_attrs_to_faces(self.vertices, faces_v_idx)
            faces_vt = nr.vertex_attrs_to_faces(self.vertices_texcoords,
                                                self._both_sides(self.faces_vt_idx) if fill_back else self.faces_vt_idx)
            self._static[key] = (faces_v_idx, faces_v, faces_vt, self.mesh_span.to(device) * 5e-3)
        return self._static[key]

    @staticmethod
    def _both_sides(idx):
        """renderer.py:209-211 / network.py:183-185: every face once more with reversed vertex order."""
        return torch.cat((idx, idx[:, :, [2, 1, 0]]), dim=1).contiguous()

    def _device_mesh(self, fill_back=False):
        if fill_back not in self._mesh:
            f = self._both_sides if fill_back else (lambda x: x)
            self._mesh[fill_back] = ops.DeviceMesh(self.vertices[0], self.vertices_texcoords[0], self.vertices_normals[0],
                                                   f(self.faces)[0], f(self.faces_vt_idx)[0], f(self.faces_vn_idx)[0],
                                                   self.vertices.device)
        return self._mesh[fill_back]

    def forward(self, proj, pose, dist_coeffs, offset, scale):
        if self.renderer.anti_aliasing:
            # the reference's own forward cannot run with it either: nr.Renderer.render returns face_index_map / weight_map
            # at 2x resolution next to a pooled depth (rasterize.py:296-330), and network.py:178 multiplies the two
            raise NotImplementedError('Rasterizer: anti_aliasing=True is not usable through network.Rasterizer.forward '
                                      '(shape mismatch at network.py:178 in the reference itself); use nr.Renderer directly')
        fill_back = bool(self.renderer.fill_back)
        mesh = self._device_mesh(fill_back)
        S = self.img_size
        N = proj.shape[0]
        R = pose[:, :3, :3].float().contiguous()
        t = pose[:, :3, 3].float().contiguous()
        f = lambda x: x.float().contiguous() if x is not None else None
        v_ndc = ops.project_vertices(mesh.v, proj.float().contiguous(), R, t, S, f(dist_coeffs),
                                     f(offset) if scale is not None else None, f(scale) if offset is not None else None)
        gb = ops.rasterize_gbuffer(mesh, v_ndc, pose.float().contiguous(), S, self.renderer.near, self.renderer.far)
        depth = gb['depth']
        # network.py:170-173: vertices on the frontal surface (batch element 0 only), pixel coordinates
        v_uvz = v_ndc.clone()
        v_uvz[..., 0] = (v_uvz[..., 0] * 0.5 + 0.5) * S
        v_uvz[..., 1] = (1 - (v_uvz[..., 1] * 0.5 + 0.5)) * S
        v_depth = misc.interpolate_bilinear(depth[0, :, :, None].contiguous(), v_uvz[..., 0], v_uvz[..., 1])
        faces_v_idx, faces_v, faces_vt, span_eps = self._static_outputs(fill_back, depth.device)
        v_front_mask = ((v_uvz[0, :, 2] - v_depth[0, :, 0]) < span_eps)[None, :]
        return (gb['uv_map'], gb['alpha'], gb['face_index_map'], gb['weight_map'][..., None], faces_v_idx, gb['normal_map'],
                gb['normal_map_cam'], faces_v, faces_vt, gb['position_map'], gb['position_map_cam'], depth[..., None],
                v_uvz, v_front_mask)


class RenderingNet(nn.Module):
    """network.py:219-253.  forward(input [N,Cin,H,W], v_fea) -> tanh(U-Net) [N,Cout,H,W]."""

    def __init__(self, nf0, in_channels, out_channels, num_down_unet=5, out_channels_gcn=512, use_gcn=True,
                 outermost_highway_mode='concat'):
        super().__init__()
        self.register_buffer('nf0', torch.tensor(nf0))
        self.register_buffer('in_channels', torch.tensor(in_channels))
        self.register_buffer('out_channels', torch.tensor(out_channels))
        self.register_buffer('num_down_unet', torch.tensor(num_down_unet))
        self.register_buffer('out_channels_gcn', torch.tensor(out_channels_gcn))
        self.net = Unet(in_channels=in_channels, out_channels=out_channels, outermost_linear=True, use_dropout=True,
                        dropout_prob=0.1, nf0=nf0, norm=nn.BatchNorm2d, max_channels=8 * nf0, num_down=num_down_unet,
                        out_channels_gcn=out_channels_gcn, use_gcn=use_gcn, outermost_highway_mode=outermost_highway_mode)
        self.tanh = nn.Tanh()

    def forward(self, input, v_fea):
        return self.net.forward_fused(input, apply_tanh=True)      # tanh fused into the layout epilogue


class Interpolater(nn.Module):
    """network.py:318-337."""

    def forward(self, data, sub_x, sub_y):
        if data.shape[0] == 1:
            return misc.interpolate_bilinear(data[0].contiguous(), sub_x, sub_y)
        if data.shape[0] == sub_x.shape[0]:
            return torch.stack([misc.interpolate_bilinear(data[i].contiguous(), sub_x[i], sub_y[i]) for i in range(data.shape[0])])
        raise ValueError('data.shape[0] should be 1 or batch size')


class RaySampler(nn.Module):
    """network.py:417-472."""

    def __init__(self, num_azi, num_polar, interval_polar=5, mode='reflect'):
        super().__init__()
        self.register_buffer('num_azi', torch.tensor(num_azi))
        self.register_buffer('num_polar', torch.tensor(num_polar))
        self.register_buffer('interval_polar', torch.tensor(interval_polar))
        self.mode = mode
        na, npol, step = int(num_azi), int(num_polar), float(interval_polar)
        self.num_ray = na * npol + 1
        pol = np.arange(1, npol + 1) * step * np.pi / 180.0
        azi = np.arange(na) * 2 * np.pi / na
        pol, azi = np.meshgrid(pol, azi)
        self.rot_rad = np.vstack((np.zeros(na * npol), pol.flatten(), azi.flatten()))
        Rs = np.zeros((self.num_ray, 3, 3), np.float32)
        Rs[0] = np.eye(3)
        for i in range(self.num_ray - 1):       # data_util.euler_to_rot with rot_x = 0 (data_util.py:175-191)
            cy, sy = np.cos(self.rot_rad[1, i]), np.sin(self.rot_rad[1, i])
            cz, sz = np.cos(self.rot_rad[2, i]), np.sin(self.rot_rad[2, i])
            Rs[i + 1] = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]]).dot(np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]]))
        self.register_buffer('Rs', torch.from_numpy(Rs))
        self.register_buffer('pivots_dir', _ray_pivots(na, npol, step))

    def forward(self, TBN_matrices, view_dir_map_tangent, alpha_map):
        reflect = self.mode == 'reflect'
        dirs, uv, dt = ops.ray_sampler(reflect, self.pivots_dir, TBN_matrices.float(), view_dir_map_tangent.float(),
                                       alpha_map.float())
        return dirs, uv, (dt if reflect else self.pivots_dir)


class RayRenderer(nn.Module):
    """network.py:475-527."""

    def __init__(self, lighting_model, interpolater):
        super().__init__()
        self.lighting_model = lighting_model
        self.interpolater = interpolater

    def forward(self, albedo_specular, rays_uv, rays_lt, lighting_idx=None, lp=None, albedo_diffuse=None,
                num_ray_diffuse=0, no_albedo=False, seperate_albedo=False, lp_scale_factor=1):
        if lp is None:
            lp = self.lighting_model(lighting_idx, is_lp=True)
        lp_in = lp.float().contiguous()
        out, o_s, o_d, l_s, l_d, color = ops.ray_renderer(
            rays_uv.float().contiguous(), rays_lt.float().contiguous(), lp_in, albedo_specular.float().contiguous(),
            albedo_diffuse.float().contiguous() if albedo_diffuse is not None else None, num_ray_diffuse, no_albedo,
            seperate_albedo, float(lp_scale_factor))
        return out, o_s, o_d, l_s, l_d, color, lp * lp_scale_factor


class LightingSH(nn.Module):
    """network.py:534-627 without pyshtools: the lmax <= 16 basis comes from the HIP SH kernel."""

    def __init__(self, l_dir, lmax, num_lighting=1, num_channel=3, init_coeff=None, fix_params=False, lp_recon_h=100,
                 lp_recon_w=200):
        super().__init__()
        self.num_sample, self.lmax, self.num_basis = l_dir.shape[1], lmax, (lmax + 1) ** 2
        self.num_lighting, self.num_channel, self.fix_params = num_lighting, num_channel, fix_params
        self.lp_recon_h, self.lp_recon_w = lp_recon_h, lp_recon_w
        # (network.py:557 goes through numpy; the values are the same float32 numbers, kept on the way as a tensor)
        basis = sph_harm.evaluate_sh_basis(lmax=lmax, directions=l_dir.detach().cpu().numpy().transpose(), as_tensor=True)
        self.register_buffer('basis_val', basis.to(l_dir.dtype).to(l_dir.device))
        self.coeff = nn.Parameter(torch.zeros((num_lighting, self.num_basis, num_channel), dtype=torch.float32))
        if init_coeff is not None:
            if init_coeff.dim() == 2:
                init_coeff = init_coeff[None].repeat((num_lighting, 1, 1))
            self.coeff.data = init_coeff
        if fix_params:
            self.coeff.requires_grad_(False)
        self.register_buffer('l_samples', torch.einsum('sb,lbc->lsc', self.basis_val.cpu(), self.coeff.data.cpu()))
        vv, uu = torch.meshgrid(torch.arange(lp_recon_h, dtype=torch.float32) / (lp_recon_h - 1),
                                torch.arange(lp_recon_w, dtype=torch.float32) / (lp_recon_w - 1), indexing='ij')
        dirs = render.spherical_mapping_inv(torch.stack([uu, vv]).flatten(1)).permute(1, 0).numpy()
        self.register_buffer('basis_val_recon',
                             sph_harm.evaluate_sh_basis(lmax=lmax, directions=dirs, as_tensor=True).cpu().to(l_dir.dtype))

    def forward(self, lighting_idx=None, coeff=None, is_lp=None):
        if coeff is not None:
            return (self.reconstruct_lp(coeff) if is_lp else sph_harm.reconstruct_sh(coeff, self.basis_val))[None]
        if lighting_idx is not None:
            if is_lp:
                return self.reconstruct_lp(self.coeff[lighting_idx])[None]
            if self.fix_params:
                return self.l_samples[lighting_idx][None]
            return sph_harm.reconstruct_sh(self.coeff[lighting_idx][None], self.basis_val)
        if is_lp:
            return self.reconstruct_lp(self.coeff)[None]
        return (self.l_samples if self.fix_params else sph_harm.reconstruct_sh(self.coeff, self.basis_val))[None]

    def get_lighting_params(self, lighting_idx):
        return self.coeff[lighting_idx]

    def normalize_lighting(self, lighting_ref_idx):
        ref = self.coeff[lighting_ref_idx].norm('fro')
        f = ref / self.coeff.norm('fro', dim=[1, 2])
        f[lighting_ref_idx] = 1.0
        self.coeff *= f[:, None, None]

    def reconstruct_lp(self, coeff):
        lp = sph_harm.reconstruct_sh(coeff.detach(), self.basis_val_recon)
        return lp.reshape(lp.shape[:-2] + (int(self.lp_recon_h), int(self.lp_recon_w), lp.shape[-1]))


def _out_of_scope(name, why):
    class _Stub(nn.Module):
        def __init__(self, *a, **k):
            raise NotImplementedError('network.%s is outside the MI355X hot-path build: %s' % (name, why))
    _Stub.__name__ = name
    return _Stub


DenseDeepGCN = _out_of_scope('DenseDeepGCN', 'train-time only; inference loads its cached output v_feature, which is dead at the output')
InterpolaterVertexAttr = _out_of_scope('InterpolaterVertexAttr', 'unused at inference')
RaysLTChromLoss = _out_of_scope('RaysLTChromLoss', 'training loss')


class Mesh(nn.Module):
    """network.py:355-388: vertex positions / normals of the low-resolution mesh with global_RT applied and the span /
    centre statistics the scripts read (test_rnr.py:129-131 uses `num_vertex` only).  Host-side tensor bookkeeping."""

    def __init__(self, obj_fp, global_RT=None):
        super().__init__()
        v_attr, _ = nr.load_obj(obj_fp, normalization=False, use_cuda=False)
        v, vn = v_attr['v'].cpu(), v_attr['vn'].cpu()
        self.num_vertex = v.shape[0]
        self.v_orig, self.vn_orig = v.clone(), vn.clone()
        self.span_orig = v.max(dim=0)[0] - v.min(dim=0)[0]
        self.span_max_orig = self.span_orig.max()
        self.center_orig = v.mean(dim=0)
        if global_RT is not None:
            g = global_RT.to(v.device).float()
            v = torch.matmul(g, torch.cat((v, torch.ones(self.num_vertex, 1)), dim=1).t()).t()[:, :3]
            vn = torch.nn.functional.normalize(torch.matmul(g[:3, :3], vn.t()).t(), dim=1)
        self.register_buffer('v', v)
        self.register_buffer('vn', vn)
        self.span = v.max(dim=0)[0] - v.min(dim=0)[0]
        self.span_max = self.span.max()
        self.center = v.mean(dim=0)

    def forward(self):
        pass


class LightingLP(nn.Module):
    """network.py:631-699: light probes -> 1600 x 3200 area-averaged probes (`lps`) -> bilinear samples at the
    `l_dir` directions (`l_samples`) -> SH projection (`fit_sh`).  The cv2 INTER_AREA resize, the bilinear taps and the
    SH basis / projection run as HIP kernels (rnr_resize_area, rnr_interpolate_bilinear, rnr_sh_basis, rnr_sh_fit) on
    `device` (default: the device of l_dir if it is a GPU tensor, else the current GPU); buffers keep the reference's
    names, shapes and (CPU-by-default) placement.  cv2 parity of the resize is unpinned (no cv2 in this image)."""

    def __init__(self, l_dir, num_lighting=1, num_channel=3, lp_dataloader=None, fix_params=False, lp_img_h=1600,
                 lp_img_w=3200, device=None):
        super().__init__()
        self.register_buffer('l_dir', l_dir)
        self.num_sample, self.num_lighting, self.num_channel = l_dir.shape[1], num_lighting, num_channel
        self.fix_params, self.lp_img_h, self.lp_img_w = fix_params, lp_img_h, lp_img_w
        if lp_dataloader is not None:
            self.num_lighting = len(lp_dataloader)
        self.register_buffer('l_samples_uv', render.spherical_mapping(l_dir))
        self.l_samples = nn.Parameter(torch.zeros((self.num_lighting, self.num_sample, self.num_channel), dtype=torch.float32))
        self._device_arg = device
        if lp_dataloader is not None:
            uv = self.l_samples_uv.to(self._dev).float()
            lps = []
            for idx, lp in enumerate(lp_dataloader):
                img = lp['lp_img'][0].permute(1, 2, 0).float().contiguous().to(self._dev)             # [H,W,C]
                img = ops.resize_area(img, lp_img_h, lp_img_w)                                          # network.py:667
                x = (uv[0] * float(lp_img_w)).clamp(max=lp_img_w - 1).contiguous()                     # network.py:669
                y = (uv[1] * float(lp_img_h)).clamp(max=lp_img_h - 1).contiguous()
                self.l_samples.data[idx] = ops.interpolate_bilinear(img, x, y).to(self.l_samples.device)
                lps.append(img.to(l_dir.device))
            self.register_buffer('lps', torch.stack(lps))
        if self.fix_params:
            self.l_samples.requires_grad_(False)

    @property
    def _dev(self):
        """GPU the HIP operators of this module run on (resolved on first use: constructing the module with
        lp_dataloader=None needs no GPU, like the reference)."""
        if self._device_arg is not None:
            return torch.device(self._device_arg)
        return self.l_dir.device if self.l_dir.is_cuda else torch.device('cuda', torch.cuda.current_device())

    def forward(self, lighting_idx=None, is_lp=False):
        src = self.lps if is_lp else self.l_samples
        return src[None] if lighting_idx is None else src[lighting_idx][None]

    def fit_sh(self, lmax):
        """network.py:694-699: registers `sh_coeff` [num_lighting, (lmax+1)^2, num_channel]."""
        dirs = self.l_dir.detach().t().contiguous().float().to(self._dev)
        basis = ops.sh_basis(dirs, int(lmax))
        coeff = sph_harm.fit_sh_coeff(samples=self.l_samples.detach().to(self._dev), sh_basis_val=basis)
        self.register_buffer('sh_coeff', coeff.to(self.l_dir.device).to(self.l_dir.dtype))
