"""The reference's inference script as a callable: module construction (test_rnr.py:129-233) and the body of its per-view
loop (test_rnr.py:265-377) written against the DROP-IN modules (`network`, `render`, `camera`, `sph_harm`) — "Level 1" of
INTEGRATION.md: a user who only switches the import path runs exactly this call sequence.  bench.py times it
(`dropin_view_loop`), tests/test_gpu_dropin.py checks it at the benchmarked size against RNRPipeline and the oracle.

Nothing here is fused: every call below is one of the reference's own calls, with the reference's tensor shapes in between
(TBN_map [N,H,W,3,3], rays_dir [N,H,W,3,13], the 113 MB torch.cat of test_rnr.py:349-356 ...).  The fused form of the same
frame is rnr_amd.pipeline.RNRPipeline (Level 2).
"""
import time

import numpy as np
import torch

STAGES = ('rasterizer', 'get_TBN_map', 'get_view_dir_map', 'view_dir_tangent(torch.matmul -> rnr_tbn_matvec; normalize)', 'evaluate_sh_basis(+host)', 'texture_mapper',
          'ray_sampler x2', 'cat(torch)', 'render_net', 'post_scale(torch)', 'ray_renderer')


class DropinViewLoop:
    def __init__(self, obj_fp, img_size, textures, unet_sd, sh_coeff, nf0, device='cuda:0', sh_lmax=10, sh_on_device=False,
                 out_channels_gcn=512):
        """obj_fp: the high-resolution mesh (opt.obj_high_fp); textures: 4 mip levels [1,S_l,S_l,C] (the checkpoint's
        texture_mapper.textures.*); unet_sd: RenderingNet state dict (checkpoint 'render_net'); sh_coeff [L,(lmax+1)^2,3]:
        LightingSH.coeff (checkpoint 'lighting_model').
        sh_on_device: False = test_rnr.py:322-328 verbatim (view_dir_map -> host numpy -> evaluate_sh_basis -> float64 numpy ->
        float32 -> device: two mandated host round trips per view); True = the one-line variant
        `sph_harm.evaluate_sh_basis(lmax=2, directions=view_dir_map.reshape(-1, 3), as_tensor=True)` that keeps the basis in HBM."""
        import network
        from rnr_amd import scene
        self.device = torch.device(device)
        self.S = int(img_size)
        self.sh_on_device = bool(sh_on_device)
        C = int(textures[0].shape[-1])
        # test_rnr.py:132-145
        self.interpolater = network.Interpolater()
        self.texture_mapper = network.TextureMapper(texture_size=int(textures[0].shape[-2]), texture_num_ch=C,
                                                    mipmap_level=len(textures), texture_init=None, fix_texture=True, apply_sh=True)
        tsd = self.texture_mapper.state_dict()
        for i, t in enumerate(textures):
            tsd['textures.%d' % i] = torch.as_tensor(t, dtype=torch.float32).reshape(1, t.shape[-3], t.shape[-2], C)
        self.texture_mapper.load_state_dict(tsd, strict=True)
        # test_rnr.py:147-166 ('train' lighting: SH coefficients from the checkpoint)
        l_dir = torch.from_numpy(np.ascontiguousarray(scene.sphere_samples(4096))).t().contiguous()
        coeff = torch.as_tensor(sh_coeff, dtype=torch.float32)
        self.lighting_model = network.LightingSH(l_dir, lmax=sh_lmax, num_lighting=coeff.shape[0], num_channel=3, fix_params=True)
        self.lighting_model.coeff.data = coeff
        # test_rnr.py:168-185 (train_rnr.py:344-354 defaults: 6 x 2 pivots, 5 / 10 degrees)
        self.ray_sampler = network.RaySampler(num_azi=6, num_polar=2, interval_polar=5)
        self.ray_sampler_diffuse = network.RaySampler(num_azi=6, num_polar=2, interval_polar=10, mode='diffuse')
        self.num_ray, self.num_ray_diffuse = self.ray_sampler.num_ray, self.ray_sampler_diffuse.num_ray
        self.num_ray_total = self.num_ray + self.num_ray_diffuse
        # test_rnr.py:187-198
        self.render_net = network.RenderingNet(nf0=nf0, in_channels=self.num_ray_total * 3 + 6 + C,
                                               out_channels=3 * self.num_ray_total, num_down_unet=5,
                                               out_channels_gcn=out_channels_gcn)
        full = self.render_net.state_dict()
        for k, v in unet_sd.items():
            full[k] = torch.as_tensor(v)
        self.render_net.load_state_dict(full, strict=True)
        self.v_feature = torch.zeros(1, out_channels_gcn)
        # test_rnr.py:203-207
        self.ray_renderer = network.RayRenderer(self.lighting_model, self.interpolater)
        self.rasterizer = network.Rasterizer(obj_fp=obj_fp, img_size=self.S, global_RT=None)
        # test_rnr.py:209-233
        for m in (self.interpolater, self.texture_mapper, self.lighting_model, self.ray_sampler, self.ray_sampler_diffuse,
                  self.render_net, self.ray_renderer, self.rasterizer):
            m.to(self.device)
            m.eval()
        self.v_feature = self.v_feature.to(self.device)
        for m in self.render_net.modules():
            if type(m) == torch.nn.BatchNorm2d:
                m.train()

    def view(self, proj, pose, proj_inv, R_inv, lighting_idx=0, events=None, keep=None, host_times=None):
        """One iteration of test_rnr.py:265-377 for device tensors proj / proj_inv / R_inv [1,3,3], pose [1,4,4]:
        -> outputs_final [1,3,S,S].  events: list that receives (stage, torch.cuda.Event) at the stage boundaries (STAGES);
        keep: dict that receives the intermediate maps parity tests compare."""
        import camera
        import render
        import sph_harm
        device = self.device

        def mark(name):
            if events is not None:
                e = torch.cuda.Event(enable_timing=True)
                e.record()
                events.append((name, e))
            if host_times is not None:          # diagnosis only: host wall clock at the boundary, device drained first (unless .drain is off)
                if getattr(self, 'drain', True):
                    torch.cuda.synchronize()
                host_times.append((name, time.perf_counter()))
        with torch.no_grad():
            mark('start')
            # rasterize (test_rnr.py:282-295)
            uv_map, alpha_map, face_index_map, weight_map, faces_v_idx, normal_map, normal_map_cam, faces_v, faces_vt, \
                position_map, position_map_cam, depth, v_uvz, v_front_mask = \
                self.rasterizer(proj=proj, pose=pose, dist_coeffs=None, offset=None, scale=None)
            mark('rasterizer')
            batch_size, img_h, img_w = alpha_map.shape[0], alpha_map.shape[1], alpha_map.shape[2]
            # test_rnr.py:309-316
            TBN_map = render.get_TBN_map(normal_map, face_index_map, faces_v=faces_v[0, :], faces_texcoord=faces_vt[0, :],
                                         tangent=None)
            mark('get_TBN_map')
            view_dir_map, _ = camera.get_view_dir_map(uv_map.shape[1:3], proj_inv, R_inv)
            mark('get_view_dir_map')
            view_dir_map_tangent = torch.matmul(TBN_map.reshape((-1, 3, 3)).transpose(-2, -1),
                                                view_dir_map.reshape((-1, 3, 1)))[..., 0].reshape(view_dir_map.shape)
            view_dir_map_tangent = torch.nn.functional.normalize(view_dir_map_tangent, dim=-1)
            mark('view_dir_tangent(torch.matmul -> rnr_tbn_matvec; normalize)')
            # SH basis value for view_dir_map (test_rnr.py:320-329, the force_recompute branch)
            if self.sh_on_device:
                sh_basis_map = sph_harm.evaluate_sh_basis(lmax=2, directions=view_dir_map.reshape((-1, 3)), as_tensor=True) \
                    .reshape((*(view_dir_map.shape[:3]), -1))
            else:
                if host_times is None:
                    sh_basis_map = sph_harm.evaluate_sh_basis(lmax=2, directions=view_dir_map.reshape((-1, 3)).cpu().detach().numpy()) \
                        .reshape((*(view_dir_map.shape[:3]), -1)).astype(np.float32)      # [N, H, W, 9]
                else:       # the same expression, statement by statement, for the host-time breakdown
                    d_np = view_dir_map.reshape((-1, 3)).cpu().detach().numpy()
                    mark('sh: .cpu().detach().numpy()')
                    b64 = sph_harm.evaluate_sh_basis(lmax=2, directions=d_np)
                    mark('sh: evaluate_sh_basis')
                    sh_basis_map = b64.reshape((*(view_dir_map.shape[:3]), -1)).astype(np.float32)
                    mark('sh: .astype(np.float32)')
                sh_basis_map = torch.from_numpy(sh_basis_map).to(device)
            mark('evaluate_sh_basis(+host)')
            # sample texture (test_rnr.py:333-336)
            neural_img = self.texture_mapper(uv_map, sh_basis_map, sh_start_ch=6)      # [N, C, H, W]
            albedo_diffuse = neural_img[:, :3, :, :]
            albedo_specular = neural_img[:, 3:6, :, :]
            mark('texture_mapper')
            # rays (test_rnr.py:338-346)
            rays_dir, rays_uv, rays_dir_tangent = self.ray_sampler(TBN_map, view_dir_map_tangent, alpha_map[..., None])
            rays_diffuse_dir, rays_diffuse_uv, _ = self.ray_sampler_diffuse(TBN_map, view_dir_map_tangent, alpha_map[..., None])
            mark('ray_sampler x2')
            # concat data (test_rnr.py:348-356)
            rays_dir = torch.cat((rays_dir, rays_diffuse_dir), dim=-1)
            rays_uv = torch.cat((rays_uv, rays_diffuse_uv), dim=-1)
            render_net_input = torch.cat((rays_dir.permute((0, -1, -2, 1, 2)).reshape((batch_size, -1, img_h, img_w)),
                                          normal_map.permute((0, 3, 1, 2)),
                                          view_dir_map.permute((0, 3, 1, 2)),
                                          neural_img), dim=1)
            mark('cat(torch)')
            rays_lt = self.render_net(render_net_input, self.v_feature).reshape((batch_size, self.num_ray_total, -1, img_h, img_w))
            mark('render_net')
            lt_max_val = 2.0
            rays_lt = (rays_lt * 0.5 + 0.5) * lt_max_val                               # test_rnr.py:358-359
            mark('post_scale(torch)')
            # test_rnr.py:368
            outputs_final, _, _, _, _, _, lp = self.ray_renderer(albedo_specular, rays_uv, rays_lt, lighting_idx=lighting_idx,
                                                                 albedo_diffuse=albedo_diffuse,
                                                                 num_ray_diffuse=self.num_ray_diffuse, lp_scale_factor=1,
                                                                 seperate_albedo=True)
            mark('ray_renderer')
        if keep is not None:
            keep.update(face_index_map=face_index_map, alpha_map=alpha_map, uv_map=uv_map, normal_map=normal_map,
                        render_net_input=render_net_input, sh_basis_map=sh_basis_map, v_uvz=v_uvz, lp=lp)
        return outputs_final


def stage_table(events_per_view):
    """[(stage, event), ...] per view -> {stage: mean ms} over the views (call after a synchronize)."""
    acc, n = {}, 0
    for evs in events_per_view:
        for (_, e0), (name, e1) in zip(evs[:-1], evs[1:]):
            acc[name] = acc.get(name, 0.0) + e0.elapsed_time(e1)
        n += 1
    return {k: v / max(n, 1) for k, v in acc.items()}
