"""-m gpu: the reference's per-view loop (test_rnr.py:265-377) written against the DROP-IN modules
(`network`, `render`, `camera`, `sph_harm`, `neural_renderer`), compared with the golden vectors the reference's own
modules produced.  This is the "a user switches the import path and nothing else" check."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def T(x):
    return torch.from_numpy(np.ascontiguousarray(x))


def test_rnr_view_loop_with_dropin_modules(golden, tmp_path):
    import camera
    import network
    import render
    import sph_harm
    from oracle import rnr_oracle as orc
    from rnr_amd import scene
    gm, gf = golden('rasterizer_module64'), golden('frame64')
    obj = str(tmp_path / 'sphere.obj')
    scene.write_obj(obj, {k: gm['mesh_' + k] for k in ['v', 'vt', 'vn', 'f_v_idx', 'f_vt_idx', 'f_vn_idx']})
    S = 64
    rasterizer = network.Rasterizer(obj_fp=obj, img_size=S, global_RT=T(gm['global_RT'])).to(DEV)
    assert torch.allclose(rasterizer.vertices.cpu(), T(gm['buf_vertices']), atol=1e-6)
    texture_mapper = network.TextureMapper(32, 16, 4, apply_sh=True)
    tsd = texture_mapper.state_dict()
    for i in range(4):
        tsd['textures.%d' % i] = T(gf['tex%d' % i])
    texture_mapper.load_state_dict(tsd, strict=True)
    ray_sampler = network.RaySampler(6, 2, 5)
    ray_sampler_diffuse = network.RaySampler(6, 2, 10, mode='diffuse')
    nr_total = ray_sampler.num_ray + ray_sampler_diffuse.num_ray
    render_net = network.RenderingNet(nf0=4, in_channels=nr_total * 3 + 6 + 16, out_channels=3 * nr_total, num_down_unet=5,
                                      out_channels_gcn=16)
    sd = render_net.state_dict()
    for k in gf.files:
        if k.startswith('sd:'):
            sd[k[3:]] = T(gf[k])
    render_net.load_state_dict(sd, strict=True)
    ray_renderer = network.RayRenderer(None, network.Interpolater())
    for m in (texture_mapper, ray_sampler, ray_sampler_diffuse, render_net, ray_renderer):
        m.to(DEV).eval()
    for m in render_net.modules():        # test_rnr.py:229-233
        if type(m) == torch.nn.BatchNorm2d:
            m.train()
    v_feature = torch.zeros(1, 16, device=DEV)
    lp = T(gf['lp']).to(DEV)
    names = ['uv_map', 'alpha', 'face_index_map', 'weight_map', 'faces_v_idx', 'normal_map', 'normal_map_cam', 'faces_v',
             'faces_vt', 'position_map', 'position_map_cam', 'depth', 'v_uvz', 'v_front_mask']
    with torch.no_grad():
        for i in range(2):
            proj, pose = T(gm['proj'][i:i + 1]).to(DEV), T(gm['pose'][i:i + 1]).to(DEV)
            proj_inv, R_inv = T(gm['proj_inv'][i:i + 1]).to(DEV), T(gm['R_inv'][i:i + 1]).to(DEV)
            tup = rasterizer(proj=proj, pose=pose, dist_coeffs=None, offset=None, scale=None)
            assert len(tup) == 14
            mism = (tup[2].cpu().numpy() != gm['view%d_face_index_map' % i])
            assert mism.mean() < 2e-3
            for n, x in zip(names, tup):
                ref = gm['view%d_%s' % (i, n)]
                assert tuple(x.shape) == tuple(ref.shape), (n, x.shape, ref.shape)
            assert torch.equal(tup[13].cpu(), T(gm['view%d_v_front_mask' % i]))
            assert np.abs(tup[12].cpu().numpy() - gm['view%d_v_uvz' % i]).max() < 1e-3
            uv_map, alpha_map, face_index_map, _, _, normal_map, _, faces_v, faces_vt = tup[:9]
            TBN_map = render.get_TBN_map(normal_map, face_index_map, faces_v=faces_v[0], faces_texcoord=faces_vt[0])
            view_dir_map, _ = camera.get_view_dir_map(uv_map.shape[1:3], proj_inv, R_inv)
            vt = torch.matmul(TBN_map.reshape((-1, 3, 3)).transpose(-2, -1), view_dir_map.reshape((-1, 3, 1)))[..., 0]
            view_dir_map_tangent = torch.nn.functional.normalize(vt.reshape(view_dir_map.shape), dim=-1)
            sh_np = sph_harm.evaluate_sh_basis(lmax=2, directions=view_dir_map.reshape((-1, 3)).cpu().numpy())
            sh_basis_map = torch.from_numpy(sh_np.reshape((1, S, S, -1)).astype(np.float32)).to(DEV)
            assert np.abs(sh_basis_map.cpu().numpy()[~mism] - gf['sh_basis_map'][i:i + 1][~mism]).max() < 2e-6
            neural_img = texture_mapper(uv_map, sh_basis_map, sh_start_ch=6)
            rays_dir, rays_uv, _ = ray_sampler(TBN_map, view_dir_map_tangent, alpha_map[..., None])
            rays_d_dir, rays_d_uv, _ = ray_sampler_diffuse(TBN_map, view_dir_map_tangent, alpha_map[..., None])
            rays_dir = torch.cat((rays_dir, rays_d_dir), dim=-1)
            rays_uv = torch.cat((rays_uv, rays_d_uv), dim=-1)
            net_in = torch.cat((rays_dir.permute((0, -1, -2, 1, 2)).reshape((1, -1, S, S)), normal_map.permute((0, 3, 1, 2)),
                                view_dir_map.permute((0, 3, 1, 2)), neural_img), dim=1)
            ok = torch.from_numpy(~mism)[:, None].expand_as(net_in.cpu())
            assert (net_in.cpu() - T(gf['net_in'][i:i + 1]).float())[ok].abs().max() < 2e-3      # golden stored as fp16
            rays_lt = render_net(net_in, v_feature).reshape((1, nr_total, -1, S, S))
            rays_lt = (rays_lt * 0.5 + 0.5) * 2.0
            out = ray_renderer(neural_img[:, 3:6], rays_uv, rays_lt, lp=lp, albedo_diffuse=neural_img[:, :3],
                               num_ray_diffuse=ray_sampler_diffuse.num_ray, seperate_albedo=True)
            assert len(out) == 7
            p = orc.psnr(out[0].cpu(), T(gf['image'][i:i + 1]))
            assert p > 55.0, p


@pytest.mark.parametrize('sh_on_device', [False, True])
def test_rnr_view_loop_at_bench_size(tmp_path, sh_on_device):
    """Level 1 of INTEGRATION.md AT THE BENCHMARKED SIZE: test_rnr.py:265-377's call sequence through the
    drop-in modules (rnr_amd.view_loop.DropinViewLoop = network.Rasterizer -> render.get_TBN_map -> camera.get_view_dir_map ->
    sph_harm.evaluate_sh_basis [numpy in / numpy out, or the one-line device variant] -> TextureMapper -> 2 x RaySampler ->
    torch.cat -> RenderingNet -> RayRenderer) on bench.py's scene: 65 536-face sphere read back from an OBJ file, 512^2, C = 24,
    nf0 = 64, SH lighting lmax 10.
      * face_index_map / alpha EQUAL to RNRPipeline's G-buffer of the same pose and to the oracle on the same projected vertices;
      * the 108-channel network input equals the fused stage's to 2e-5;
      * frame vs RNRPipeline.render of the same pose <= 1e-5; frame PSNR vs the oracle >= 60 dB."""
    from oracle import rnr_oracle as orc
    from rnr_amd import scene
    from rnr_amd.pipeline import RNRPipeline
    from rnr_amd.view_loop import DropinViewLoop
    from test_gpu_frame import _bench_scene
    S = 512
    sc = _bench_scene()
    obj = str(tmp_path / 'sphere65536.obj')
    scene.write_obj(obj, sc['mesh'])
    loop = DropinViewLoop(obj, S, sc['textures'], sc['unet_sd'], sc['sh_coeff'], nf0=64, device=DEV, sh_on_device=sh_on_device)
    assert loop.rasterizer.num_face == 65536 and loop.num_ray_total == 26
    pipe = RNRPipeline(sc['mesh'], S, sc['textures'], sc['unet_sd'], sc['pivots_spec'], sc['pivots_diff'], None, nf0=64,
                       max_views=1, device=DEV, sh_coeff=sc['sh_coeff'], sh_lmax=10, skip_background_tiles=False)
    mesh_t = {k: torch.as_tensor(v) for k, v in sc['mesh'].items()}
    basis = torch.from_numpy(orc.sh_basis(10, orc.lp_recon_dirs().numpy()).astype(np.float32))
    lp_ref = orc.reconstruct_lp(sc['sh_coeff'][0], basis)[None]
    for vid in ([111, 640] if not sh_on_device else [37]):
        views = {k: T(v) for k, v in scene.spiral_views(S, [vid]).items()}
        dv = {k: v.to(DEV) for k, v in views.items()}
        keep = {}
        out = loop.view(dv['proj'], dv['pose'], dv['proj_inv'], dv['R_inv'], lighting_idx=0, keep=keep)
        assert tuple(out.shape) == (1, 3, S, S)
        ref = pipe.render(dv['proj'], dv['pose'], dv['proj_inv'], dv['R_inv'], keep_intermediates=True)
        last = pipe.last
        assert torch.equal(keep['face_index_map'], last['gb']['face_index_map'])
        assert torch.equal(keep['alpha_map'], last['gb']['alpha'])
        ni = last['net_in'][..., :108].permute(0, 3, 1, 2)
        assert float((keep['render_net_input'] - ni).abs().max()) <= 2e-5
        assert float((keep['lp'][0] - lp_ref[0].to(DEV)).abs().max()) <= 2e-5
        d = float((out - ref).abs().max())
        assert d <= 1e-5, (vid, d)
        # oracle, one view per call, on the same projected vertices
        gb = orc.rasterizer_forward(mesh_t, views['proj'], views['pose'], S, v_uvz_ndc=last['v_uvz'].cpu())
        assert torch.equal(keep['face_index_map'].cpu(), gb['face_index_map'])
        sh_in = orc.shade_inputs(gb, views['proj_inv'], views['R_inv'], sc['textures'], sc['pivots_spec'], sc['pivots_diff'], 6)
        y_ref = orc.unet_forward(sc['unet_sd'], sh_in['net_in'])
        rays_lt = (y_ref.reshape(1, 26, -1, S, S) * 0.5 + 0.5) * 2.0
        neural = sh_in['neural_img']
        ref_img = orc.ray_renderer(neural[:, 3:6], sh_in['rays_uv'], rays_lt, lp_ref, albedo_diffuse=neural[:, :3],
                                   num_ray_diffuse=13, seperate_albedo=True)[0]
        p = orc.psnr(out.cpu(), ref_img)
        assert p >= 60.0, (vid, p)


def test_dropin_ops_vs_golden(golden):
    """TextureMapper (any C), RaySampler, RayRenderer, get_TBN_map, get_view_dir_map one by one."""
    import camera
    import network
    import render
    g = golden('texture_mapper')
    tm = network.TextureMapper(32, 16, 4, apply_sh=True)
    sd = tm.state_dict()
    for i in range(4):
        sd['textures.%d' % i] = T(g['tex%d' % i])
    tm.load_state_dict(sd, strict=True)
    tm.to(DEV)
    uv, sh = T(g['uv']).to(DEV), T(g['sh']).to(DEV)
    assert torch.allclose(tm(uv, sh, sh_start_ch=6).cpu(), T(g['out_sh6']), atol=2e-6)
    assert torch.allclose(tm(uv, sh).cpu(), T(g['out_sh3']), atol=2e-6)
    assert torch.allclose(tm(uv, None).cpu(), T(g['out_nosh']), atol=2e-6)
    g = golden('shading_geometry64')
    tbn = render.get_TBN_map(T(g['normal_map']).to(DEV), T(g['face_index_map']).to(DEV), faces_v=T(g['faces_v'])[0].to(DEV),
                             faces_texcoord=T(g['faces_vt'])[0].to(DEV), check_nan=True)
    assert torch.allclose(tbn.cpu(), T(g['tbn']), atol=2e-6)
    vd, vdc = camera.get_view_dir_map((64, 64), T(g['proj_inv']).to(DEV), T(g['R_inv']).to(DEV))
    assert torch.allclose(vd.cpu(), T(g['view_dir']), atol=2e-6) and torch.allclose(vdc.cpu(), T(g['view_dir_cam']), atol=2e-6)
    rs, rd = network.RaySampler(6, 2, 5).to(DEV), network.RaySampler(6, 2, 10, mode='diffuse').to(DEV)
    assert torch.allclose(rs.Rs.cpu(), T(g['Rs_spec']), atol=1e-7) and torch.allclose(rs.pivots_dir.cpu(), T(g['pivots_spec']), atol=1e-7)
    alpha = T(g['alpha'])[..., None].to(DEV)
    d, uvr, dt = rs(T(g['tbn']).to(DEV), T(g['view_tangent']).to(DEV), alpha)
    assert torch.allclose(d.cpu(), T(g['rays_dir_spec']), atol=2e-5)
    assert torch.allclose(uvr.cpu(), T(g['rays_uv_spec']), atol=2e-5)
    assert torch.allclose(dt.cpu(), T(g['rays_dir_tangent_spec']), atol=2e-5)
    d, uvr, _ = rd(T(g['tbn']).to(DEV), T(g['view_tangent']).to(DEV), alpha)
    assert torch.allclose(d.cpu(), T(g['rays_dir_diff']), atol=2e-5) and torch.allclose(uvr.cpu(), T(g['rays_uv_diff']), atol=2e-5)
    g = golden('ray_renderer')
    rr = network.RayRenderer(None, network.Interpolater())
    out = rr(T(g['albedo_specular']).to(DEV), T(g['rays_uv']).to(DEV), T(g['rays_lt']).to(DEV), lp=T(g['lp']).to(DEV),
             albedo_diffuse=T(g['albedo_diffuse']).to(DEV), num_ray_diffuse=13, seperate_albedo=True)
    for a, k in zip(out, ['out', 'out_specular', 'out_diffuse', 'ltt_specular', 'ltt_diffuse', 'rays_color']):
        assert torch.allclose(a.cpu(), T(g[k]), atol=5e-6), k
    out = rr(T(g['albedo_specular']).to(DEV), T(g['rays_uv']).to(DEV), T(g['rays_lt']).to(DEV), lp=T(g['lp']).to(DEV))
    assert torch.allclose(out[0].cpu(), T(g['out_nodiffuse']), atol=5e-6)


def test_lighting_sh_module():
    import network
    from oracle import rnr_oracle as orc
    from rnr_amd import scene
    l_dir = T(scene.sphere_samples(4096)).t().contiguous()
    coeff = T(scene.synthetic_sh_coeff(2, 10, 1))
    lm = network.LightingSH(l_dir, lmax=10, num_lighting=2, init_coeff=coeff, fix_params=True).to(DEV)
    assert lm.basis_val.shape == (4096, 121) and lm.basis_val_recon.shape == (20000, 121)
    lp = lm(lighting_idx=1, is_lp=True)
    assert lp.shape == (1, 100, 200, 3)
    basis = torch.from_numpy(orc.sh_basis(10, orc.lp_recon_dirs().numpy()).astype(np.float32))
    ref = orc.reconstruct_lp(coeff[1], basis)
    assert torch.allclose(lp[0].cpu(), ref, atol=2e-5)


def test_neural_renderer_api_on_gpu(golden):
    """nr.Renderer / nr.rasterize_rgbad through the drop-in package vs the kernel golden (flip included)."""
    import neural_renderer as nr
    g = golden('raster_sphere128')
    faces = T(g['faces']).to(DEV)
    out = nr.rasterize_rgbad(faces, None, 128, anti_aliasing=False, near=0.0, far=1e5, return_rgb=False)
    assert np.array_equal(out['face_index_map'].cpu().numpy(), g['face_index_map'][:, ::-1])
    assert np.array_equal(out['depth'].cpu().numpy(), g['depth_map'][:, ::-1])
    assert np.array_equal(out['alpha'].cpu().numpy(), (g['face_index_map'][:, ::-1] >= 0).astype(np.float32))
    tex = torch.zeros(2, faces.shape[1], 4, 4, 4, 3, device=DEV)
    out = nr.rasterize_rgbad(faces, tex, 64, anti_aliasing=True, near=0.0, far=1e5, eps=1e-3)
    assert out['rgb'].shape == (2, 3, 64, 64) and out['alpha'].shape == (2, 64, 64)
    assert float(out['rgb'].abs().max()) == 0.0


def test_dnr_view_with_dropin_modules():
    """BASELINE config 1 (test_dnr.py:166-217): texture_mapper(uv, sh) [sh_start_ch=3] -> RenderingNet(30 -> 3, nf0 = 80,
    use_gcn=False) -> (y*0.5+0.5)*2 * alpha, one 256x256 view, real DNR sizes (train_dnr.py:29-38)."""
    import network
    from oracle import rnr_oracle as orc
    from rnr_amd import ops, scene, testing
    S, C, nf0 = 256, 30, 80
    mesh = scene.uv_sphere(32, 64)
    dm = ops.DeviceMesh(mesh['v'], mesh['vt'], mesh['vn'], mesh['f_v_idx'], mesh['f_vt_idx'], mesh['f_vn_idx'], DEV)
    v = {k: T(x) for k, x in scene.spiral_views(S, [33]).items()}
    proj, pose = v['proj'].to(DEV), v['pose'].to(DEV)
    v_uvz = ops.project_vertices(dm.v, proj, pose[:, :3, :3].contiguous(), pose[:, :3, 3].contiguous(), S)
    gb = ops.rasterize_gbuffer(dm, v_uvz, pose, S)
    import camera
    import sph_harm
    vd, _ = camera.get_view_dir_map((S, S), v['proj_inv'].to(DEV), v['R_inv'].to(DEV))
    sh = torch.from_numpy(sph_harm.evaluate_sh_basis(lmax=2, directions=vd.reshape(-1, 3).cpu().numpy())
                          .reshape(1, S, S, 9).astype(np.float32)).to(DEV)
    tm = network.TextureMapper(512, C, 4, apply_sh=True)
    tex = testing.synthetic_textures(512, C, 4, 5)
    tsd = tm.state_dict()
    for i in range(4):
        tsd['textures.%d' % i] = tex[i]
    tm.load_state_dict(tsd, strict=True)
    net = network.RenderingNet(nf0=nf0, in_channels=C, out_channels=3, num_down_unet=5, use_gcn=False)
    sd = testing.unet_state_dict(C, 3, nf0, seed=5, use_gcn=False)
    full = net.state_dict()
    for k, val in sd.items():
        full[k[4:] if not k.startswith('net.') else k] = val
    net.load_state_dict(full, strict=True)      # aliases (in_layer.0.weight ...) are tied to the same Parameters
    tm.to(DEV).eval()
    net.to(DEV).eval()
    for m in net.modules():
        if type(m) == torch.nn.BatchNorm2d:
            m.train()
    with torch.no_grad():
        neural_img = tm(gb['uv_map'], sh)                               # default sh_start_ch = 3 (test_dnr.py:210)
        y = net(neural_img, None)
        out = (y * 0.5 + 0.5) * 2.0 * gb['alpha'][:, None]
    ref_neural = orc.texture_mapper([t for t in tex], gb['uv_map'].cpu(), sh.cpu(), 3)
    assert torch.allclose(neural_img.cpu(), ref_neural, atol=2e-5)
    ref_y = orc.unet_forward(sd, ref_neural)
    ref = (ref_y * 0.5 + 0.5) * 2.0 * gb['alpha'].cpu()[:, None]
    assert out.shape == (1, 3, S, S)
    assert orc.psnr(out.cpu(), ref, peak=2.0) > 60.0


def test_precompute_export(golden, tmp_path):
    """G-buffer export (.mat, precompute.py's directory layout / keys) from the HIP rasterizer, read back and compared
    with the reference-generated maps of the rasterizer_module64 fixture."""
    import scipy.io
    import network
    from rnr_amd import precompute, scene
    gm = golden('rasterizer_module64')
    obj = str(tmp_path / 'sphere.obj')
    scene.write_obj(obj, {k: gm['mesh_' + k] for k in ['v', 'vt', 'vn', 'f_v_idx', 'f_vt_idx', 'f_vn_idx']})
    ras = network.Rasterizer(obj_fp=obj, img_size=64, global_RT=T(gm['global_RT'])).to(DEV)
    view = {k: T(gm[k][:1]) for k in ['proj', 'pose', 'proj_inv', 'R_inv']}
    out = str(tmp_path / 'precomp')
    precompute.export_view_maps(ras, view, out, '00000')
    r = scipy.io.loadmat(out + '/raster/00000.mat')
    assert set(['face_index_map', 'weight_map', 'faces_v_idx', 'v_uvz', 'v_front_mask']) <= set(r.keys())
    mism = r['face_index_map'] != gm['view0_face_index_map'][0]
    assert mism.mean() < 2e-3
    uv = scipy.io.loadmat(out + '/uv_map/00000.mat')['uv_map']
    d = np.abs(uv - gm['view0_uv_map'][0])[~mism]
    assert np.minimum(d, 1 - d).max() < 1e-4
    nm = scipy.io.loadmat(out + '/normal_map/00000.mat')['normal_map']
    assert np.abs(nm - gm['view0_normal_map'][0])[~mism].max() < 1e-4
    for sub in ['TBN_map', 'pose', 'proj', 'normal_map_cam', 'position_map', 'position_map_cam', 'view_dir_map',
                'view_dir_map_cam', 'view_dir_map_tangent', 'sh_basis_map', 'reflect_dir_map']:
        assert os.path.isfile('%s/%s/00000.mat' % (out, sub)), sub
    gs = golden('shading_geometry64')
    vd = scipy.io.loadmat(out + '/view_dir_map/00000.mat')['view_dir_map']
    assert np.abs(vd - gs['view_dir'][0]).max() < 2e-6


def test_texture_extension_modules_golden(golden):
    """neural_renderer.cuda.load_textures / create_texture_image on the HIP kernels vs the reference kernels' outputs
    (bit-exact: same binary32 operation order, no contraction)."""
    import neural_renderer.cuda.load_textures as lt
    import neural_renderer.cuda.create_texture_image as cti
    g = golden('load_textures40')
    dev = 'cuda:0'
    for w in range(4):
        for b in (1, 0):
            faces = torch.from_numpy(g['faces_uv']).to(dev).contiguous()
            tex = torch.from_numpy(g['textures_in']).to(dev).contiguous()
            out = lt.load_textures(torch.from_numpy(g['image']).to(dev), faces, tex, torch.from_numpy(g['is_update']).to(dev), w, b)
            assert out is tex                                         # in place, same tensor returned
            assert np.array_equal(tex.cpu().numpy(), g['textures_w%d_b%d' % (w, b)]), (w, b)
            assert np.array_equal(faces.cpu().numpy(), g['faces_w%d' % w]), w
    g = golden('create_texture_image')
    for tag in 'ab':
        want = g['image_' + tag]
        img = torch.zeros(want.shape, device=dev)
        cti.create_texture_image(torch.from_numpy(g['vertices_' + tag]).to(dev), torch.from_numpy(g['textures_' + tag]).to(dev),
                                 img, float(g['eps']))
        assert np.array_equal(img.cpu().numpy(), want), tag
    with pytest.raises(RuntimeError):
        lt.load_textures(torch.zeros(4, 4, 3), torch.zeros(1, 3, 2), torch.zeros(1, 2, 2, 2, 3), torch.zeros(1, dtype=torch.int32), 0, 1)


def test_textured_obj_round_trip(tmp_path):
    """nr.save_obj writes OBJ + MTL + atlas PNG (create_texture_image kernel); nr.load_obj(load_texture=True) reads
    them back through the load_textures kernel: geometry identical, face colours survive the 8-bit atlas."""
    import neural_renderer as nr
    from rnr_amd import scene
    mesh = scene.uv_sphere(6, 8)
    v = torch.from_numpy(mesh['v']).cuda()
    f = torch.from_numpy(mesh['f_v_idx']).cuda()
    nf = f.shape[0]
    g = torch.Generator().manual_seed(3)
    colour = torch.rand(nf, 1, 1, 1, 3, generator=g)
    tex = colour.expand(nf, 4, 4, 4, 3).contiguous().cuda()             # one flat colour per face
    path = str(tmp_path / 'm.obj')
    nr.save_obj(path, v, f, tex)
    assert all((tmp_path / n).exists() for n in ['m.obj', 'm.mtl', 'm.png'])
    v_attr, f_attr, tex2 = nr.load_obj(path, normalization=False, texture_size=4, load_texture=True,
                                       texture_wrapping='CLAMP_TO_EDGE', use_bilinear=False)
    assert torch.allclose(v_attr['v'], v, atol=1e-6)
    assert torch.equal(f_attr['f_v_idx'], f)
    assert tex2.shape == (nf, 4, 4, 4, 3)
    # interior texels of each cube sample inside the face's atlas triangle: the flat colour comes back to 8-bit accuracy
    centre = tex2[:, 1, 1, 1, :].cpu()
    assert (centre - colour[:, 0, 0, 0, :]).abs().max() < 2.5 / 255


def test_renderer_look_at_modes():
    """nr.Renderer(camera_mode='look_at'): silhouettes / depth / rgb modes and the 8-tuple, vs the C oracle rasterizer
    fed with the same look_at + perspective vertices (SSAA off and on)."""
    import neural_renderer as nr
    from oracle import raster as oras
    from rnr_amd import scene
    mesh = scene.uv_sphere(12, 24)
    v = torch.from_numpy(mesh['v'] * 0.6)[None].cuda()
    f = torch.from_numpy(mesh['f_v_idx'])[None].cuda()
    nf = f.shape[1]
    tex = torch.rand(1, nf, 2, 2, 2, 3, device='cuda:0')
    for aa in (False, True):
        r = nr.Renderer(image_size=64, anti_aliasing=aa, camera_mode='look_at', fill_back=False, viewing_angle=30,
                        light_intensity_ambient=1.0, light_intensity_directional=0.0)
        r.eye = nr.get_points_from_angles(2.732, 20.0, 35.0)
        sil = r(v, f, mode='silhouettes')
        dep = r(v, f, mode='depth')
        rgb = r(v, f, tex, mode='rgb')
        S = 128 if aa else 64
        vv = nr.perspective(nr.look_at(v, r.eye), angle=30)
        faces = vv[0][f[0].long()][None].cpu().numpy()
        o = oras.face_index_map(faces, S, 0.1, 100.0)
        alpha = torch.from_numpy((o['face_index_map'] >= 0).astype(np.float32)).flip(1)
        depth = torch.from_numpy(o['depth_map']).flip(1)
        if aa:
            pool = lambda x: torch.nn.functional.avg_pool2d(x[:, None], 2)[:, 0]
            alpha, depth = pool(alpha), pool(depth)
        assert sil.shape == (1, 64, 64) and 0.05 < float(sil.mean()) < 0.8
        assert torch.allclose(sil.cpu(), alpha, atol=1e-6)
        assert torch.allclose(dep.cpu(), depth, rtol=1e-5, atol=1e-4)
        assert rgb.shape == (1, 3, 64, 64) and float((rgb.cpu().sum(1) > 0).float().mean()) > 0.05
        out = r(v, f, tex)
        assert len(out) == 8 and torch.allclose(out[0], rgb) and torch.allclose(out[2], sil)


def test_extension_accepts_float64(golden):
    """rasterize_cuda_kernel.cu:614 dispatches float / double; the drop-in extension accepts float64 buffers (computed by
    the float32 kernels on converted copies, written back in place, inputs untouched) and rejects mixed scalar types."""
    import warnings
    import neural_renderer.cuda.rasterize as ext
    g = golden('raster_soup64')
    S, far = int(g['image_size']), float(g['far'])
    f64 = T(g['faces']).double().to(DEV)
    f64_before = f64.clone()
    B, nf = f64.shape[:2]
    fim = torch.full((B, S, S), -1, dtype=torch.int32, device=DEV)
    wm = torch.zeros(B, S, S, 3, dtype=torch.float64, device=DEV)
    dm = torch.full((B, S, S), far, dtype=torch.float64, device=DEV)
    fivm = torch.zeros(B, S, S, 3, 3, dtype=torch.float64, device=DEV)
    finv = torch.zeros_like(f64)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        ext._warned[0] = False
        out = ext.forward_face_index_map(f64, fim, wm, dm, fivm, finv, S, 0.0, far, 1, 1, 1)
    assert any('float64' in str(x.message) for x in w)
    assert out[1] is wm and out[2] is dm and out[3] is fivm and wm.dtype == torch.float64
    assert torch.equal(f64.view(torch.int64), f64_before.view(torch.int64))      # bitwise: the soup holds NaN / inf vertices
    assert np.array_equal(fim.cpu().numpy(), g['face_index_map'])
    ok = np.isfinite(g['weight_map'])
    assert np.array_equal(wm.cpu().numpy()[ok], g['weight_map'].astype(np.float64)[ok])
    assert np.array_equal(dm.cpu().numpy(), g['depth_map'].astype(np.float64))
    with pytest.raises(RuntimeError):
        ext.forward_face_index_map(f64, fim, wm.float(), dm, fivm, finv, S, 0.0, far, 1, 1, 1)


def test_rasterizer_fill_back(golden, tmp_path):
    """network.Rasterizer with renderer.fill_back = True (renderer.py:209-211, network.py:183-198): every face once more with
    reversed winding, so the far side of the mesh is drawn where the near side is cut away (here: near plane through the
    sphere).  Compared with the oracle run on the explicitly doubled mesh; anti_aliasing raises with the reason."""
    import network
    from oracle import rnr_oracle as orc
    from rnr_amd import scene
    g = golden('rasterizer_module64')
    mesh = {k: g['mesh_' + k] for k in ['v', 'vt', 'vn', 'f_v_idx', 'f_vt_idx', 'f_vn_idx']}
    fp = str(tmp_path / 'm.obj')
    scene.write_obj(fp, mesh)
    ras = network.Rasterizer(fp, 64).to(DEV)
    ras.renderer.fill_back = True
    ras.renderer.near = 2.9                       # camera at radius 3, unit sphere: the near plane cuts the front cap off
    views = scene.spiral_views(64, [11])
    proj, pose = T(views['proj']).to(DEV), T(views['pose']).to(DEV)
    out = ras(proj, pose, None, None, None)
    fim = out[2].cpu()
    nf = mesh['f_v_idx'].shape[0]
    assert int((fim >= nf).sum()) > 50            # back faces (second copy) are visible through the cut
    assert tuple(out[4].shape) == (1, 2 * nf, 3) and tuple(out[7].shape) == (1, 2 * nf, 3, 3)
    rev = lambda a: np.concatenate([a, a[:, ::-1]], 0)
    mesh2 = {k: T(np.ascontiguousarray(v)) for k, v in mesh.items()}
    for k in ['f_v_idx', 'f_vt_idx', 'f_vn_idx']:
        mesh2[k] = T(np.ascontiguousarray(rev(mesh[k])))
    v_ndc = ops_project(ras, proj, pose)
    ref = orc.rasterizer_forward(mesh2, T(views['proj']), T(views['pose']), 64, near=2.9, v_uvz_ndc=v_ndc.cpu())
    assert torch.equal(fim, ref['face_index_map'])
    for i, k in [(0, 'uv_map'), (5, 'normal_map'), (9, 'position_map')]:
        d = (out[i].cpu() - ref[k]).abs()
        if k == 'uv_map':
            d = torch.minimum(d, 1.0 - d)
        assert d.max() < 5e-6, (k, d.max())
    ras.renderer.anti_aliasing = True
    with pytest.raises(NotImplementedError):
        ras(proj, pose, None, None, None)


def ops_project(ras, proj, pose):
    from rnr_amd import ops
    return ops.project_vertices(ras.vertices[0].contiguous(), proj, pose[:, :3, :3].contiguous(), pose[:, :3, 3].contiguous(),
                                ras.img_size)



def test_tbn_map_matmul_is_one_launch_and_equals_torch():
    """test_rnr.py:314's batched product on a get_TBN_map result goes through rnr_tbn_matvec and equals torch's batched GEMM; other
    forms of the product, and every other operation, stay torch's."""
    import render
    g = torch.Generator().manual_seed(11)
    tbn_plain = torch.randn(2, 96, 80, 3, 3, generator=g).to(DEV)
    v = torch.randn(2, 96, 80, 3, generator=g).to(DEV)
    tbn = tbn_plain.as_subclass(render.TBNMap)
    calls = []
    from rnr_amd import ops
    orig = ops.tbn_matvec
    ops.tbn_matvec = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    import warnings
    render.TBNMap.reset_stats()
    try:
        got = torch.matmul(tbn.reshape((-1, 3, 3)).transpose(-2, -1), v.reshape((-1, 3, 1)))           # the script's line
        got_plain_order = torch.matmul(tbn.reshape((-1, 3, 3)), v.reshape((-1, 3, 1)))
        assert len(calls) == 2 and render.TBNMap.stats['hits'] == 2 and render.TBNMap.stats['misses'] == 0
        with warnings.catch_warnings(record=True) as wl:
            warnings.simplefilter('always')
            other = torch.matmul(tbn.reshape((-1, 3, 3)).transpose(-2, -1), torch.cat([v, v], -1).reshape((-1, 3, 2)))     # not the form: torch's
        assert len(calls) == 2
        # the miss is visible: counted, explained, and warned about exactly once per process
        assert render.TBNMap.stats['misses'] == 1 and 'shapes' in render.TBNMap.stats['last_miss']
        assert sum(issubclass(w.category, RuntimeWarning) and 'rnr_tbn_matvec' in str(w.message) for w in wl) == 1
        with warnings.catch_warnings(record=True) as wl2:
            warnings.simplefilter('always')
            torch.matmul(tbn.reshape((-1, 3, 3)).transpose(-2, -1), torch.cat([v, v], -1).reshape((-1, 3, 2)))
        assert render.TBNMap.stats['misses'] == 2 and not wl2
        # opt-outs: the environment switch and get_TBN_map(..., plain=True) leave the product to torch
        os.environ['RNR_TBN_MATMUL'] = '0'
        try:
            off = torch.matmul(tbn.reshape((-1, 3, 3)).transpose(-2, -1), v.reshape((-1, 3, 1)))
        finally:
            del os.environ['RNR_TBN_MATMUL']
        assert len(calls) == 2 and render.TBNMap.stats['misses'] == 2
    finally:
        ops.tbn_matvec = orig
        render.TBNMap.reset_stats()
    assert torch.equal(off, torch.matmul(tbn_plain.reshape((-1, 3, 3)).transpose(-2, -1), v.reshape((-1, 3, 1))))
    want = torch.matmul(tbn_plain.reshape((-1, 3, 3)).transpose(-2, -1), v.reshape((-1, 3, 1)))
    want_plain_order = torch.matmul(tbn_plain.reshape((-1, 3, 3)), v.reshape((-1, 3, 1)))
    want_other = torch.matmul(tbn_plain.reshape((-1, 3, 3)).transpose(-2, -1), torch.cat([v, v], -1).reshape((-1, 3, 2)))
    assert type(got) is torch.Tensor and got.shape == want.shape
    scale = want.abs().max().item()
    assert (got - want).abs().max().item() <= 4e-7 * scale          # three fused multiply-adds against the GEMM's rounding
    assert (got_plain_order - want_plain_order).abs().max().item() <= 4e-7 * scale
    assert torch.equal(other, want_other) and type(other) is torch.Tensor
    # the rest of the script's expression, and what the ray samplers get
    vt = got[..., 0].reshape(v.shape)
    assert vt.shape == v.shape and type(torch.nn.functional.normalize(vt, dim=-1)) is torch.Tensor
    assert type(tbn.reshape((-1, 3, 3))) is render.TBNMap and tbn.data_ptr() == tbn_plain.data_ptr()


def test_ray_sampler_pivot_cache_follows_the_tensor_not_the_address():
    """ops._host_copy (the cached host copy of the pivots buffer, no stream drain per call) is tied to the tensor OBJECT: a
    second sampler whose pivots land at the recycled address of a dropped one, a `.data =` swap and an in-place edit all get
    their own pivots (the address / version key alone returned the dead sampler's directions)."""
    import network
    from rnr_amd import ops
    g = torch.Generator().manual_seed(5)
    tbn = torch.nn.functional.normalize(torch.randn(1, 8, 8, 3, 3, generator=g), dim=-2).to(DEV)
    vt = torch.nn.functional.normalize(torch.randn(1, 8, 8, 3, generator=g), dim=-1).to(DEV)
    alpha = torch.ones(1, 8, 8, 1, device=DEV)

    def direct(pivots_cpu):         # CPU pivots take the uncached path
        return ops.ray_sampler(False, pivots_cpu, tbn, vt, alpha)[0]

    for step_a, step_b in ((5, 10), (10, 5)):
        a = network.RaySampler(6, 2, step_a, mode='diffuse').to(DEV)
        want_a = direct(a.pivots_dir.cpu())
        assert torch.equal(a(tbn, vt, alpha)[0], want_a)
        ptr = a.pivots_dir.data_ptr()
        del a
        b = network.RaySampler(6, 2, step_b, mode='diffuse').to(DEV)     # the caching allocator hands the block out again
        want_b = direct(b.pivots_dir.cpu())
        assert not torch.equal(want_a, want_b)
        got_b = b(tbn, vt, alpha)[0]
        assert torch.equal(got_b, want_b), 'stale pivots (same address: %s)' % (b.pivots_dir.data_ptr() == ptr)
        # .data swap and in-place edit on a live module
        other = network.RaySampler(6, 2, step_a, mode='diffuse').pivots_dir.to(DEV)
        b.pivots_dir.data = other
        assert torch.equal(b(tbn, vt, alpha)[0], direct(other.cpu()))
        b.pivots_dir.mul_(-1.0)
        assert torch.equal(b(tbn, vt, alpha)[0], direct(b.pivots_dir.cpu()))
    assert len(ops._HOST_COPIES) <= 4


def test_evaluate_sh_basis_numpy_contract_and_cached_float32():
    """sph_harm.evaluate_sh_basis keeps the reference's contract (numpy in, float64 numpy out, sph_harm.py:41-71) and answers the
    conversion test_rnr.py:324 applies next from the float32 block that came down with the result: same values bit for bit as
    numpy's own conversion, counted in SHBasisArray.stats, off with RNR_SH_FAST_ASTYPE=0."""
    import sph_harm
    g = torch.Generator().manual_seed(21)
    d = torch.nn.functional.normalize(torch.randn(1, 48, 40, 3, generator=g), dim=-1)
    dn = d.reshape(-1, 3).numpy()
    sph_harm.SHBasisArray.stats.update(fast=0, plain=0)
    b = sph_harm.evaluate_sh_basis(lmax=2, directions=dn)
    assert isinstance(b, np.ndarray) and b.dtype == np.float64 and b.shape == (48 * 40, 9)
    plain = np.asarray(b).astype(np.float32)                            # numpy's conversion of the float64 container
    fast = b.reshape((1, 48, 40, -1)).astype(np.float32)                # the script's expression
    assert sph_harm.SHBasisArray.stats['fast'] == 1
    assert fast.dtype == np.float32 and np.array_equal(fast.reshape(-1, 9), plain)
    on_dev = sph_harm.evaluate_sh_basis(lmax=2, directions=d.reshape(-1, 3).to(DEV), as_tensor=True)
    assert torch.equal(torch.from_numpy(fast).to(DEV).reshape(-1, 9), on_dev)
    os.environ['RNR_SH_FAST_ASTYPE'] = '0'
    try:
        b2 = sph_harm.evaluate_sh_basis(lmax=2, directions=dn)
        assert type(b2) is np.ndarray and b2.flags.writeable and np.array_equal(b2, np.asarray(b))
    finally:
        del os.environ['RNR_SH_FAST_ASTYPE']
