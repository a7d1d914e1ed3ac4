"""-m gpu: shading kernels vs the oracle / the reference golden vectors.
Tolerances (float32 path): 2e-5 abs on O(1) quantities; integer tap indices bit-exact."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def T(x):
    return torch.from_numpy(np.ascontiguousarray(x))


def test_interpolate_bilinear_golden(golden):
    from oracle import rnr_oracle as orc
    from rnr_amd import ops
    g = golden('bilinear')
    out, taps = ops.interpolate_bilinear(T(g['data']).to(DEV), T(g['x']).to(DEV), T(g['y']).to(DEV), want_taps=True)
    assert torch.allclose(out.cpu(), T(g['out']), atol=1e-6)
    (x0, y0, x1, y1), _ = orc.bilinear_taps(g['data'].shape[0], g['data'].shape[1], T(g['x']), T(g['y']))
    ref = torch.stack([x0, y0, x1, y1], -1).int()
    assert torch.equal(taps.cpu(), ref)          # integer texel indices: bit-exact


def test_sh_basis_vs_oracle():
    from oracle import rnr_oracle as orc
    from rnr_amd import ops, scene
    d = scene.sphere_samples(4096)
    d[:5] = [[0, 0, 1], [0, 0, -1], [1, 0, 0], [0, 2, 0], [3, 4, 12]]      # poles, axes, non-unit
    for lmax in (2, 10):
        got = ops.sh_basis(T(d).to(DEV), lmax).cpu().numpy()
        ref = orc.sh_basis(lmax, d)
        assert np.abs(got - ref).max() < 2e-6, (lmax, np.abs(got - ref).max())


def test_sh_linear_golden(golden):
    from rnr_amd import ops
    g = golden('sh_linear')
    rec = ops.sh_reconstruct(T(g['basis']).to(DEV), T(g['coeff'][0]).to(DEV))
    assert torch.allclose(rec.cpu(), T(g['recon2']), atol=1e-5)
    fit = ops.sh_fit(T(g['samples'][0]).to(DEV), T(g['basis']).to(DEV))
    assert torch.allclose(fit.cpu(), T(g['fit2']), atol=1e-5)


def test_projection_golden(golden):
    """Tolerance: 2e-5 relative on well-conditioned vertices (inside ~2x the field of view).  Vertices far outside it
    go through the r^6 distortion polynomial with catastrophic cancellation; there only 1e-3 relative is asked."""
    from rnr_amd import ops
    g = golden('projection')
    for b in range(g['vertices'].shape[0]):      # the kernel takes one shared mesh; run each batch element
        a = lambda k: T(g[k][b:b + 1]).to(DEV)
        out = ops.project_vertices(T(g['vertices'][b]).to(DEV), a('K'), a('R'), T(g['t'][b]).to(DEV), int(g['orig_size']))
        ref = T(g['out_nodist'][b])
        sane = (ref[:, :2].abs().max(-1)[0] < 4.0) & (ref[:, 2].abs() > 0.2)   # x/(z+eps) is ill-conditioned near z = 0
        assert torch.allclose(out.cpu()[0][sane], ref[sane], atol=2e-5, rtol=2e-5)
        assert torch.allclose(out.cpu()[0][~sane], ref[~sane], atol=1e-3, rtol=1e-3)
        out = ops.project_vertices(T(g['vertices'][b]).to(DEV), a('K'), a('R'), T(g['t'][b]).to(DEV), int(g['orig_size']),
                                   dist_coeffs=a('dist'), offset=a('offset'), scale=a('scale'))
        ref_d = T(g['out_dist'][b])
        assert sane.sum() >= 8
        assert torch.allclose(out.cpu()[0][sane], ref_d[sane], atol=5e-5, rtol=2e-5)
        assert torch.allclose(out.cpu()[0][~sane], ref_d[~sane], atol=1e-3, rtol=1e-3)


def test_projection_well_conditioned_vs_oracle():
    """All vertices in front of the camera (the hot-path situation): 5e-6 abs on NDC coordinates in [-1,1]."""
    from oracle import rnr_oracle as orc
    from rnr_amd import ops, scene
    mesh = scene.uv_sphere(32, 64)
    v = scene.spiral_views(512, [0, 123, 700])
    proj, pose = T(v['proj']), T(v['pose'])
    ref = orc.projection(T(mesh['v'])[None].expand(3, -1, -1), proj, pose[:, :3, :3], pose[:, :3, 3][:, None, :],
                         torch.zeros(1, 5), 512)
    out = ops.project_vertices(T(mesh['v']).to(DEV), proj.to(DEV), pose[:, :3, :3].contiguous().to(DEV),
                               pose[:, :3, 3].contiguous().to(DEV), 512)
    assert (out.cpu() - ref).abs().max() < 5e-6


def _mesh_from_golden(gm):
    from rnr_amd import ops
    return ops.DeviceMesh(gm['buf_vertices'][0], gm['mesh_vt'], gm['buf_vertices_normals'][0], gm['mesh_f_v_idx'],
                          gm['mesh_f_vt_idx'], gm['mesh_f_vn_idx'], DEV)


def test_shade_inputs_vs_reference_frame(golden):
    """G-buffer (reference's own) -> network input, compared with the reference's assembled render_net_input
    (frame64, stored as fp16) and, tighter, with the oracle."""
    from oracle import rnr_oracle as orc
    from rnr_amd import ops, testing
    gm, gf, gs = golden('rasterizer_module64'), golden('frame64'), golden('shading_geometry64')
    mesh = _mesh_from_golden(gm)
    tex = [T(gf['tex%d' % i]).to(DEV) for i in range(4)]
    ps, pd = testing.ray_pivots(6, 2, 5), testing.ray_pivots(6, 2, 10)
    for i in range(2):
        gb = {'face_index_map': T(gm['view%d_face_index_map' % i]).to(DEV), 'alpha': T(gm['view%d_alpha' % i]).to(DEV),
              'uv_map': T(gm['view%d_uv_map' % i]).to(DEV), 'normal_map': T(gm['view%d_normal_map' % i]).to(DEV)}
        out = ops.shade_inputs(gb, mesh, T(gm['proj_inv'][i:i + 1]).to(DEV), T(gm['R_inv'][i:i + 1]).to(DEV), tex, ps, pd,
                               6, want_rays_uv=True, want_neural_img=True, want_sh=True)
        c_in = 26 * 3 + 6 + 16
        net_in = out['net_in'][..., :c_in].permute(0, 3, 1, 2).cpu()
        assert float(out['net_in'][..., c_in:].abs().max()) == 0.0
        ref16 = T(gf['net_in'][i:i + 1]).float()
        assert torch.allclose(net_in, ref16, atol=2e-3)
        # oracle at full precision
        gbo = {k: v.cpu() for k, v in gb.items()}
        gbo['faces_v'] = T(gm['view%d_faces_v' % i])
        gbo['faces_vt'] = T(gm['view%d_faces_vt' % i])
        o = orc.shade_inputs(gbo, T(gm['proj_inv'][i:i + 1]), T(gm['R_inv'][i:i + 1]), [t.cpu() for t in tex], ps, pd)
        assert torch.allclose(net_in, o['net_in'], atol=2e-5), (net_in - o['net_in']).abs().max()
        assert torch.allclose(out['sh_basis_map'].cpu(), o['sh_basis_map'], atol=2e-6)
        assert torch.allclose(out['neural_img'].cpu(), o['neural_img'], atol=2e-5)
        assert torch.allclose(out['rays_uv'].cpu(), o['rays_uv'], atol=2e-5)
        if i == 0:   # reference-generated geometry of view 0
            assert torch.allclose(out['rays_uv'].cpu()[..., :13], T(gs['rays_uv_spec']), atol=2e-5)
            assert torch.allclose(out['rays_uv'].cpu()[..., 13:], T(gs['rays_uv_diff']), atol=2e-5)


def test_ray_render_golden(golden):
    """rnr_ray_render consumes ray DIRECTIONS (from net_in) and the raw out-layer output; build both from the golden
    rays_uv / rays_lt by inverting the mappings, then compare with the reference RayRenderer output."""
    from oracle import rnr_oracle as orc
    from rnr_amd import ops
    g = golden('ray_renderer')
    uv, lt = T(g['rays_uv']), T(g['rays_lt'])               # [N,H,W,2,26], [N,26,3,H,W]
    N, H, W, _, R = uv.shape
    alpha = (uv[..., 0, 0] >= 0).float()                    # background pixels carry uv = -1
    uvc = uv.clamp(min=1e-4, max=1 - 1e-4)
    dirs = orc.spherical_mapping_inv(uvc.permute(3, 0, 1, 2, 4).reshape(2, -1)).reshape(3, N, H, W, R)
    uv_back = orc.spherical_mapping(dirs, dim=0).permute(1, 2, 3, 0, 4) * alpha[..., None, None] - (alpha[..., None, None] == 0).float()
    c_pad = 112
    net_in = torch.zeros(N, H, W, c_pad)
    net_in[..., :3 * R] = dirs.permute(1, 2, 3, 4, 0).reshape(N, H, W, 3 * R)
    net_in[..., 84:87] = T(g['albedo_diffuse']).permute(0, 2, 3, 1)
    net_in[..., 87:90] = T(g['albedo_specular']).permute(0, 2, 3, 1)
    # rays_lt = (tanh(raw + bias)*0.5 + 0.5)*2  =>  raw = atanh(lt - 1) - bias
    bias = torch.linspace(-0.2, 0.2, 80)
    y = (lt - 1.0).clamp(-0.999, 0.999)
    lt_eff = (y * 0.5 + 0.5) * 2.0
    raw = torch.zeros(N, H, W, 80)
    raw[..., :78] = torch.atanh(y).permute(0, 3, 4, 1, 2).reshape(N, H, W, 78) - bias[:78]
    img = ops.ray_render(raw.to(DEV), bias.to(DEV), net_in.to(DEV), alpha.to(DEV), T(g['lp']).to(DEV), 13, 13)
    ref = orc.ray_renderer(T(g['albedo_specular']), uv_back, lt_eff, T(g['lp']), albedo_diffuse=T(g['albedo_diffuse']),
                           num_ray_diffuse=13, seperate_albedo=True)[0]
    assert torch.allclose(img.cpu(), ref, atol=5e-4, rtol=1e-4), (img.cpu() - ref).abs().max()
    # and the reference's own output where the uv round trip is harmless (interior uv only)
    interior = ((uv > 2e-4) & (uv < 1 - 2e-4)).all(-1).all(-1) & (lt.permute(0, 3, 4, 1, 2).reshape(N, H, W, -1) < 1.99).all(-1) \
        & (lt.permute(0, 3, 4, 1, 2).reshape(N, H, W, -1) > 0.01).all(-1)
    d = (img.cpu() - T(g['out'])).abs().permute(0, 2, 3, 1)[interior]
    assert d.numel() > 0 and d.max() < 2e-3, d.max()


def test_lighting_front_end_vs_oracle():
    """Env map -> 4096 bilinear samples -> SH (lmax 10) projection -> 100x200 reconstruction (config 5's lighting
    path; network.py:665-672, 694-699, 622-627) vs the oracle."""
    from oracle import rnr_oracle as orc
    from rnr_amd import lighting, scene, testing
    env = testing.synthetic_light_probe(160, 320, 5)[0]
    l_dir = T(scene.sphere_samples(4096)).t().contiguous()
    coeff, samples, basis = lighting.envmap_to_sh(env.to(DEV), l_dir.to(DEV), 10)
    uv = orc.spherical_mapping(l_dir)
    ref_samples = orc.interpolate_bilinear(env, (uv[0] * 320.0).clamp(max=319), (uv[1] * 160.0).clamp(max=159))
    assert torch.allclose(samples.cpu(), ref_samples, atol=2e-5)
    ref_basis = torch.from_numpy(orc.sh_basis(10, l_dir.t().numpy()).astype(np.float32))
    ref_coeff = orc.fit_sh_coeff(ref_samples, ref_basis)
    assert torch.allclose(coeff.cpu(), ref_coeff, atol=2e-4)
    shl = lighting.SHLighting(10, DEV)
    lp = shl.light_probe(coeff)
    ref_lp = orc.reconstruct_lp(ref_coeff, torch.from_numpy(orc.sh_basis(10, orc.lp_recon_dirs().numpy()).astype(np.float32)))
    assert lp.shape == (100, 200, 3)
    assert torch.allclose(lp.cpu(), ref_lp, atol=5e-4)


@pytest.mark.parametrize('n_views', [1, 3])
def test_frame_prepare_equals_the_separate_launches(n_views):
    """rnr_frame_prepare (r04: projection from the [N,4,4] poses + per-face tangents + SH light probe + workspace clearing in
    ONE launch) against the stand-alone entry points: every output bit-identical, and rnr_rasterize_gbuffer_prepared on the
    workspace it cleared gives the maps of rnr_rasterize_gbuffer — also when the workspace is dirty from a previous call
    with another view count (the layout depends on it)."""
    from rnr_amd import ops, scene
    from rnr_amd.lighting import SHLighting
    S = 128
    m = scene.uv_sphere(24, 48)
    mesh = ops.DeviceMesh(m['v'], m['vt'], m['vn'], m['f_v_idx'], m['f_vt_idx'], m['f_vn_idx'], DEV)
    views = {k: torch.from_numpy(v).to(DEV) for k, v in scene.spiral_views(S, [5, 300, 650][:n_views]).items()}
    K, pose = views['proj'].contiguous(), views['pose'].contiguous()
    shl = SHLighting(10, torch.device(DEV))
    coeff = torch.from_numpy(scene.synthetic_sh_coeff(2, 10, 1)).to(DEV)[1].contiguous()
    # separate launches
    v_ref = ops.project_vertices(mesh.v, K, pose[:, :3, :3].contiguous(), pose[:, :3, 3].contiguous(), S)
    t_ref = mesh.tangents().clone()
    lp_ref = shl.light_probe(coeff)
    maps = ['face_index_map', 'alpha', 'uv_map', 'normal_map']
    gb_ref = ops.rasterize_gbuffer(mesh, v_ref, None, S, maps=maps)
    # one launch
    L = ops._lib.load()
    ws = torch.full((L.rnr_gbuffer_workspace_bytes(3, mesh.num_faces, S),), 0x5A, dtype=torch.uint8, device=DEV)   # dirty, sized for 3 views
    v = torch.empty(n_views, mesh.num_vertices, 3, device=DEV)
    t = torch.empty(mesh.num_faces, 3, device=DEV)
    lp = torch.empty(shl.h, shl.w, 3, device=DEV)
    ops.frame_prepare(mesh, K, pose, S, v_uvz=v, tangents=t, lp_basis=shl.basis_recon, lp_coeff=coeff, light_probe=lp, workspace=ws)
    assert torch.equal(v, v_ref) and torch.equal(t, t_ref) and torch.equal(lp.reshape(lp_ref.shape), lp_ref)
    gb = ops.rasterize_gbuffer(mesh, v, None, S, maps=maps, workspace=ws, prepared=True)
    for k in maps:
        assert torch.equal(gb[k], gb_ref[k]), k
    assert int((gb['face_index_map'] >= 0).sum()) > 1000
    # parts are optional: projection only
    v2 = torch.zeros_like(v)
    ops.frame_prepare(mesh, K, pose, S, v_uvz=v2)
    assert torch.equal(v2, v_ref)
    with pytest.raises(ValueError):
        ops.frame_prepare(mesh, K, pose, S, v_uvz=v[:, :-1].contiguous())


def test_ray_render_background_workgroups_write_exact_zeros():
    """r04: a ray_render workgroup whose 32 pixels are all background returns after writing zeros, without reading unet_raw /
    net_in.  The frame must not depend on what those buffers hold on background pixels (NaN here), must be exactly 0 there, and
    must equal the frame computed with finite garbage in their place — for fully covered, fully empty and mixed 32-pixel runs,
    a ragged last workgroup included."""
    from rnr_amd import ops
    g = torch.Generator().manual_seed(77)
    N, H, W, R = 2, 24, 44, 13                      # 2112 pixels = 66 workgroups of 32 pixels; view boundary inside a workgroup
    c_pad, c_out_pad = 112, 80
    dirs = torch.nn.functional.normalize(torch.randn(N, H, W, 2 * R, 3, generator=g), dim=-1)
    net_in = torch.zeros(N, H, W, c_pad)
    net_in[..., :6 * R] = dirs.reshape(N, H, W, 6 * R)
    net_in[..., 6 * R + 6:6 * R + 12] = torch.rand(N, H, W, 6, generator=g)
    raw = torch.randn(N, H, W, c_out_pad, generator=g)
    bias = torch.randn(c_out_pad, generator=g) * 0.1
    lp = torch.rand(20, 40, 3, generator=g)
    alpha = torch.zeros(N * H * W)
    alpha[100:700] = 1.0                            # whole workgroups inside, partial ones at both ends
    alpha[1000:1003] = 1.0                          # three foreground pixels in one workgroup
    alpha[2090:] = 1.0                              # up to the last pixel of the batch
    alpha = alpha.reshape(N, H, W)
    bg = alpha == 0
    ref = ops.ray_render(raw.to(DEV), bias.to(DEV), net_in.to(DEV), alpha.to(DEV), lp.to(DEV), R, R).clone()
    raw_p, ni_p = raw.clone(), net_in.clone()
    raw_p[bg] = float('nan')                        # the out layer may have skipped the tile: garbage on every background pixel
    # net_in is written by shade_inputs on every pixel (finite); only workgroups that are background throughout never read it
    runs = bg.reshape(-1, 32).all(dim=1, keepdim=True).expand(-1, 32).reshape(N, H, W)
    ni_p[runs] = float('nan')
    assert int(runs.sum()) > 1000 and int((bg & ~runs).sum()) > 20
    out = ops.ray_render(raw_p.to(DEV), bias.to(DEV), ni_p.to(DEV), alpha.to(DEV), lp.to(DEV), R, R)
    assert torch.isfinite(out).all() and torch.equal(out, ref)
    assert float(out.permute(0, 2, 3, 1)[bg.to(DEV)].abs().max()) == 0.0
    assert float(out.permute(0, 2, 3, 1)[(~bg).to(DEV)].abs().max()) > 0.05


@pytest.mark.parametrize('n,c,h,w,c_pad', [(1, 108, 64, 64, 112), (3, 30, 20, 13, 32), (2, 78, 16, 24, 80), (1, 3, 7, 9, 16), (2, 250, 8, 8, 256)])
def test_layout_ops_vs_torch_permute(n, c, h, w, c_pad):
    """rnr_nchw_to_nhwc / rnr_nhwc_to_nchw (the layout changes around the drop-in RenderingNet; LDS-tiled since r05, the
    one-element-per-thread form for > 240 channels): exact copies, zero padding channels, bias + tanh on the way back; views
    whose pixel count is not a multiple of the 64-pixel tile, tiles that straddle two views."""
    from rnr_amd import ops
    g = torch.Generator().manual_seed(n * 1000 + c)
    x = torch.randn(n, c, h, w, generator=g)
    y = ops.nchw_to_nhwc(x.to(DEV), c_pad).cpu()
    assert tuple(y.shape) == (n, h, w, c_pad)
    assert torch.equal(y[..., :c], x.permute(0, 2, 3, 1)) and float(y[..., c:].abs().sum()) == 0.0
    raw = torch.randn(n, h, w, c_pad, generator=g)
    b = torch.randn(c_pad, generator=g)
    z = ops.nhwc_to_nchw(raw.to(DEV), c, bias=b.to(DEV), apply_tanh=True).cpu()
    ref = torch.tanh(raw[..., :c] + b[:c]).permute(0, 3, 1, 2)
    assert tuple(z.shape) == (n, c, h, w) and float((z - ref).abs().max()) <= 2e-6
    z0 = ops.nhwc_to_nchw(raw.to(DEV), c).cpu()
    assert torch.equal(z0, raw[..., :c].permute(0, 3, 1, 2))
