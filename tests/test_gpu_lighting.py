"""-m gpu: the lighting front-end and the remaining drop-in classes on the HIP operators:
area resize (cv2 INTER_AREA restated), network.LightingLP (probe -> 1600x3200 -> 4096 samples -> SH fit), network.Mesh,
whole-batch BatchNorm of the drop-in Unet, operator launches on the device of their tensors."""
import os
import sys
import tempfile

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def T(x):
    return torch.from_numpy(np.ascontiguousarray(x))


@pytest.mark.parametrize('src,dst', [((24, 40), (6, 10)), ((24, 40), (7, 11)), ((64, 128), (23, 57)), ((50, 100), (50, 100)),
                                     ((16, 32), (48, 80)), ((40, 40), (10, 64)), ((1, 9), (1, 2))])
def test_resize_area_vs_oracle(src, dst):
    """rnr_resize_area vs the oracle restatement of cv2 INTER_AREA (shrinking: fractional box filter; otherwise the
    area-mode bilinear).  Tolerance 2e-6: same float32 weights, different summation order."""
    from oracle import rnr_oracle as orc
    from rnr_amd import ops
    rng = np.random.RandomState(src[0] * 7 + dst[1])
    img = (rng.rand(src[0], src[1], 3) * 5).astype(np.float32)
    out = ops.resize_area(T(img).to(DEV), dst[0], dst[1]).cpu().numpy()
    ref = orc.resize_area(img, dst[0], dst[1])
    assert out.shape == ref.shape
    assert np.abs(out - ref).max() <= 2e-6 * 5, np.abs(out - ref).max()
    if src[0] % dst[0] == 0 and src[1] % dst[1] == 0:
        fy, fx = src[0] // dst[0], src[1] // dst[1]
        box = img.reshape(dst[0], fy, dst[1], fx, 3).mean((1, 3))
        assert np.abs(out - box).max() <= 1e-5


def test_lighting_lp_class_vs_oracle():
    """network.LightingLP (network.py:631-699) end to end: two synthetic probes of different sizes -> `lps`
    [2,160,320,3] (small lp_img_h/w keeps the oracle quick; the 1600x3200 default is exercised below), `l_samples`
    [2,4096,3] and `sh_coeff` [2,121,3] vs the oracle's resize + bilinear + SH fit."""
    import network
    from oracle import rnr_oracle as orc
    from rnr_amd import scene, testing
    l_dir = T(scene.sphere_samples(4096)).t().contiguous()
    probes = [testing.synthetic_light_probe(400, 800, 3)[0], testing.synthetic_light_probe(333, 640, 4)[0]]
    loader = [{'lp_img': p.permute(2, 0, 1)[None]} for p in probes]
    lp = network.LightingLP(l_dir, num_channel=3, lp_dataloader=loader, fix_params=True, lp_img_h=160, lp_img_w=320, device=DEV)
    assert lp.num_lighting == 2 and tuple(lp.lps.shape) == (2, 160, 320, 3) and tuple(lp.l_samples.shape) == (2, 4096, 3)
    assert not lp.l_samples.requires_grad
    lp.fit_sh(lmax=10)
    assert tuple(lp.sh_coeff.shape) == (2, 121, 3)
    uv = orc.spherical_mapping(l_dir)
    basis = torch.from_numpy(orc.sh_basis(10, l_dir.t().numpy()).astype(np.float32))
    for i, p in enumerate(probes):
        small = T(orc.resize_area(p.numpy(), 160, 320))
        assert (lp.lps[i] - small).abs().max() < 1e-5
        x = (uv[0] * 320.0).clamp(max=319)
        y = (uv[1] * 160.0).clamp(max=159)
        samples = orc.interpolate_bilinear(small, x, y)
        assert (lp.l_samples[i].cpu() - samples).abs().max() < 2e-5
        coeff = orc.fit_sh_coeff(samples, basis)
        assert (lp.sh_coeff[i].cpu() - coeff).abs().max() < 2e-5
    assert tuple(lp(0, is_lp=True).shape) == (1, 160, 320, 3) and tuple(lp(None).shape) == (1, 2, 4096, 3)
    # the chain test_rnr.py:153-157 builds on top: LightingSH initialised from the fitted coefficients
    sh = network.LightingSH(l_dir, lmax=10, num_lighting=lp.num_lighting, num_channel=3, init_coeff=lp.sh_coeff,
                            fix_params=True).to(DEV)            # test_rnr.py:162 moves it to the device
    probe = sh(1, is_lp=True)
    assert tuple(probe.shape) == (1, 100, 200, 3) and torch.isfinite(probe).all()
    # the low-order reconstruction of a smooth probe stays close to the probe itself
    ref = T(orc.resize_area(probes[1].numpy(), 100, 200))
    assert (probe[0].cpu() - ref).abs().mean() < 0.15 * ref.abs().mean()


def test_lighting_lp_default_1600x3200():
    """Config 5's shape: a 2048x4096 probe area-averaged to 1600x3200 (ratio 1.28: fractional cells), 4096 samples,
    lmax 10.  Checked through size-independent properties: mean preserved by the resize, finite samples inside the
    probe's range, l = 0 coefficient = sqrt(4 pi) x mean radiance over the sphere samples."""
    import network
    from rnr_amd import scene, testing
    l_dir = T(scene.sphere_samples(4096)).t().contiguous()
    probe = testing.synthetic_light_probe(2048, 4096, 5)[0]
    lp = network.LightingLP(l_dir, lp_dataloader=[{'lp_img': probe.permute(2, 0, 1)[None]}], fix_params=True, device=DEV)
    assert tuple(lp.lps.shape) == (1, 1600, 3200, 3)
    assert abs(float(lp.lps.mean()) - float(probe.mean())) < 1e-4
    assert float(lp.l_samples.min()) >= float(probe.min()) - 1e-5 and float(lp.l_samples.max()) <= float(probe.max()) + 1e-5
    lp.fit_sh(10)
    c0 = lp.sh_coeff[0, 0]
    want = lp.l_samples[0].mean(0) * (4 * np.pi) * 0.28209479177387814
    assert torch.allclose(c0.cpu(), want.cpu(), rtol=1e-4)


def test_mesh_module(golden):
    import network
    from rnr_amd import scene
    g = golden('rasterizer_module64')
    mesh = {k: g['mesh_' + k] for k in ['v', 'vt', 'vn', 'f_v_idx', 'f_vt_idx', 'f_vn_idx']}
    tmp = tempfile.mkdtemp(prefix='rnr_mesh_')
    fp = os.path.join(tmp, 'm.obj')
    scene.write_obj(fp, mesh)
    m = network.Mesh(fp, global_RT=T(g['global_RT']))
    assert m.num_vertex == mesh['v'].shape[0]
    assert torch.allclose(m.v, T(g['buf_vertices'])[0], atol=1e-6) and torch.allclose(m.vn, T(g['buf_vertices_normals'])[0], atol=1e-6)
    assert torch.allclose(m.v_orig, T(mesh['v']), atol=1e-6)
    assert float(m.span_max) == float((m.v.max(0)[0] - m.v.min(0)[0]).max())
    assert set(m.state_dict().keys()) == {'v', 'vn'}


def test_dropin_unet_whole_batch_batchnorm():
    """torch semantics for N > 1: train-mode BatchNorm2d reduces over the whole batch of the call.  The drop-in
    RenderingNet must equal a plain torch evaluation of the same live path with F.batch_norm(training=True) on the full
    batch, update running_mean / running_var / num_batches_tracked of the live layers like torch does, and see
    in-place weight edits."""
    import network
    import torch.nn.functional as F
    torch.manual_seed(0)
    net = network.RenderingNet(nf0=8, in_channels=12, out_channels=6, num_down_unet=5, use_gcn=False).to(DEV)
    g = torch.Generator().manual_seed(1)
    for k, p in net.named_parameters():
        if p.dim() == 1:
            p.data = ((torch.rand(p.shape, generator=g) - 0.5) * 0.5 + (0.0 if k.endswith('bias') else 1.0)).to(DEV)
    net.eval()
    for mod in net.modules():
        if type(mod) == torch.nn.BatchNorm2d:
            mod.train()
    x = torch.randn(3, 12, 64, 64, generator=g)
    sd0 = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    y = net(x.to(DEV), None).cpu()

    def ref_forward(sd, x):
        w = lambda k: sd['net.' + k].double()
        run = {}

        def bn(h, key):
            m = h.mean((0, 2, 3)); v = h.var((0, 2, 3), unbiased=False)
            n = h.numel() / h.shape[1]
            run[key] = (m, v * n / (n - 1))
            hn = (h - m[None, :, None, None]) / torch.sqrt(v[None, :, None, None] + 1e-5)
            return hn * w(key + '.weight')[None, :, None, None] + w(key + '.bias')[None, :, None, None]
        cr = lambda h, k, s=1, b=None: F.conv2d(F.pad(h, (1, 1, 1, 1), mode='reflect'), w(k), b, stride=s)

        def block(h, path, depth):
            d, u = path + 'down.net.', path + 'up.net.'
            if depth == 4:
                t = F.leaky_relu(cr(h, d + '1.weight', 1, w(d + '1.bias')), 0.2)
                t = F.leaky_relu(cr(t, d + '5.weight', 2, w(d + '5.bias')), 0.2)
                t = F.relu(F.conv_transpose2d(t, w(u + '0.weight'), w(u + '0.bias'), stride=2, padding=1))
                t = F.relu(cr(t, u + '3.net.1.weight', 1, w(u + '3.net.1.bias')))
            else:
                t = F.leaky_relu(bn(cr(h, d + '1.weight'), d + '2'), 0.2)
                t = F.leaky_relu(bn(cr(t, d + '6.weight', 2), d + '7'), 0.2)
                t = block(t, path + 'submodule.', depth + 1)
                t = F.relu(bn(F.conv_transpose2d(t, w(u + '0.weight'), None, stride=2, padding=1), u + '1'))
                t = F.relu(bn(cr(t, u + '4.net.1.weight'), u + '5'))
            return torch.cat([h, t], 1)
        h = F.leaky_relu(bn(cr(x.double(), 'in_layer.0.net.1.weight'), 'in_layer.1'), 0.2)
        h = block(h, 'unet_block.', 0)
        return torch.tanh(cr(h, 'out_layer.0.net.1.weight', 1, w('out_layer.0.net.1.bias'))), run
    ref, run = ref_forward(sd0, x)
    assert (y.double() - ref).abs().max() < 3e-4, (y.double() - ref).abs().max()
    # a batch is NOT three independent views here (that is RNRPipeline's per-view mode)
    y1 = net(x[:1].to(DEV), None).cpu()
    assert (y1 - y[:1]).abs().max() > 1e-4
    sd1 = net.state_dict()
    for key, (m, v) in run.items():
        rm0, rv0 = sd0['net.' + key + '.running_mean'].double(), sd0['net.' + key + '.running_var'].double()
        # two forwards ran (N = 3, then N = 1): check the first update through num_batches_tracked and the second one
        # only for consistency of the count
        assert int(sd1['net.' + key + '.num_batches_tracked']) == int(sd0['net.' + key + '.num_batches_tracked']) + 2
    # running statistics after exactly ONE forward on a fresh copy
    net2 = network.RenderingNet(nf0=8, in_channels=12, out_channels=6, num_down_unet=5, use_gcn=False).to(DEV)
    net2.load_state_dict(sd0)
    net2.eval()
    for mod in net2.modules():
        if type(mod) == torch.nn.BatchNorm2d:
            mod.train()
    net2(x.to(DEV), None)
    sd2 = net2.state_dict()
    for key, (m, v) in run.items():
        rm0, rv0 = sd0['net.' + key + '.running_mean'].double(), sd0['net.' + key + '.running_var'].double()
        assert torch.allclose(sd2['net.' + key + '.running_mean'].cpu().double(), 0.9 * rm0 + 0.1 * m, atol=1e-5), key
        assert torch.allclose(sd2['net.' + key + '.running_var'].cpu().double(), 0.9 * rv0 + 0.1 * v, rtol=1e-4, atol=1e-6), key
    # in-place weight edit must be seen by the next forward (cached packed weights are keyed on tensor versions)
    with torch.no_grad():
        net2.net.out_layer[0].net[1].bias.add_(0.25)
    sd3 = {k: v.detach().cpu().clone() for k, v in net2.state_dict().items()}
    y3 = net2(x.to(DEV), None).cpu()
    ref3, _ = ref_forward(sd3, x)
    assert (y3.double() - ref3).abs().max() < 3e-4
    # eval-mode BatchNorm uses the (updated) running statistics
    net2.eval()
    ye = net2(x.to(DEV), None).cpu()
    assert torch.isfinite(ye).all() and (ye - y3).abs().max() > 1e-4


def test_ops_follow_tensor_device_not_current_device():
    """Operators must launch on the device / stream of their tensors: with a single GPU this checks the
    bookkeeping — mixed-device arguments raise, `device` index-less tensors work, sph_harm follows its input."""
    import sph_harm
    from rnr_amd import ops
    d = torch.nn.functional.normalize(torch.randn(64, 3), dim=-1)
    a = ops.sh_basis(d.to('cuda'), 3)
    b = ops.sh_basis(d.to('cuda:0'), 3)
    assert torch.equal(a, b) and a.device == torch.device('cuda:0')
    out = sph_harm.evaluate_sh_basis(lmax=3, directions=d.to(DEV))
    assert np.abs(out - a.cpu().numpy()).max() < 1e-6
    if torch.cuda.device_count() > 1:
        with pytest.raises(RuntimeError):
            ops.sh_reconstruct(a, torch.zeros(16, 3, device='cuda:1'))
        with torch.cuda.device(0):
            c = ops.sh_basis(d.to('cuda:1'), 3)         # current device 0, tensors on 1
        assert c.device == torch.device('cuda:1') and torch.allclose(c.cpu(), a.cpu(), atol=1e-6)
