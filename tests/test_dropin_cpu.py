"""CPU: the drop-in modules expose the reference's names and checkpoint layout (no kernel launches)."""
import json
import os

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope='module')
def ref_keys():
    return json.load(open(os.path.join(HERE, 'golden', 'state_dict_keys.json')))


def _shapes(mod):
    return {k: list(v.shape) for k, v in mod.state_dict().items()}


def test_state_dict_layout_matches_reference(ref_keys):
    import network
    assert _shapes(network.RenderingNet(64, 108, 78, 5, 512)) == ref_keys['RenderingNet(64,108,78,5,512)']
    assert _shapes(network.RenderingNet(80, 30, 3, 5, use_gcn=False)) == ref_keys['RenderingNet(80,30,3,5,use_gcn=False)']
    assert _shapes(network.TextureMapper(512, 24, 4, apply_sh=True)) == ref_keys['TextureMapper(512,24,4)']
    assert _shapes(network.RaySampler(6, 2, 5)) == ref_keys['RaySampler(6,2,5)']


def test_strict_load_of_reference_checkpoint(golden):
    """A state-dict produced by the reference's RenderingNet loads with strict=True (running stats filled in)."""
    import network
    g = golden('unet_nf4')
    net = network.RenderingNet(nf0=4, in_channels=10, out_channels=6, num_down_unet=5, out_channels_gcn=16)
    sd = net.state_dict()
    for k in g.files:
        if k.startswith('sd:'):
            assert k[3:] in sd, k
            sd[k[3:]] = torch.from_numpy(g[k])
    net.load_state_dict(sd, strict=True)
    assert torch.equal(net.net.in_layer[0].weight, net.net.in_layer[0].net[1].weight)       # alias kept
    assert net.net.out_layer_weight is net.net.out_layer[0].weight


def test_neural_renderer_api_names():
    import neural_renderer as nr
    for name in ['load_obj', 'projection', 'lighting', 'vertices_to_faces', 'vertex_attrs_to_faces', 'rasterize_rgbad',
                 'rasterize', 'rasterize_silhouettes', 'rasterize_depth', 'Rasterize', 'Renderer', 'look', 'look_at',
                 'perspective', 'save_obj', 'Mesh', 'get_points_from_angles']:
        assert hasattr(nr, name), name
    import neural_renderer.cuda.rasterize as ext
    for name in ['forward_face_index_map', 'forward_texture_sampling', 'backward_pixel_map', 'backward_textures',
                 'backward_depth_map']:
        assert hasattr(ext, name)
    with pytest.raises(RuntimeError):          # real HIP kernels now: CPU tensors are rejected like CHECK_INPUT does
        ext.backward_textures(torch.zeros(1, 4, 4, dtype=torch.int32), torch.zeros(1, 4, 4, 8),
                              torch.zeros(1, 4, 4, 8, dtype=torch.int32), torch.zeros(1, 4, 4, 3),
                              torch.zeros(1, 2, 2, 2, 2, 3), 2)
    assert issubclass(nr.rasterize.RasterizeFunction, torch.autograd.Function) if hasattr(nr.rasterize, 'RasterizeFunction') \
        else True


def test_load_obj_matches_reference_parser(golden, tmp_path):
    """nr.load_obj vs the tensors the reference parser produced for the same OBJ (rasterizer_module64 fixture)."""
    import numpy as np
    import neural_renderer as nr
    from rnr_amd import scene
    g = golden('rasterizer_module64')
    mesh = {k: g['mesh_' + k] for k in ['v', 'vt', 'vn', 'f_v_idx', 'f_vt_idx', 'f_vn_idx']}
    p = str(tmp_path / 'm.obj')
    scene.write_obj(p, mesh)
    v_attr, f_attr = nr.load_obj(p, normalization=False, use_cuda=False)
    for k in ['v', 'vt', 'vn']:
        assert np.array_equal(v_attr[k].numpy(), g['loaded_' + k]), k
    for k in ['f_v_idx', 'f_vt_idx', 'f_vn_idx']:
        assert f_attr[k].dtype == torch.int32 and np.array_equal(f_attr[k].numpy(), g['loaded_' + k]), k


def test_native_obj_reader_edge_cases(golden, tmp_path):
    """csrc/objparse.hip (rnr_obj_scan / rnr_obj_parse) vs the reference parser's output (fixture `obj_edge_cases`,
    load_obj.py:108-209 run in the build container) on awkwardly formatted OBJ text: bit-identical float32 / int32."""
    import neural_renderer as nr
    g = golden('obj_edge_cases')
    for name in sorted({k.split(':')[0] for k in g.files}):
        p = str(tmp_path / (name + '.obj'))
        with open(p, 'wb') as fh:
            fh.write(g[name + ':text'].tobytes())
        va, fa = nr.load_obj(p, normalization=False, use_cuda=False)
        for k in ['v', 'vn', 'vt']:
            assert va[k].dtype == torch.float32
            assert np.array_equal(va[k].numpy().view(np.uint32), g[name + ':' + k].view(np.uint32)), (name, k)
        for k in ['f_v_idx', 'f_vn_idx', 'f_vt_idx']:
            assert fa[k].dtype == torch.int32 and np.array_equal(fa[k].numpy(), g[name + ':' + k]), (name, k)


def test_native_obj_reader_errors_and_absent_attributes(tmp_path):
    """Own behaviour where the reference parser simply crashes: a file without vt lines gives empty vt / f_vt_idx
    (the reference dies in np.vstack([]), load_obj.py:172); quads and malformed numbers raise ValueError with the line."""
    from neural_renderer.load_obj import parse_obj_bytes
    v, vn, vt, fv, fvt, fvn = parse_obj_bytes(b"v 0 0 0\nv 1 0 0\nv 0 1 0\nvn 0 0 1\nf 1//1 2//1 3//1\nf 3//1 2//1 1//1\n")
    assert v.shape == (3, 3) and vn.shape == (1, 3) and vt.shape == (0, 2) and fvt.shape == (0, 3)
    assert np.array_equal(fv, [[0, 1, 2], [2, 1, 0]]) and np.array_equal(fvn, np.zeros((2, 3), np.int32))
    v, vn, vt, fv, fvt, fvn = parse_obj_bytes(b"")
    assert v.shape == (0, 3) and fv.shape == (0, 3)
    v, vn, vt, fv, fvt, fvn = parse_obj_bytes(b"v 1 2 3")           # no trailing newline
    assert np.array_equal(v, [[1, 2, 3]])
    with pytest.raises(ValueError, match='line 4'):
        parse_obj_bytes(b"v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3 1\n")
    with pytest.raises(ValueError, match='line 2'):
        parse_obj_bytes(b"v 0 0 0\nv 1 x 0\n")
    with pytest.raises(ValueError, match='texcoord'):
        parse_obj_bytes(b"v 0 0 0\nvt 0 0\nf 1 1 1\n")


def test_native_obj_number_grammar_is_pythons_float():
    """The reference parser converts coordinates with float() (load_obj.py:120-135): the native reader accepts the same
    spellings with the same values and rejects what float() rejects — hexadecimal, 'nan(...)', doubled signs (strtod, the
    fallback of r02, took those).  Documented deviations: underscore literals ('1_0') and non-ASCII digits are rejected."""
    from neural_renderer.load_obj import parse_obj_bytes
    toks = ['0x10', 'inf', '-Infinity', 'NaN', '+1.5', '-2e3', '.5', '+-1', 'nan(1)', '-inf', '+inf', '+nan', '1e400', '-1e400',
            '1e-400', '1.', '-.5e-3', 'infinit', '--1', '+', '1e', 'e5', '0x1p3', '1.5f', 'INF', 'iNfInItY', '1,5']
    for t in toks:
        try:
            want = np.float32(float(t))
        except ValueError:
            want = None
        try:
            v = parse_obj_bytes(('v %s 2 3\n' % t).encode('utf-8'))[0]
            got = v[0, 0]
        except ValueError:
            got = None
        if want is None:
            assert got is None, (t, got)
        else:
            assert got is not None and (got == want or (np.isnan(got) and np.isnan(want))), (t, got, want)
            assert np.signbit(got) == np.signbit(want) or np.isnan(want), t
    with pytest.raises(ValueError):
        parse_obj_bytes(b'v 1_0 2 3\n')             # float('1_0') == 10.0 in Python: the documented deviation
    with pytest.raises(ValueError):
        parse_obj_bytes('v \u0661 2 3\n'.encode('utf-8'))   # float('\u0661') == 1.0 (Arabic-Indic digit): likewise


def test_module_function_names():
    import camera, misc, render, sph_harm, network
    for mod, names in [(camera, ['get_view_dir_map', 'get_reflect_dir', 'RT_from_pos_lookat', 'get_spiral']),
                       (misc, ['interpolate_bilinear', 'interpolate_bilinear_np']),
                       (render, ['get_TBN_map', 'spherical_mapping', 'spherical_mapping_batch', 'spherical_mapping_inv']),
                       (sph_harm, ['evaluate_sh_basis', 'fit_sh_coeff', 'reconstruct_sh', 'cart2sph', 'sph2cart']),
                       (network, ['TextureMapper', 'Rasterizer', 'RenderingNet', 'Interpolater', 'RaySampler',
                                  'RayRenderer', 'LightingSH', 'LightingLP', 'DenseDeepGCN', 'Mesh'])]:
        for n in names:
            assert hasattr(mod, n), (mod.__name__, n)
    with pytest.raises(NotImplementedError):
        network.DenseDeepGCN()
    # LightingLP without probes is pure bookkeeping (network.py:631-664): constructible on a CPU-only host
    l_dir = torch.nn.functional.normalize(torch.randn(3, 32, generator=torch.Generator().manual_seed(0)), dim=0)
    lp = network.LightingLP(l_dir, num_lighting=2, num_channel=3)
    assert tuple(lp.l_samples.shape) == (2, 32, 3) and tuple(lp.l_samples_uv.shape) == (2, 32)
    assert tuple(lp(1).shape) == (1, 32, 3) and set(lp.state_dict()) == {'l_samples', 'l_dir', 'l_samples_uv'}


def test_host_helpers_match_oracle():
    import numpy as np
    import camera, misc, render
    from oracle import rnr_oracle as orc
    g = torch.Generator().manual_seed(0)
    d = torch.nn.functional.normalize(torch.randn(3, 50, generator=g), dim=0)
    assert torch.allclose(render.spherical_mapping(d), orc.spherical_mapping(d), atol=1e-7)
    uv = torch.rand(2, 50, generator=g)
    assert torch.allclose(render.spherical_mapping_inv(uv), orc.spherical_mapping_inv(uv), atol=1e-7)
    data = np.random.RandomState(0).rand(7, 9, 3).astype(np.float32)
    x = np.random.RandomState(1).rand(40).astype(np.float32) * 10 - 1
    y = np.random.RandomState(2).rand(40).astype(np.float32) * 8 - 1
    ref = orc.interpolate_bilinear(torch.from_numpy(data), torch.from_numpy(x), torch.from_numpy(y)).numpy()
    assert np.allclose(misc.interpolate_bilinear_np(data, x, y), ref, atol=1e-6)
    azi, ele = camera.get_spiral()
    assert azi.shape == (720,) and abs(azi[1] + 2.0) < 1e-9 and abs(ele[1] - 0.125) < 1e-9


def test_view_dataset_matches_reference(golden, tmp_path):
    """calib.mat -> (proj, pose, proj_inv, R_inv, ...) vs the reference's ViewDataset.read_view for every sampling
    pattern."""
    import numpy as np
    import scipy.io
    import dataio
    g = golden('dataio_views')
    calib = {k[6:]: g[k] for k in g.files if k.startswith('calib:')}
    fp = str(tmp_path / 'calib.mat')
    scipy.io.savemat(fp, calib)
    for pat in ['all', 'skip_3', 'first_5', 'after_6', 'skipinv_3', 'filter', 'only_2']:
        ds = dataio.ViewDataset(root_dir=str(tmp_path), calib_path=fp, calib_format='convert', img_size=[64, 96],
                                sampling_pattern=pat, load_img=False, load_precompute=False)
        assert len(ds) == int(g[pat + ':num']), pat
        ds.buffer_all()
        for i in range(len(ds)):
            v = ds[i][0]
            again = ds.read_view(i)                    # pure: a second call gives the same values
            for key in ['proj_orig', 'proj', 'pose', 'dist_coeffs', 'offset', 'scale', 'view_dir', 'proj_inv', 'R_inv']:
                ref = g[pat + ':' + key][i]
                assert v[key].dtype == torch.float32
                assert np.allclose(v[key].numpy(), ref, rtol=1e-6, atol=1e-6), (pat, i, key)
                assert torch.equal(v[key], again[key])
    with pytest.raises(NotImplementedError):
        dataio.ViewDataset(str(tmp_path), fp, 'convert', [64, 64], 'all', load_img=True)


def test_camera_modes_match_reference(golden):
    """nr.look_at / look / perspective / get_points_from_angles vs the reference package's outputs."""
    import numpy as np
    import neural_renderer as nr
    g = golden('camera_modes')
    v, eye3 = torch.from_numpy(g['vertices']), torch.from_numpy(g['eye3'])
    close = lambda a, k: np.allclose(a.numpy(), g[k], rtol=1e-6, atol=1e-6)
    assert close(nr.look_at(v, [0.3, -0.2, -2.7]), 'look_at_list')
    assert close(nr.look_at(v, eye3, at=[0.1, 0.0, 0.2], up=[0.0, 1.0, 0.1]), 'look_at_batch')
    assert close(nr.look(v, [0.3, -0.2, -2.7], direction=[0.1, -0.1, 1.0], up=torch.tensor([0.0, 1.0, 0.0])), 'look')
    assert close(nr.look(v, [0.3, -0.2, -2.7], direction=[0.1, -0.1, 1.0]), 'look')          # default up = +y
    assert close(nr.perspective(v + torch.tensor([0.0, 0.0, 4.0]), angle=25.0), 'perspective')
    assert np.allclose(np.array(nr.get_points_from_angles(2.7, 30.0, -40.0)), g['points_scalar'])
    pts = nr.get_points_from_angles(torch.tensor([2.0, 2.5, 3.0]), torch.tensor([10.0, -20.0, 45.0]), torch.tensor([0.0, 90.0, 200.0]))
    assert close(pts, 'points_tensor')
    with pytest.raises(ValueError):
        nr.look_at(torch.zeros(4, 3), [0, 0, -1])
    with pytest.raises(ValueError):
        nr.Renderer(camera_mode='orbit')
    r = nr.Renderer(camera_mode='look_at', viewing_angle=30)
    assert abs(r.eye[2] + (1.0 / np.tan(np.radians(30)) + 1)) < 1e-12


def test_tbn_map_type_is_a_plain_tensor_off_the_gpu():
    """render.TBNMap (what get_TBN_map returns) overrides one device-side matmul form; on the CPU, and for every other operation,
    it is torch's own tensor: same values, plain result types, views keep the type."""
    import render
    g = torch.Generator().manual_seed(3)
    t = torch.randn(2, 4, 4, 3, 3, generator=g).as_subclass(render.TBNMap)
    v = torch.randn(2, 4, 4, 3, generator=g)
    got = torch.matmul(t.reshape((-1, 3, 3)).transpose(-2, -1), v.reshape((-1, 3, 1)))
    want = torch.matmul(t.as_subclass(torch.Tensor).reshape((-1, 3, 3)).transpose(-2, -1), v.reshape((-1, 3, 1)))
    assert type(got) is torch.Tensor and torch.equal(got, want)
    assert type(t.reshape(-1, 3, 3)) is render.TBNMap and type(t.reshape(-1, 3, 3).transpose(-2, -1)) is render.TBNMap
    assert type(t[0]) is render.TBNMap
    assert type(t + 1) is torch.Tensor and type(t.sum()) is torch.Tensor and type(torch.cat([t, t])) is torch.Tensor
    assert not bool(torch.isnan(t).sum() > 0)


def test_sh_basis_array_cache_rules():
    """sph_harm.SHBasisArray (what evaluate_sh_basis returns): float64 data as the reference's contract says, one cached answer —
    `.astype(np.float32)` of the array or of a C-contiguous reshape of it returns the float32 block that came down with it, once —
    and numpy's own behaviour for everything else (slices, copies, arithmetic, transposes, other dtypes)."""
    import sph_harm
    A = sph_harm.SHBasisArray
    f32 = np.random.RandomState(0).rand(6, 9).astype(np.float32)
    x = A._wrap(f32.astype(np.float64), f32.copy())
    assert isinstance(x, np.ndarray) and x.dtype == np.float64 and not x.flags.writeable
    A.stats.update(fast=0, plain=0)
    y = x.reshape((1, 2, 3, -1))
    z = y.astype(np.float32)                                            # test_rnr.py:324
    assert type(z) is np.ndarray and z.dtype == np.float32 and z.shape == (1, 2, 3, 9) and z.flags.writeable
    assert np.array_equal(z, f32.reshape(1, 2, 3, 9)) and A.stats == {'fast': 1, 'plain': 0}
    z[...] = 0                                                          # the caller owns the block ...
    again = x.astype(np.float32)                                        # ... so a second conversion is numpy's own
    assert np.array_equal(again, f32) and A.stats == {'fast': 1, 'plain': 1}
    # everything else: plain numpy semantics, no cache
    x2 = A._wrap(f32.astype(np.float64), f32.copy())
    assert type(x2 + 1) is np.ndarray and type(x2.sum()) is np.float64
    assert np.array_equal(x2[1:3].astype(np.float32), f32[1:3])         # a slice converts the ordinary way
    assert np.array_equal(x2.T.astype(np.float32), f32.T)               # so does a non-contiguous view
    assert x2.astype(np.float16).dtype == np.float16
    c = x2.copy()
    assert c.flags.writeable and np.array_equal(c, x2)
    with pytest.raises(ValueError):
        x2[0, 0] = 1.0                                                  # read-only while it carries the cache
    assert np.array_equal(x2.astype(np.float32), f32)                   # still cached (nothing above consumed it) and right
