"""-m gpu: MFMA implicit-GEMM convolutions and the U-Net plan vs plain PyTorch fp32 (F.conv2d etc. on CPU) and the
reference golden vectors.  Tolerance: exact-fp32 MFMA vs CPU fp32 with a different summation order -> 1e-4 relative
to the output scale per layer; 2e-4 abs on the tanh output of the whole net (PSNR > 70 dB)."""
import ctypes

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def T(x):
    return torch.from_numpy(np.ascontiguousarray(x))


def _act(x, a):
    return F.leaky_relu(x, 0.2) if a == 1 else (F.relu(x) if a == 2 else x)


def run_conv(kind, srcs, weight, c_out, N, H, W, flags=0):
    """srcs: list of (raw NCHW cpu tensor, scale [N,C] or None, shift [N,C] or None, act)."""
    from rnr_amd import _lib
    from rnr_amd.ops import _ptr, _stream
    L = _lib.load()
    pad16 = lambda c: (c + 15) // 16 * 16
    keep, csrc, cs = [], [], []
    for raw, sc, sh, act in srcs:
        C = raw.shape[1]
        cp = pad16(C)
        d = torch.zeros(N, H, W, cp)
        d[..., :C] = raw.permute(0, 2, 3, 1)
        d = d.to(DEV)
        scd = shd = None
        if sc is not None:
            scd = torch.zeros(N, cp); scd[:, :C] = sc; scd = scd.to(DEV)
        if sh is not None:
            shd = torch.zeros(N, cp); shd[:, :C] = sh; shd = shd.to(DEV)
        keep += [d, scd, shd]
        csrc.append(_lib.RnrConvSrc(d.data_ptr(), scd.data_ptr() if scd is not None else None,
                                    shd.data_ptr() if shd is not None else None, cp, act))
        cs.append((C, cp))
    desc = _lib.RnrConvDesc(kind, cs[0][0], cs[0][1], cs[1][0] if len(cs) > 1 else 0, cs[1][1] if len(cs) > 1 else 0,
                            c_out, pad16(c_out), flags)
    packed = torch.empty(L.rnr_packed_weight_floats(ctypes.byref(desc)), device=DEV)
    wd = weight.contiguous().to(DEV)
    _lib.check(L.rnr_pack_conv_weight(ctypes.byref(desc), _ptr(wd), _ptr(packed), _stream()))
    oh, ow = (H, W) if kind == 0 else ((H // 2, W // 2) if kind == 1 else (2 * H, 2 * W))
    out = torch.full((N, oh, ow, desc.c_out_pad), float('nan'), device=DEV)
    stats = torch.zeros(N, desc.c_out_pad, 2, dtype=torch.float64, device=DEV)
    wsb = L.rnr_conv_workspace_bytes(ctypes.byref(desc), N, H, W)
    ws = torch.empty(wsb, dtype=torch.uint8, device=DEV)
    _lib.check(L.rnr_conv2d(ctypes.byref(desc), ctypes.byref(csrc[0]), ctypes.byref(csrc[1]) if len(csrc) > 1 else None,
                            _ptr(packed), _ptr(out), _ptr(stats), N, H, W, _ptr(ws), wsb, _stream()))
    torch.cuda.synchronize()
    return out.cpu(), stats.cpu()


def run_conv_fused(kind, srcs, weight, c_out, N, H, W, gamma=None, beta=None, flags=0, repeats=1):
    """The product entry point rnr_conv2d_fused (one launch: convolution + BatchNorm finalise + shallow split-K combine).
    Returns (out_raw, scale, shift, sync buffer) as CPU tensors; `repeats` > 1 re-runs the call on the same sync buffer."""
    from rnr_amd import _lib
    from rnr_amd.ops import _ptr, _stream
    L = _lib.load()
    pad16 = lambda c: (c + 15) // 16 * 16
    keep, csrc, cs = [], [], []
    for raw, sc, sh, act in srcs:
        C = raw.shape[1]
        cp = pad16(C)
        d = torch.zeros(N, H, W, cp)
        d[..., :C] = raw.permute(0, 2, 3, 1)
        d = d.to(DEV)
        scd = shd = None
        if sc is not None:
            scd = torch.zeros(N, cp); scd[:, :C] = sc; scd = scd.to(DEV)
        if sh is not None:
            shd = torch.zeros(N, cp); shd[:, :C] = sh; shd = shd.to(DEV)
        keep += [d, scd, shd]
        csrc.append(_lib.RnrConvSrc(d.data_ptr(), scd.data_ptr() if scd is not None else None,
                                    shd.data_ptr() if shd is not None else None, cp, act))
        cs.append((C, cp))
    desc = _lib.RnrConvDesc(kind, cs[0][0], cs[0][1], cs[1][0] if len(cs) > 1 else 0, cs[1][1] if len(cs) > 1 else 0,
                            c_out, pad16(c_out), flags)
    packed = torch.empty(L.rnr_packed_weight_floats(ctypes.byref(desc)), device=DEV)
    wd = weight.contiguous().to(DEV)
    _lib.check(L.rnr_pack_conv_weight(ctypes.byref(desc), _ptr(wd), _ptr(packed), _stream()))
    oh, ow = (H, W) if kind == 0 else ((H // 2, W // 2) if kind == 1 else (2 * H, 2 * W))
    out = torch.full((N, oh, ow, desc.c_out_pad), float('nan'), device=DEV)
    wsb = L.rnr_conv_workspace_bytes(ctypes.byref(desc), N, H, W)
    ws = torch.empty(wsb, dtype=torch.uint8, device=DEV)
    sync = torch.zeros(L.rnr_conv_sync_bytes(ctypes.byref(desc), N, H, W), dtype=torch.uint8, device=DEV)
    scale = torch.full((N, desc.c_out_pad), float('nan'), device=DEV)
    shift = torch.full((N, desc.c_out_pad), float('nan'), device=DEV)
    cbn = None
    if gamma is not None:
        g, b = gamma.to(DEV), beta.to(DEV)
        keep += [g, b]
        cbn = _lib.RnrConvBn(g.data_ptr(), b.data_ptr(), scale.data_ptr(), shift.data_ptr(), 1e-5)
    for _ in range(repeats):
        _lib.check(L.rnr_conv2d_fused(ctypes.byref(desc), ctypes.byref(csrc[0]), ctypes.byref(csrc[1]) if len(csrc) > 1 else None,
                                      _ptr(packed), _ptr(out), ctypes.byref(cbn) if cbn else None, N, H, W, _ptr(ws), wsb,
                                      _ptr(sync), sync.numel(), None, _stream()))
    torch.cuda.synchronize()
    return out.cpu(), scale.cpu(), shift.cpu(), sync.cpu()


FUSED_CASES = [
    # kind, N, H, cins, c_out                      what the plan does at this size
    (0, 1, 64, [512], 512),                        # 128 tiles of 128x128 -> split 4, slices meet inside the launch, tickets per tile
    (0, 1, 32, [512], 512),                        # 64x64 tiles, split 4
    (1, 1, 32, [512], 512),                        # 16-pixel-wide output: 64x64 tiles, deep split (reduce kernel + separate finalise)
    (2, 1, 16, [512], 512),                        # transposed, 16 pixels wide
    (0, 1, 128, [256], 256),                       # 128x64 tiles (four waves per SIMD)
    (1, 1, 256, [128], 256),                       # 256 tiles of 128x128 split two ways
    (2, 2, 64, [256, 256], 256),                   # skip concat, two views, parity classes as neighbours
    (0, 3, 256, [64], 64),                         # 256x64 tiles, several rounds: in-kernel finalise, per-view tickets
    (0, 2, 128, [64, 64], 78),                     # 80-column remainder configuration WITH a BatchNorm behind it
    (0, 2, 24, [40], 24),                          # odd size: the gather kernel (one counter for the launch)
    (1, 5, 8, [16], 16),                           # tiny maps, tiles straddle views
]


@pytest.mark.parametrize('kind,N,H,cins,c_out', FUSED_CASES)
def test_conv_fused_equals_separate_launches(kind, N, H, cins, c_out):
    """rnr_conv2d_fused == rnr_conv2d + rnr_bn_finalize: out_raw BITWISE (same plan, same summation order — also where the
    split-K slices meet inside the launch), scale / shift to 1e-6 relative (float64 atomics arrive in any order), padded
    channels exactly 0, and the sync buffer is all zeros again after every call (three calls in a row on one buffer)."""
    from rnr_amd import _lib
    from rnr_amd.ops import _ptr, _stream
    g = torch.Generator().manual_seed(1000 * kind + 10 * H + N)
    srcs = []
    for j, C in enumerate(cins):
        raw = torch.randn(N, C, H, H, generator=g)
        sc = torch.rand(N, C, generator=g) * 0.5 + 0.75
        sh = torch.randn(N, C, generator=g) * 0.25
        srcs.append((raw, sc, sh, 1 if kind != 2 and j == 0 else 2))
    cin = sum(cins)
    k = 3 if kind == 0 else 4
    shape = (cin, c_out, 4, 4) if kind == 2 else (c_out, cin, k, k)
    w = (torch.rand(shape, generator=g) * 2 - 1) / (cin * k * k) ** 0.5
    gamma = torch.rand(c_out, generator=g) + 0.5
    beta = torch.randn(c_out, generator=g)
    out_l, stats = run_conv(kind, srcs, w, c_out, N, H, H)
    out_f, scale, shift, sync = run_conv_fused(kind, srcs, w, c_out, N, H, H, gamma, beta, repeats=3)
    assert torch.equal(out_l.view(torch.int32), out_f.view(torch.int32)), 'out_raw differs between the two entry points'
    assert int(sync.abs().max()) == 0, 'sync buffer not returned to zero'
    L = _lib.load()
    cp = (c_out + 15) // 16 * 16
    oh = H if kind == 0 else (H // 2 if kind == 1 else 2 * H)
    st = stats.to(DEV)
    sc_l, sh_l = torch.empty(N, cp, device=DEV), torch.empty(N, cp, device=DEV)
    gd, bd = gamma.to(DEV), beta.to(DEV)
    _lib.check(L.rnr_bn_finalize(_ptr(st), _ptr(gd), _ptr(bd), _ptr(sc_l), _ptr(sh_l), N, c_out, cp, float(oh * oh), 1e-5, _stream()))
    torch.cuda.synchronize()
    assert torch.allclose(scale[:, :c_out], sc_l.cpu()[:, :c_out], rtol=1e-6, atol=1e-7)
    assert torch.allclose(shift[:, :c_out], sh_l.cpu()[:, :c_out], rtol=1e-5, atol=1e-6)
    assert float(scale[:, c_out:].abs().max() if cp > c_out else 0.0) == 0.0
    assert float(shift[:, c_out:].abs().max() if cp > c_out else 0.0) == 0.0
    # no BatchNorm behind the convolution: same out_raw, scale / shift untouched
    out_n, sc_n, _, sync_n = run_conv_fused(kind, srcs, w, c_out, N, H, H)
    assert torch.equal(out_l.view(torch.int32), out_n.view(torch.int32)) and bool(torch.isnan(sc_n).all())
    assert int(sync_n.abs().max()) == 0


def ref_conv(kind, srcs, weight):
    xs = []
    for raw, sc, sh, act in srcs:
        x = raw
        if sc is not None:
            x = x * sc[:, :, None, None]
        if sh is not None:
            x = x + sh[:, :, None, None]
        xs.append(_act(x, act))
    x = torch.cat(xs, 1).double()
    w = weight.double()
    if kind == 0:
        return F.conv2d(F.pad(x, (1, 1, 1, 1), mode='reflect'), w)
    if kind == 1:
        return F.conv2d(F.pad(x, (1, 1, 1, 1), mode='reflect'), w, stride=2)
    return F.conv_transpose2d(x, w, stride=2, padding=1)


CASES = [
    # kind, N, H, W, [C per source], c_out
    (0, 1, 32, 32, [20], 16),          # padded channels both sides, cfg 256x64
    (0, 2, 16, 24, [64], 78),          # cfg 256x96, non-square, 2 views
    (0, 1, 16, 16, [64, 64], 128),     # skip concat, cfg 128x128
    (1, 2, 16, 16, [32], 64),          # 4x4 stride 2
    (1, 1, 32, 32, [64], 128),
    (2, 1, 8, 8, [64, 64], 32),        # transposed conv, concat input
    (2, 2, 4, 4, [512], 512),          # tiny map, deep K -> split-K + view-straddling tiles
    (0, 3, 2, 2, [32], 32),            # 2x2 maps: reflect on both sides, tiles straddle views
    (0, 1, 64, 64, [112], 64),         # first-layer-like (halo kernel, 32x8 tiles)
    (0, 2, 32, 64, [64], 78),          # halo kernel, 256x96 config, two views, non-square
    (0, 1, 32, 32, [64, 64], 128),     # halo kernel, 32x4 tiles, skip concat
    (0, 1, 64, 64, [256], 256),        # halo kernel + split-K over channel chunks
    (0, 2, 8, 32, [32], 32),           # halo kernel: map exactly one tile high
    (1, 1, 64, 64, [64], 128),         # 4x4 s2 halo kernel (de-interleaved columns)
    (1, 2, 64, 128, [32], 16),         # 4x4 s2 halo kernel, narrow output forced onto the 128-column config
    (1, 1, 128, 64, [128], 256),       # 4x4 s2 halo, two column tiles
    (2, 1, 32, 32, [64, 64], 64),      # transposed conv halo kernel, 32x8 tiles, concat
    (2, 2, 32, 64, [128], 128),        # transposed conv halo kernel, 32x4 tiles
    (2, 1, 8, 32, [256], 78),          # transposed conv halo, 256x96 config, split-K
    (0, 1, 16, 16, [64, 64], 128),     # map 16 px wide: halo kernel on 16 x 8 pixel tiles (two image rows per MFMA row block)
    (0, 2, 16, 16, [512], 512),        # 16 px wide, split-K, four column tiles, two views
    (1, 1, 32, 32, [512], 512),        # the U-Net's layer 11: 4x4 s2 onto a 16 x 16 map
    (1, 2, 32, 32, [64], 96),          # the same with a ragged column tile
    (2, 3, 16, 16, [512], 512),        # the U-Net's layer 12: transposed conv from a 16 x 16 map, three views
    (2, 1, 16, 16, [128, 64], 64),     # transposed conv, concat, 64 columns on the 128-column tile
    (0, 1, 8, 16, [32], 32),           # 16 wide, 8 rows: exactly one tile
    (0, 1, 512, 512, [16], 128),       # enough tiles for the 256x128 fp32 config (two waves per SIMD)
    (2, 1, 256, 256, [16, 16], 128),   # the same for the transposed conv, concat input
]


def test_conv_fused_with_in_launch_combine_opt_in():
    """The in-launch split-K combine is off by default since r06 (slabs + reduce kernel measured faster at one view per call);
    RNR_CONV_COMBINE=1 turns it on.  The library reads the switch once per process, so the split cases of the test above run
    again in a process of their own with the switch set: same bit-exact comparisons."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, RNR_CONV_COMBINE='1')
    p = subprocess.run([sys.executable, '-m', 'pytest', __file__, '-m', 'gpu', '-x', '-q', '-k',
                        'test_conv_fused_equals_separate_launches and (cins0 or cins1 or cins2 or cins3 or cins5)'], env=env, capture_output=True, text=True,
                       timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    assert ' passed' in p.stdout and 'deselected' in p.stdout


@pytest.mark.parametrize('kind,N,H,W,cins,c_out', CASES)
def test_conv_vs_torch(kind, N, H, W, cins, c_out):
    g = torch.Generator().manual_seed(kind * 100 + H + c_out)
    srcs = []
    for j, C in enumerate(cins):
        raw = torch.randn(N, C, H, W, generator=g)
        sc = torch.rand(N, C, generator=g) + 0.5 if j == 0 else None
        sh = torch.randn(N, C, generator=g) * 0.3
        srcs.append((raw, sc, sh, 1 if j == 0 else 2))
    cin = sum(cins)
    if kind == 2:
        w = torch.randn(cin, c_out, 4, 4, generator=g) / (cin * 4) ** 0.5
    else:
        k = 3 if kind == 0 else 4
        w = torch.randn(c_out, cin, k, k, generator=g) / (cin * k * k) ** 0.5
    out, stats = run_conv(kind, srcs, w, c_out, N, H, W)
    ref = ref_conv(kind, srcs, w).permute(0, 2, 3, 1)
    got = out[..., :c_out].double()
    assert torch.isfinite(out).all()
    assert float(out[..., c_out:].abs().max() if out.shape[-1] > c_out else 0.0) == 0.0
    scale = ref.abs().max()
    assert (got - ref).abs().max() < 1e-4 * scale, ((got - ref).abs().max(), scale)
    s1 = ref.sum(dim=(1, 2))
    s2 = (ref * ref).sum(dim=(1, 2))
    assert torch.allclose(stats[:, :c_out, 0], s1, rtol=1e-4, atol=1e-3 * float(scale) * H * W ** 0.5)
    assert torch.allclose(stats[:, :c_out, 1], s2, rtol=1e-4)


WINO_CASES = [
    # kind, N, H, W, [C per source], c_out, algorithm rnr_conv_algorithm must report (0 direct, 1 F(2x2,3x3), 2 F(2x2,2x2),
    # 3 F(2x2,3x3) on the 80-column out layer);
    # the Winograd plans need >= 256 (3x3) / >= 200 (4x4 stride 2) workgroups
    (0, 2, 64, 128, [64], 128, 1),      # 3x3: two column tiles, 8 x 8 pixel tiles per view, reflection on all four borders
    (0, 2, 128, 128, [108], 64, 1),     # the input layer's channel count (7 chunks, 4 padding channels), one column tile
    (0, 16, 32, 32, [64, 64], 128, 1),  # skip concat (two sources with their own scale / shift / activation), 16 views
    (0, 9, 8, 16, [256], 256, 0),       # too few tiles (36 workgroups, 144 when split four ways): the direct kernels
    (0, 1, 32, 32, [512], 512, 1),      # 64 workgroups: split-K four ways (partial outputs per slice + splitk_reduce_kernel + finalise)
    (2, 1, 64, 64, [256, 256], 256, 2), # 128 workgroups: the transposed kernel split two ways, skip concat across the slices
    (1, 1, 256, 256, [128], 256, 2),    # 128 workgroups: the stride-2 kernel split two ways (4 chunks x 4 phases per slice)
    (0, 64, 8, 16, [256], 256, 1),      # map = exactly one 16 x 8 tile per view: every halo pixel reflected
    (0, 2, 128, 128, [64, 64], 78, 3),  # the out layer's 78 columns: conv_wino80_kernel (16 x 16 x 4 MFMA, five column blocks)
    (0, 4, 64, 64, [112], 78, 3),       # ... 7 chunks, four 64 x 64 maps: exactly 256 workgroups
    (0, 1, 32, 64, [64], 78, 0),        # ... too few tiles: the direct 80-column plan
    (0, 4, 64, 64, [64], 72, 3),        # 72 live columns in the 80-column kernel: the padding columns stay zero
    (2, 2, 64, 64, [128], 256, 2),      # transposed: four parity classes on one staged halo, zero border, four column tiles
    (2, 16, 16, 16, [512], 512, 2),     # layer 12 at 16 views: 2 tiles per view, 32 chunks
    (2, 16, 32, 64, [64, 64], 64, 2),   # skip concat, non-square, one column tile
    (2, 1, 16, 16, [512], 512, 0),      # layer 12 at one view: 16 workgroups -> direct (split-K)
    (1, 4, 256, 256, [16], 128, 2),     # 4x4 s2: four phases per chunk, 16 x 16 output tiles, reflection on all borders
    (1, 4, 128, 256, [128], 256, 2),    # two column tiles of 128, non-square, 8 chunks x 4 phases
    (1, 64, 32, 32, [64], 512, 2),      # output = exactly one tile per view
    (1, 2, 64, 64, [64], 96, 0),        # 96 columns: not a multiple of 128 -> direct
]


@pytest.mark.parametrize('kind,N,H,W,cins,c_out,algo', WINO_CASES)
def test_conv_winograd_vs_torch(kind, N, H, W, cins, c_out, algo):
    """RNR_CONV_WINOGRAD (conv_wino_kernel: F(2x2, 3x3); conv_wino2_kernel: F(2x2, 2x2) for the 4x4 stride-2 convolutions) through
    the product entry point rnr_conv2d_fused against a float64 torch convolution: output, BatchNorm scale / shift from the
    statistics of the same launch, padding columns, the sync buffer left at zero, and which algorithm the plan reports.
    Tolerance 3e-5 of the output scale (measured <= 1.1e-5 of the rms; the direct kernels are held to 1e-4)."""
    from rnr_amd import _lib
    g = torch.Generator().manual_seed(kind * 1000 + H + W + c_out + N)
    srcs = []
    for j, C in enumerate(cins):
        raw = torch.randn(N, C, H, W, generator=g)
        sc = torch.rand(N, C, generator=g) + 0.5
        sh = torch.randn(N, C, generator=g) * 0.3
        srcs.append((raw, sc, sh, 1 if kind != 2 and j == 0 else 2))
    cin = sum(cins)
    if kind == 2:
        w = torch.randn(cin, c_out, 4, 4, generator=g) / (cin * 4) ** 0.5
    else:
        k = 3 if kind == 0 else 4
        w = torch.randn(c_out, cin, k, k, generator=g) / (cin * k * k) ** 0.5
    pad16 = lambda c: (c + 15) // 16 * 16
    desc = _lib.RnrConvDesc(kind, cins[0], pad16(cins[0]), cins[1] if len(cins) > 1 else 0,
                            pad16(cins[1]) if len(cins) > 1 else 0, c_out, pad16(c_out), _lib.CONV_WINOGRAD)
    assert _lib.load().rnr_conv_algorithm(ctypes.byref(desc), N, H, W) == algo
    desc.flags = 0
    assert _lib.load().rnr_conv_algorithm(ctypes.byref(desc), N, H, W) == 0
    gamma, beta = torch.rand(c_out, generator=g) + 0.5, torch.randn(c_out, generator=g)
    out, scale, shift, sync = run_conv_fused(kind, srcs, w, c_out, N, H, W, gamma, beta, flags=_lib.CONV_WINOGRAD, repeats=2)
    ref = ref_conv(kind, srcs, w).permute(0, 2, 3, 1)
    got = out[..., :c_out].double()
    assert torch.isfinite(out).all() and int(sync.to(torch.int32).abs().sum()) == 0
    assert float(out[..., c_out:].abs().max() if out.shape[-1] > c_out else 0.0) == 0.0
    peak = ref.abs().max()
    assert (got - ref).abs().max() < 3e-5 * peak, ((got - ref).abs().max(), peak)
    mean = ref.mean(dim=(1, 2))
    var = ref.var(dim=(1, 2), unbiased=False)
    sc_ref = gamma.double()[None] / torch.sqrt(var + 1e-5)
    sh_ref = beta.double()[None] - mean * sc_ref
    assert torch.allclose(scale[:, :c_out].double(), sc_ref, rtol=2e-5, atol=1e-6)
    assert torch.allclose(shift[:, :c_out].double(), sh_ref, rtol=2e-5, atol=2e-5)
    assert float(scale[:, c_out:].abs().max() if scale.shape[1] > c_out else 0.0) == 0.0


@pytest.mark.parametrize('kind,N,H,W,cins,c_out', [(0, 2, 64, 128, [64], 128), (0, 1, 32, 32, [512], 512), (2, 2, 64, 64, [128], 256),
                                                   (1, 4, 256, 256, [16], 128), (0, 2, 128, 128, [64, 64], 78)])
def test_conv_winograd_legacy_entry_point_statistics(kind, N, H, W, cins, c_out):
    """rnr_conv2d (the entry point of bn_mode 'batch_all' and of external callers: statistics into a caller buffer, no
    BatchNorm finalise, one statistics shard) with RNR_CONV_WINOGRAD, split-K case included: output and per-view
    sum / sum of squares against float64."""
    from rnr_amd import _lib
    g = torch.Generator().manual_seed(kind * 77 + H + c_out)
    srcs = [(torch.randn(N, C, H, W, generator=g), torch.rand(N, C, generator=g) + 0.5, torch.randn(N, C, generator=g) * 0.3,
             1 if kind != 2 and j == 0 else 2) for j, C in enumerate(cins)]
    cin = sum(cins)
    if kind == 2:
        w = torch.randn(cin, c_out, 4, 4, generator=g) / (cin * 4) ** 0.5
    else:
        k = 3 if kind == 0 else 4
        w = torch.randn(c_out, cin, k, k, generator=g) / (cin * k * k) ** 0.5
    out, stats = run_conv(kind, srcs, w, c_out, N, H, W, flags=_lib.CONV_WINOGRAD)
    ref = ref_conv(kind, srcs, w).permute(0, 2, 3, 1)
    assert torch.isfinite(out).all()
    assert (out[..., :c_out].double() - ref).abs().max() < 3e-5 * ref.abs().max()
    assert torch.allclose(stats[:, :c_out, 0], ref.sum(dim=(1, 2)), rtol=1e-4, atol=1e-3 * float(ref.abs().max()) * H * W ** 0.5)
    assert torch.allclose(stats[:, :c_out, 1], (ref * ref).sum(dim=(1, 2)), rtol=1e-4)


def test_conv_winograd_is_run_to_run_stable_and_flag_is_checked():
    """out_raw of the Winograd kernels is bit-reproducible run to run (fixed summation order; only the statistics are
    floating-point atomics), and the flag is refused together with an emulation format."""
    from rnr_amd import _lib
    g = torch.Generator().manual_seed(5)
    srcs = [(torch.randn(2, 64, 64, 128, generator=g), torch.rand(2, 64, generator=g) + 0.5, torch.randn(2, 64, generator=g), 1)]
    w = torch.randn(128, 64, 3, 3, generator=g) / 24.0
    a = run_conv_fused(0, srcs, w, 128, 2, 64, 128, flags=_lib.CONV_WINOGRAD)[0]
    b = run_conv_fused(0, srcs, w, 128, 2, 64, 128, flags=_lib.CONV_WINOGRAD)[0]
    assert torch.equal(a, b)
    with pytest.raises(RuntimeError, match='WINOGRAD'):
        run_conv_fused(0, srcs, w, 128, 2, 64, 128, flags=_lib.CONV_WINOGRAD | _lib.CONV_F32_EMU_F16X3)


def test_unet_plan_winograd_vs_direct():
    """UNetPlan(conv_algo='winograd') against conv_algo='direct' on a network wide enough for every Winograd kernel to run
    (the benchmark's nf0 64 at 256 x 256, 8 views): same frames to fp32 rounding, and the plan reports which layers run which
    algorithm."""
    from rnr_amd.unet import UNetPlan
    from rnr_amd.scene import unet_state_dict
    sd = unet_state_dict(30, 78, 64, 5, seed=3)
    dev = torch.device(DEV)
    wino = UNetPlan(sd, 30, 78, 64, 5, (256, 256), 8, dev, conv_algo='winograd')
    direct = UNetPlan(sd, 30, 78, 64, 5, (256, 256), 8, dev, conv_algo='direct')
    assert wino.conv_algo == 'winograd' and direct.conv_algo == 'direct'
    algos = [wino.L.rnr_conv_algorithm(ctypes.byref(s['desc']), 8, *s['in_hw']) for s in wino.steps]
    assert algos.count(1) >= 8 and algos.count(2) >= 3 and algos[-1] == 3, algos
    assert all(direct.L.rnr_conv_algorithm(ctypes.byref(s['desc']), 8, *s['in_hw']) == 0 for s in direct.steps)
    assert wino.mfma_flops_per_view(8) < 0.75 * wino.flops_per_view and direct.mfma_flops_per_view(8) == direct.flops_per_view
    x = torch.randn(8, 256, 256, wino.in_c_pad, generator=torch.Generator().manual_seed(1)).to(dev)
    x[..., 30:] = 0
    a = wino.forward(x).clone()
    b = direct.forward(x).clone()
    assert torch.isfinite(a).all()
    assert float((a - b).abs().max()) < 2e-4 * float(b.abs().max())
    # emulated precisions have no Winograd form: the option is ignored there
    assert UNetPlan(sd, 30, 78, 64, 5, (64, 64), 1, dev, precision='f16x3', conv_algo='winograd').conv_algo == 'direct'


EMU_CASES = [
    (0, 1, 64, 64, [112], 64),         # 256x64 tiles, 7 chunks
    (0, 2, 32, 64, [64, 64], 128),     # 128x128 tiles, skip concat, two views
    (0, 1, 64, 64, [256], 256),        # split-K over chunks
    (2, 1, 32, 32, [64, 64], 64),      # transposed conv, 256x64 tiles
    (2, 2, 32, 64, [128], 128),        # transposed conv, 128x128 tiles
    (0, 2, 32, 64, [64], 78),          # 80-column layer on 256 x 96 tiles
    (1, 1, 64, 64, [64], 128),         # 4x4 s2, 32x2 tiles
    (1, 2, 64, 128, [32], 16),         # 4x4 s2, narrow output on the 128-column config, two views
    (1, 1, 128, 64, [128], 256),       # 4x4 s2, two column tiles
    (1, 2, 8, 64, [32], 64),           # 4x4 s2, output 4 rows high: 128-row tiles (per-phase halo 5 x 33)
    (1, 1, 4, 128, [48], 128),         # 4x4 s2, output 2 rows high: 64-row tiles, three chunks x four phases
    (1, 1, 512, 512, [16], 128),       # 4x4 s2 at full width: reflection on all four borders, 256-row tiles
    (0, 1, 16, 16, [64, 64], 128),     # map 16 px wide: 16 x 8 pixel tiles (two image rows per MFMA row block), skip concat
    (0, 2, 16, 16, [512], 512),        # 16 px wide, 32 chunks: split-K, four column tiles, two views
    (1, 2, 32, 32, [64], 128),         # 4x4 s2 onto a 16 x 16 map (per-phase halo 17 x 9)
    (2, 1, 16, 16, [128], 64),         # transposed conv from a 16 x 16 map
    (2, 3, 16, 16, [512, 512], 512),   # the U-Net's layer-12/14 shape class at 16 x 16, three views
    (0, 1, 8, 16, [32], 32),           # 16 wide but only 8 rows: exactly one tile
    (0, 1, 8, 8, [32], 32),            # narrower than 16 px: not covered, falls back to the fp32 kernel from the same buffer
]


EMU_FORMATS = ['bf16x6', 'f16x3']


@pytest.mark.parametrize('fmt', EMU_FORMATS)
@pytest.mark.parametrize('kind,N,H,W,cins,c_out', EMU_CASES)
def test_conv_f32_emulation(kind, N, H, W, cins, c_out, fmt):
    """RNR_CONV_F32_EMU_BF16X6 / _F16X3: fp32 emulated on the 16-bit matrix cores must be as close to a float64
    convolution as the exact-fp32 MFMA path is (its error is fp32 accumulation rounding, not the operand split)."""
    from rnr_amd import _lib
    g = torch.Generator().manual_seed(7 + kind + H + c_out)
    srcs = []
    for j, C in enumerate(cins):
        raw = torch.randn(N, C, H, W, generator=g) * (3.0 if j == 0 else 0.05)       # mixed magnitudes
        sc = torch.rand(N, C, generator=g) + 0.5 if j == 0 else None
        sh = torch.randn(N, C, generator=g) * 0.3
        srcs.append((raw, sc, sh, 1 if j == 0 else 2))
    cin = sum(cins)
    k = 4 if kind else 3
    w = (torch.randn(cin, c_out, 4, 4, generator=g) if kind == 2 else torch.randn(c_out, cin, k, k, generator=g)) / (cin * k * k) ** 0.5
    ref = ref_conv(kind, srcs, w).permute(0, 2, 3, 1)
    f32, st32 = run_conv(kind, srcs, w, c_out, N, H, W)
    emu, stemu = run_conv(kind, srcs, w, c_out, N, H, W, flags=_lib.EMU_FLAGS[fmt])
    e32 = (f32[..., :c_out].double() - ref).abs()
    eemu = (emu[..., :c_out].double() - ref).abs()
    scale = float(ref.abs().max())
    assert torch.isfinite(emu).all()
    assert eemu.max() < 1e-4 * scale
    # same error class as exact fp32: rms within 1.5x, max within 3x (usually it is the smaller of the two)
    assert eemu.pow(2).mean().sqrt() <= 1.5 * e32.pow(2).mean().sqrt() + 1e-9 * scale, (eemu.pow(2).mean().sqrt(), e32.pow(2).mean().sqrt())
    assert eemu.max() <= 3.0 * e32.max() + 1e-8 * scale
    assert torch.allclose(stemu[:, :c_out, 0], st32[:, :c_out, 0], rtol=1e-4, atol=1e-3 * scale * H * W ** 0.5)
    assert torch.allclose(stemu[:, :c_out, 1], st32[:, :c_out, 1], rtol=1e-4)


@pytest.mark.parametrize('wmag', [1e-6, 3e-2, 40.0, 3e4])
def test_f16x3_weight_scale_invariance(wmag):
    """f16x3 pre-scales the weights by a per-layer power of two so that their fp16 residual terms stay normal: the
    relative error vs float64 must not depend on the magnitude of the weights (1e-6 ... 3e4, beyond fp16's range)."""
    from rnr_amd import _lib
    g = torch.Generator().manual_seed(11)
    N, H, W, C, c_out = 1, 32, 32, 64, 64
    srcs = [(torch.randn(N, C, H, W, generator=g), None, None, 1)]
    w = torch.randn(c_out, C, 3, 3, generator=g) / (C * 9) ** 0.5 * wmag
    ref = ref_conv(0, srcs, w).permute(0, 2, 3, 1)
    f32, _ = run_conv(0, srcs, w, c_out, N, H, W)
    emu, _ = run_conv(0, srcs, w, c_out, N, H, W, flags=_lib.CONV_F32_EMU_F16X3)
    rel = lambda x: float(((x[..., :c_out].double() - ref).pow(2).mean().sqrt()) / ref.pow(2).mean().sqrt())
    assert torch.isfinite(emu).all()
    assert rel(emu) <= 1.1 * rel(f32), (rel(emu), rel(f32))
    assert rel(emu) < 2e-7


@pytest.mark.parametrize('fmt', EMU_FORMATS)
def test_emulation_halo_swap_stress(fmt):
    """Regression: at 512^2 the 256x96 bf16x6 kernel once produced sporadically corrupted 16-lane groups of its halo
    (duplicate stores of the slots past the halo).  Ten launches of the out-layer shape must all agree with fp32."""
    from rnr_amd import _lib
    g = torch.Generator().manual_seed(1)
    N, H, W, cins, c_out = 1, 512, 512, [64, 64], 78
    srcs = [(torch.randn(N, C, H, W, generator=g), None, torch.randn(N, C, generator=g) * 0.3, 1) for C in cins]
    w = torch.randn(c_out, sum(cins), 3, 3, generator=g) / (sum(cins) * 9) ** 0.5
    nat, _ = run_conv(0, srcs, w, c_out, N, H, W)
    for rep in range(10):
        emu, _ = run_conv(0, srcs, w, c_out, N, H, W, flags=_lib.EMU_FLAGS[fmt])
        assert (emu - nat).abs().max() < 1e-4, rep


def _random_conv_cases(n, seed):
    rng = np.random.RandomState(seed)
    cases = []
    while len(cases) < n:
        kind = int(rng.randint(0, 3))
        N = int(rng.randint(1, 4))
        # output-space width: 16, 32, 48, 64, 96 ... exercises the 16-wide tiles, the 32-wide tiles and the gather kernel (48)
        wo = int(rng.choice([8, 16, 16, 32, 32, 48, 64, 96]))
        ho = int(rng.choice([2, 4, 8, 12, 16, 24, 32, 40]))
        H, W = (ho, wo) if kind != 1 else (2 * ho, 2 * wo)
        two = rng.rand() < 0.4
        cins = [int(rng.choice([3, 16, 20, 32, 48, 64, 100]))] + ([int(rng.choice([16, 32, 64]))] if two else [])
        c_out = int(rng.choice([3, 16, 30, 64, 78, 96, 128, 160, 256]))
        cases.append((kind, N, H, W, cins, c_out))
    return cases


@pytest.mark.parametrize('fmt', ['f32', 'bf16x6', 'f16x3'])
def test_conv_random_shapes(fmt):
    """Seeded sweep over layer shapes (all three kinds, 1-3 views, ragged channel counts, map widths that select the 32-wide
    halo tiles, the 16-wide ones, the per-phase stride-2 scheme and the gather kernel, with and without skip concat):
    every conv path must agree with a float64 convolution; catches tile / index arithmetic slips the fixed cases miss."""
    from rnr_amd import _lib
    flag = _lib.EMU_FLAGS[fmt]
    for ci, (kind, N, H, W, cins, c_out) in enumerate(_random_conv_cases(36, 2026)):
        g = torch.Generator().manual_seed(1000 + ci)
        srcs = []
        for j, C in enumerate(cins):
            raw = torch.randn(N, C, H, W, generator=g)
            sc = torch.rand(N, C, generator=g) + 0.5 if j == 0 else None
            sh = torch.randn(N, C, generator=g) * 0.3
            srcs.append((raw, sc, sh, 1 if j == 0 else 2))
        cin = sum(cins)
        k = 4 if kind else 3
        w = (torch.randn(cin, c_out, 4, 4, generator=g) if kind == 2 else torch.randn(c_out, cin, k, k, generator=g)) / (cin * k * k) ** 0.5
        out, stats = run_conv(kind, srcs, w, c_out, N, H, W, flags=flag)
        ref = ref_conv(kind, srcs, w).permute(0, 2, 3, 1)
        got = out[..., :c_out].double()
        tag = (fmt, ci, kind, N, H, W, cins, c_out)
        assert torch.isfinite(out).all(), tag
        scale = float(ref.abs().max())
        assert float((got - ref).abs().max()) < 1e-4 * scale, (tag, float((got - ref).abs().max()), scale)
        assert float(out[..., c_out:].abs().max() if out.shape[-1] > c_out else 0.0) == 0.0, tag
        assert torch.allclose(stats[:, :c_out, 1], (ref * ref).sum(dim=(1, 2)), rtol=1e-4), tag


def _sd(g):
    return {k[3:]: T(g[k]) for k in g.files if k.startswith('sd:')}


@pytest.mark.parametrize('name,cin,cout,hw', [('unet_nf4', 10, 6, 64), ('unet_dnr_nf4', 7, 3, 32)])
def test_unet_plan_golden(golden, name, cin, cout, hw):
    from rnr_amd import ops
    from rnr_amd.unet import UNetPlan
    g = golden(name)
    x = T(g['x'])
    N = x.shape[0]
    plan = UNetPlan(_sd(g), cin, cout, 4, 5, (hw, hw), N, torch.device(DEV))
    raw = plan.forward(ops.nchw_to_nhwc(x.to(DEV), plan.in_c_pad))
    y = ops.nhwc_to_nchw(raw, cout, bias=plan.out_bias, apply_tanh=True).cpu()
    ref = T(g['y'])
    assert (y - ref).abs().max() < 3e-4, (y - ref).abs().max()
    # one view at a time must give the same numbers: statistics are per view
    y0 = ops.nhwc_to_nchw(plan.forward(ops.nchw_to_nhwc(x[:1].to(DEV), plan.in_c_pad)), cout, bias=plan.out_bias,
                          apply_tanh=True).cpu()
    assert (y0 - y[:1]).abs().max() < 1e-5


def test_unet_nf16_vs_oracle():
    """Wider net (channels 16..128) at 128^2, 2 views: per-view statistics, every tile configuration."""
    from oracle import rnr_oracle as orc
    from rnr_amd import ops, testing
    from rnr_amd.unet import UNetPlan
    sd = testing.unet_state_dict(30, 78, 16, seed=3, out_channels_gcn=16)
    x = torch.randn(2, 30, 128, 128, generator=torch.Generator().manual_seed(0))
    plan = UNetPlan(sd, 30, 78, 16, 5, (128, 128), 2, torch.device(DEV))
    raw = plan.forward(ops.nchw_to_nhwc(x.to(DEV), plan.in_c_pad))
    y = ops.nhwc_to_nchw(raw, 78, bias=plan.out_bias, apply_tanh=True).cpu()
    ref = orc.unet_forward(sd, x)
    assert (y - ref).abs().max() < 5e-4, (y - ref).abs().max()
    assert orc.psnr(y, ref, peak=2.0) > 70


def test_plan_built_for_many_views_runs_fewer():
    """Split-K is chosen per call from the number of views passed; a plan sized for 4 views must also serve 1 or 3
    (workspace sizing regression)."""
    from oracle import rnr_oracle as orc
    from rnr_amd import ops, testing
    from rnr_amd.unet import UNetPlan
    sd = testing.unet_state_dict(16, 8, 16, seed=7, use_gcn=False)
    x = torch.randn(4, 16, 128, 128, generator=torch.Generator().manual_seed(3))
    plan = UNetPlan(sd, 16, 8, 16, 5, (128, 128), 4, torch.device(DEV))
    ref = orc.unet_forward(sd, x)
    for n in (1, 3, 4):
        raw = plan.forward(ops.nchw_to_nhwc(x[:n].to(DEV), plan.in_c_pad))
        y = ops.nhwc_to_nchw(raw, 8, bias=plan.out_bias, apply_tanh=True).cpu()
        assert (y - ref[:n]).abs().max() < 5e-4, n


def test_unet_plan_fused_equals_separate_launches(monkeypatch):
    """UNetPlan on rnr_conv2d_fused (one launch per convolution, the default) vs the same plan on the separate launches of
    round 2 (RNR_UNET_UNFUSED=1: rnr_conv2d_masked + rnr_bn_finalize_reset per layer): same raw output to 2e-6 (BatchNorm
    statistics are float64 atomics in both), for 1, 2 and 3 views on a plan built for 3; every sync buffer is zero again
    afterwards."""
    from rnr_amd import ops, testing
    from rnr_amd.unet import UNetPlan
    sd = testing.unet_state_dict(30, 78, 16, seed=11, out_channels_gcn=16)
    x = torch.randn(3, 30, 128, 128, generator=torch.Generator().manual_seed(5))
    fused = UNetPlan(sd, 30, 78, 16, 5, (128, 128), 3, torch.device(DEV))
    assert fused.fused
    monkeypatch.setenv('RNR_UNET_UNFUSED', '1')
    legacy = UNetPlan(sd, 30, 78, 16, 5, (128, 128), 3, torch.device(DEV))
    assert not legacy.fused
    for n in (1, 3, 2):
        a = fused.forward(ops.nchw_to_nhwc(x[:n].to(DEV), fused.in_c_pad)).clone()
        b = legacy.forward(ops.nchw_to_nhwc(x[:n].to(DEV), legacy.in_c_pad))
        assert float((a - b)[..., :78].abs().max()) < 2e-6 * max(1.0, float(b[..., :78].abs().max())), n
    torch.cuda.synchronize()
    assert all(int(s['sync'].max()) == 0 for s in fused.steps)


def test_f16x3_range_guard_is_available():
    """UNetPlan(check_finite=True): the debugging aid for the fp16 range of the f16x3 emulation.  Ordinary weights
    pass; a network input scaled to 1e6 lies beyond 65504 -> inf in the fp16 split of the first layer's operand ->
    FloatingPointError instead of silent NaN frames (the exact-fp32 plan takes the same input without complaint)."""
    from rnr_amd import ops, testing
    from rnr_amd.unet import UNetPlan
    sd = testing.unet_state_dict(16, 8, 16, seed=21, use_gcn=False)
    x = torch.randn(1, 16, 64, 64, generator=torch.Generator().manual_seed(9))
    plan = UNetPlan(sd, 16, 8, 16, 5, (64, 64), 1, torch.device(DEV), precision='f16x3', check_finite=True)
    plan.forward(ops.nchw_to_nhwc(x.to(DEV), plan.in_c_pad))
    with pytest.raises(FloatingPointError):
        plan.forward(ops.nchw_to_nhwc((x * 1e6).to(DEV), plan.in_c_pad))
    exact = UNetPlan(sd, 16, 8, 16, 5, (64, 64), 1, torch.device(DEV), check_finite=True)
    exact.forward(ops.nchw_to_nhwc((x * 1e6).to(DEV), exact.in_c_pad))


def test_winograd_plan_checks_its_first_call_for_non_finite_values():
    """UNetPlan's default under conv_algo='winograd' (VERDICT r03 weak 9): the FIRST forward checks input and output once and
    warns when they hold inf / NaN (Winograd spreads them over whole 2 x 2 tiles, a direct convolution does not); later calls
    are unchecked, check_finite=False never checks, the direct plan has no such default."""
    import warnings
    from rnr_amd.unet import UNetPlan
    from rnr_amd.scene import unet_state_dict
    sd = unet_state_dict(30, 78, 64, 5, seed=3)
    dev = torch.device(DEV)
    x = torch.randn(1, 256, 256, 32, generator=torch.Generator().manual_seed(1)).to(dev)
    x[..., 30:] = 0
    bad = x.clone()
    bad[0, 100, 100, 3] = float('inf')
    plan = UNetPlan(sd, 30, 78, 64, 5, (256, 256), 1, dev, conv_algo='winograd')
    assert plan.check_finite == 'first'
    with pytest.warns(RuntimeWarning, match='non-finite'):
        plan.forward(bad)
    assert plan.check_finite is False
    with warnings.catch_warnings():
        warnings.simplefilter('error')
        plan.forward(bad)                                   # only the first call is checked
        clean = UNetPlan(sd, 30, 78, 64, 5, (256, 256), 1, dev, conv_algo='winograd', share_weights_with=plan)
        assert clean.conv_algo == 'winograd'
        clean.forward(x)                                    # finite: no warning
        UNetPlan(sd, 30, 78, 64, 5, (256, 256), 1, dev, conv_algo='winograd', check_finite=False, share_weights_with=plan).forward(bad)
        d = UNetPlan(sd, 30, 78, 64, 5, (256, 256), 1, dev, conv_algo='direct')
        assert d.check_finite is False
        d.forward(bad)
    # a plan that shares packed weights inherits the donor's algorithm and refuses a contradicting one
    assert UNetPlan(sd, 30, 78, 64, 5, (256, 256), 1, dev, share_weights_with=d).conv_algo == 'direct'
    with pytest.raises(ValueError):
        UNetPlan(sd, 30, 78, 64, 5, (256, 256), 1, dev, conv_algo='winograd', share_weights_with=d)
    with pytest.raises(ValueError):
        UNetPlan(sd, 30, 78, 64, 5, (256, 256), 1, dev, precision='f16x3', share_weights_with=d)


WINO4_CASES = [
    # N, H, W, [C per source], c_out, algorithm rnr_conv_algorithm must report with RNR_CONV_WINOGRAD | RNR_CONV_WINOGRAD4
    # (the F(4x4, 3x3) plan needs >= 256 workgroups of 32 x 16 pixels x 64 columns)
    (16, 64, 64, [64], 128, 4),         # 8 tiles per view x 2 column tiles x 16 views = 256 workgroups: reflection on all four borders
    (8, 128, 128, [108], 64, 4),        # the input layer's channel count (7 chunks, 4 padding channels), one column tile
    (16, 32, 32, [64, 64], 128, 1),     # 64 workgroups of F(4x4, 3x3): too few -> F(2x2, 3x3) from the same packed buffer (256 of its tiles)
    (64, 16, 32, [256], 256, 4),        # map = exactly one 32 x 16 tile per view: every halo pixel reflected; 4 column tiles
    (16, 64, 128, [32, 96], 64, 4),     # skip concat with unequal sources (2 + 6 chunks), non-square
    (2, 128, 128, [64, 64], 78, 3),     # 80 columns: no F(4x4, 3x3) image, the out layer's F(2x2, 3x3) kernel
    (4, 64, 72, [64], 64, 0),           # width not a multiple of 16: direct
    (1, 64, 64, [512], 512, 4),         # 64 workgroups: F(4x4, 3x3) split four ways over K (partial outputs + splitk_reduce_kernel + finalise)
    (1, 128, 128, [256], 256, 4),       # 128 workgroups: split two ways
    (16, 32, 32, [272], 128, 4),        # 64 workgroups x 17 chunks: four slices of 5, 5, 5, 2 chunks (the last one short)
    (16, 32, 32, [512], 512, 4),        # L10 / L13 of the 16-view bench plan: exactly 256 UN-SPLIT workgroups, 32 chunks
    (16, 64, 64, [512], 512, 4),        # L8 / L15 at 16 views: 1024 workgroups, 32 chunks, 8 column tiles
]


@pytest.mark.parametrize('N,H,W,cins,c_out,algo', WINO4_CASES)
def test_conv_winograd_f4x4_vs_torch(N, H, W, cins, c_out, algo):
    """RNR_CONV_WINOGRAD4 (conv_wino4_kernel: F(4x4, 3x3) on the points 0, +-3/4, +-3/2, inf) through rnr_conv2d_fused against a
    float64 torch convolution: output, BatchNorm scale / shift of the same launch, padding columns, sync buffer left at zero,
    the algorithm the plan reports, and the fallback (F(2x2, 3x3) / direct from the same packed buffer) for the shapes it does
    not cover.  Tolerance 1e-4 of the output peak (VERDICT r03 item 6's gate; measured ~1e-5)."""
    from rnr_amd import _lib
    g = torch.Generator().manual_seed(4000 + H + W + c_out + N)
    srcs = []
    for j, C in enumerate(cins):
        raw = torch.randn(N, C, H, W, generator=g)
        sc = torch.rand(N, C, generator=g) + 0.5
        sh = torch.randn(N, C, generator=g) * 0.3
        srcs.append((raw, sc, sh, 1 if j == 0 else 2))
    cin = sum(cins)
    w = torch.randn(c_out, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
    pad16 = lambda c: (c + 15) // 16 * 16
    flags = _lib.CONV_WINOGRAD | _lib.CONV_WINOGRAD4
    desc = _lib.RnrConvDesc(0, cins[0], pad16(cins[0]), cins[1] if len(cins) > 1 else 0, pad16(cins[1]) if len(cins) > 1 else 0,
                            c_out, pad16(c_out), flags)
    assert _lib.load().rnr_conv_algorithm(ctypes.byref(desc), N, H, W) == algo
    gamma, beta = torch.rand(c_out, generator=g) + 0.5, torch.randn(c_out, generator=g)
    out, scale, shift, sync = run_conv_fused(0, srcs, w, c_out, N, H, W, gamma, beta, flags=flags, repeats=2)
    ref = ref_conv(0, srcs, w).permute(0, 2, 3, 1)
    got = out[..., :c_out].double()
    assert torch.isfinite(out).all() and int(sync.to(torch.int32).abs().sum()) == 0
    assert float(out[..., c_out:].abs().max() if out.shape[-1] > c_out else 0.0) == 0.0
    peak = ref.abs().max()
    err = (got - ref).abs().max()
    assert err < (3e-5 if algo in (1, 3) else 1e-4) * peak, (err, peak)
    mean = ref.mean(dim=(1, 2))
    var = ref.var(dim=(1, 2), unbiased=False)
    sc_ref = gamma.double()[None] / torch.sqrt(var + 1e-5)
    sh_ref = beta.double()[None] - mean * sc_ref
    assert torch.allclose(scale[:, :c_out].double(), sc_ref, rtol=5e-5, atol=1e-6)
    assert torch.allclose(shift[:, :c_out].double(), sh_ref, rtol=5e-5, atol=5e-5)
    # the flag without its companion is refused
    with pytest.raises(RuntimeError, match='WINOGRAD4'):
        run_conv_fused(0, srcs, w, c_out, N, H, W, gamma, beta, flags=_lib.CONV_WINOGRAD4)



@pytest.mark.parametrize('N,H,W,cins,c_out', [(1, 64, 64, [512], 512), (1, 128, 128, [256], 256), (16, 64, 64, [64], 128)])
def test_conv_winograd_f4x4_legacy_entry_point_and_combine_ab(N, H, W, cins, c_out, monkeypatch):
    """F(4x4, 3x3) through the legacy entry point rnr_conv2d (statistics into a caller buffer; a split grid writes slabs that
    splitk_reduce_kernel adds) against float64, and — for the split grids — the product entry point's in-launch combine against
    it: the same partial tiles are added in the same slice order, so out_raw is BIT-identical between the two paths."""
    from rnr_amd import _lib
    g = torch.Generator().manual_seed(4400 + H + c_out)
    srcs = [(torch.randn(N, C, H, W, generator=g), torch.rand(N, C, generator=g) + 0.5, torch.randn(N, C, generator=g) * 0.3, 1)
            for C in cins]
    cin = sum(cins)
    w = torch.randn(c_out, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
    flags = _lib.CONV_WINOGRAD | _lib.CONV_WINOGRAD4
    out, stats = run_conv(0, srcs, w, c_out, N, H, W, flags=flags)
    ref = ref_conv(0, srcs, w).permute(0, 2, 3, 1)
    assert torch.isfinite(out).all()
    assert (out[..., :c_out].double() - ref).abs().max() < 1e-4 * ref.abs().max()
    assert torch.allclose(stats[:, :c_out, 0], ref.sum(dim=(1, 2)), rtol=1e-4, atol=1e-3 * float(ref.abs().max()) * H * W ** 0.5)
    assert torch.allclose(stats[:, :c_out, 1], (ref * ref).sum(dim=(1, 2)), rtol=1e-4)
    gamma, beta = torch.rand(c_out, generator=g) + 0.5, torch.randn(c_out, generator=g)
    fused = run_conv_fused(0, srcs, w, c_out, N, H, W, gamma, beta, flags=flags)[0]
    assert torch.equal(fused, out)


FULL_SIZE_LAYERS = [
    # kind, H, [C per source], c_out, Winograd algorithm expected at N = 4        the U-Net layer of SURVEY App. A it stands for
    (0, 512, [64], 64, 4),              # L2 / L21: conv_wino4_kernel, 4 chunks
    (0, 512, [64, 64], 78, 3),          # L22: the 80-column out layer (conv_wino80_kernel)
    (1, 512, [64], 128, 2),             # L3: 4x4 stride 2 (conv_wino2_kernel<1>)
    (2, 256, [128, 128], 64, 2),        # L20: transposed 4x4 stride 2 over the skip concat (conv_wino2p_kernel<2>)
]


@pytest.mark.parametrize('kind,H,cins,c_out,algo', FULL_SIZE_LAYERS)
def test_winograd_layers_at_full_size_properties(kind, H, cins, c_out, algo):
    """The benchmark's own layer sizes (512^2 / 256^2 maps, too big for a CPU convolution in a test) through size-independent
    properties: (1) the product kernel agrees with the direct kernel on the same input, (2) linearity: conv(a x + b y) =
    a conv(x) + b conv(y), (3) the statistics the launch returns are the sums over the output it wrote (a checksum of
    checksums, float64), (4) 3x3 only: the same image shifted by one pixel (tile phase of the F(4x4) / F(2x2) tiling changes
    for every output) gives the shifted result away from the borders.  Tolerance 1e-4 of the output peak, as for the small
    shapes that are checked against torch."""
    from rnr_amd import _lib
    N = 4
    g = torch.Generator().manual_seed(7000 + kind * 13 + H + c_out)
    cin = sum(cins)
    k = 3 if kind == 0 else 4
    w = (torch.randn(c_out, cin, k, k, generator=g) if kind != 2 else torch.randn(cin, c_out, k, k, generator=g)) / (cin * k * k / (4 if kind == 2 else 1)) ** 0.5
    xs = [torch.randn(N, C, H, H, generator=g) for C in cins]
    ys = [torch.randn(N, C, H, H, generator=g) for C in cins]
    flags = _lib.CONV_WINOGRAD | _lib.CONV_WINOGRAD4
    pad16 = lambda c: (c + 15) // 16 * 16
    desc = _lib.RnrConvDesc(kind, cins[0], pad16(cins[0]), cins[1] if len(cins) > 1 else 0, pad16(cins[1]) if len(cins) > 1 else 0,
                            c_out, pad16(c_out), flags)
    assert _lib.load().rnr_conv_algorithm(ctypes.byref(desc), N, H, H) == algo
    plain = lambda ts: [(t, None, None, 0) for t in ts]
    ox, sx = run_conv(kind, plain(xs), w, c_out, N, H, H, flags=flags)
    od, _ = run_conv(kind, plain(xs), w, c_out, N, H, H, flags=0)
    peak = float(od[..., :c_out].abs().max())
    assert torch.isfinite(ox).all() and float((ox - od)[..., :c_out].abs().max()) < 1e-4 * peak
    # (3) statistics = sums over the written output
    o64 = ox[..., :c_out].double()
    s1, s2 = o64.sum(dim=(1, 2)), (o64 * o64).sum(dim=(1, 2))
    # (per-lane partial sums of 16 - 48 outputs are float32, everything above them float64)
    assert torch.allclose(sx[:, :c_out, 0], s1, rtol=1e-6, atol=1e-6 * float(s2.max()) ** 0.5)
    assert torch.allclose(sx[:, :c_out, 1], s2, rtol=1e-6)
    # (2) linearity
    a, b = 0.75, -1.5
    oy, _ = run_conv(kind, plain(ys), w, c_out, N, H, H, flags=flags)
    oz, _ = run_conv(kind, plain([a * x + b * y for x, y in zip(xs, ys)]), w, c_out, N, H, H, flags=flags)
    lin = a * ox + b * oy
    assert float((oz - lin)[..., :c_out].abs().max()) < 1e-4 * float(lin[..., :c_out].abs().max())
    # (4) tile phase
    if kind == 0:
        sh = [torch.roll(x, shifts=(1, 1), dims=(2, 3)) for x in xs]
        osh, _ = run_conv(kind, plain(sh), w, c_out, N, H, H, flags=flags)
        d = (osh[:, 3:-3, 3:-3, :c_out] - ox[:, 2:-4, 2:-4, :c_out]).abs().max()
        assert float(d) < 1e-4 * peak, float(d)
