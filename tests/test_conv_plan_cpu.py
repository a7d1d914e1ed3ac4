"""CPU: host-side planning of the convolutions through the C ABI (no kernel launches): which algorithm a call runs
(rnr_conv_algorithm), the packed-weight, workspace and sync sizes that go with it."""
import ctypes

import pytest

from rnr_amd import _lib

pad16 = lambda c: (c + 15) // 16 * 16


def desc(kind, cins, c_out, flags=0):
    return _lib.RnrConvDesc(kind, cins[0], pad16(cins[0]), cins[1] if len(cins) > 1 else 0,
                            pad16(cins[1]) if len(cins) > 1 else 0, c_out, pad16(c_out), flags)


# the 22 live layers of the benchmarked RenderingNet (nf0 = 64, 512 x 512): kind, input size, input channels, output channels
LAYERS = [(0, 512, (108,), 64), (0, 512, (64,), 64), (1, 512, (64,), 128), (0, 256, (128,), 128), (1, 256, (128,), 256),
          (0, 128, (256,), 256), (1, 128, (256,), 512), (0, 64, (512,), 512), (1, 64, (512,), 512), (0, 32, (512,), 512),
          (1, 32, (512,), 512), (2, 16, (512,), 512), (0, 32, (512,), 512), (2, 32, (512, 512), 512), (0, 64, (512,), 512),
          (2, 64, (512, 512), 256), (0, 128, (256,), 256), (2, 128, (256, 256), 128), (0, 256, (128,), 128),
          (2, 256, (128, 128), 64), (0, 512, (64,), 64), (0, 512, (64, 64), 78)]


def test_algorithm_per_layer_of_the_benchmark_network():
    """Without the flag every layer is a direct implicit GEMM; with it the 3x3 layers run F(2x2, 3x3) (1; the 80-column out
    layer 3) and the 4x4 stride-2 ones F(2x2, 2x2) (2) wherever the grid — split over K if need be — fills the chip: all 22
    layers at 8 views per call, all but the three 16-pixel-wide ones at one view."""
    L = _lib.load()
    want_kind = {0: 1, 1: 2, 2: 2}
    for n_views, direct_layers in ((8, set()), (1, {8, 10, 11})):
        for i, (kind, h, cins, c_out) in enumerate(LAYERS):
            d0, d1 = desc(kind, cins, c_out), desc(kind, cins, c_out, _lib.CONV_WINOGRAD)
            assert L.rnr_conv_algorithm(ctypes.byref(d0), n_views, h, h) == 0
            want = 0 if i in direct_layers else (3 if c_out == 78 else want_kind[kind])
            assert L.rnr_conv_algorithm(ctypes.byref(d1), n_views, h, h) == want, (n_views, i)
    assert L.rnr_conv_algorithm(None, 1, 64, 64) == -1
    assert L.rnr_conv_algorithm(ctypes.byref(desc(0, (64,), 64)), 0, 64, 64) == -1


@pytest.mark.parametrize('kind,h,cins,c_out', [(0, 64, (64,), 64), (0, 64, (64, 64), 78), (1, 64, (64,), 128),
                                               (2, 64, (128, 128), 64), (0, 64, (64,), 48), (1, 64, (64,), 96)])
def test_packed_weight_sizes(kind, h, cins, c_out):
    """The Winograd image sits behind the direct one: 16 planes instead of 9 taps (3x3), 9 instead of 4 per parity class
    (transposed; K-step pairs since r05), 4 phases x 9 instead of 16 taps (stride 2), per 64- / 128- / 80-column tile, plus the look-ahead padding;
    shapes no Winograd kernel tiles (48 or 96 columns) get no image."""
    L = _lib.load()
    f32 = L.rnr_packed_weight_floats(ctypes.byref(desc(kind, cins, c_out)))
    wino = L.rnr_packed_weight_floats(ctypes.byref(desc(kind, cins, c_out, _lib.CONV_WINOGRAD)))
    ctot = sum(pad16(c) for c in cins)
    cpad = pad16(c_out)
    wstride = (cpad + 127) // 128 * 128
    assert f32 == (9 if kind == 0 else 16) * ctot * wstride
    if kind == 0 and cpad == 80:
        extra = (ctot // 4 + 3) * 5120          # + the look-ahead padding: three K steps since r05 (W80_BDIST_K)
    elif kind == 0:
        extra = 0 if cpad % 64 else (cpad // 64) * (ctot // 2 + 5) * 2048
    elif kind == 2:     # r05: the pair layout of conv_wino2p_kernel — K steps in pairs of 2 x 4608 floats, one pair of look-ahead padding
        extra = 0 if cpad % 64 else (cpad // 64) * (ctot // 4 + 1) * 9216
    else:
        extra = 0 if cpad % 128 else (cpad // 128) * (4 * ctot // 2 + 3) * 2304
    assert wino == f32 + extra


def test_split_winograd_grids_have_workspace_and_sync():
    """A small Winograd grid is split over K: the workspace holds one partial-output slab per slice, the sync buffer is
    sized for every view count up to the maximum (the plan changes with it)."""
    L = _lib.load()
    d = desc(0, (512,), 512, _lib.CONV_WINOGRAD)          # 32 x 32 map, one view: 64 workgroups -> four K slices
    assert L.rnr_conv_algorithm(ctypes.byref(d), 1, 32, 32) == 1
    out_bytes = 32 * 32 * 512 * 4
    assert L.rnr_conv_workspace_bytes(ctypes.byref(d), 1, 32, 32) >= 4 * out_bytes
    assert L.rnr_conv_workspace_bytes(ctypes.byref(d), 8, 32, 32) == 256          # 512 workgroups: no split
    assert L.rnr_conv_sync_bytes(ctypes.byref(d), 8, 32, 32) >= L.rnr_conv_sync_bytes(ctypes.byref(d), 1, 32, 32) > 0
    # masked launches: only the out layer's Winograd kernel takes a tile mask (16 x 4 pixel tiles); other layers run the
    # direct kernels when masked, on the direct kernels' 32 x 8 tiles
    out = desc(0, (64, 64), 78, _lib.CONV_WINOGRAD)
    assert L.rnr_conv_tile_count(ctypes.byref(out), 1, 512, 512) == (512 // 4) * (512 // 16)
    assert L.rnr_conv_tile_count(ctypes.byref(desc(0, (64, 64), 78)), 1, 512, 512) == (512 // 8) * (512 // 32)
    # (one 512 x 512 view of a 64-column layer: the direct plan uses 32 x 4 pixel tiles)
    assert L.rnr_conv_tile_count(ctypes.byref(desc(0, (64,), 64, _lib.CONV_WINOGRAD)), 1, 512, 512) == (512 // 4) * (512 // 32)


@pytest.mark.parametrize('kind,hw,cout', [(0, 64, 64), (1, 128, 256), (2, 64, 64)])
def test_split_winograd_grids_have_no_empty_slice(kind, hw, cout):
    """with ceil(chunks / sk) chunks per slice some chunk counts (29 cut 7 ways: 5 per slice) left a trailing slice
    that starts behind the last chunk; its weight look-ahead would then read past the column tile's image.  The plan keeps
    the slice length and drops such slices: (sk - 1) * ceil(chunks / sk) < chunks for every chunk count."""
    L = _lib.load()
    oh = hw // 2 if kind == 1 else (2 * hw if kind == 2 else hw)
    seen_split = 0
    for chunks in range(4, 70):
        d = desc(kind, (16 * chunks,), cout, _lib.CONV_WINOGRAD)
        if L.rnr_conv_algorithm(ctypes.byref(d), 1, hw, hw) not in (1, 2):
            continue
        sk = L.rnr_conv_workspace_bytes(ctypes.byref(d), 1, hw, hw) // (oh * oh * cout * 4)
        if sk < 2:
            continue
        seen_split += 1
        per = -(-chunks // sk)
        assert (sk - 1) * per < chunks and per >= 4, (chunks, sk, per)
    assert seen_split > 10


def test_winograd_f4x4_plan_and_image_size():
    """RNR_CONV_WINOGRAD4: the 36-plane image sits behind the F(2x2, 3x3) one (3x3 layers with 64 k columns only); the plan
    takes F(4x4, 3x3) when the map tiles into 32 x 16 pixels and the grid gives every CU a workgroup, and falls back otherwise."""
    L = _lib.load()
    both = _lib.CONV_WINOGRAD | _lib.CONV_WINOGRAD4
    for cins, c_out in (((128,), 128), ((64, 64), 64), ((108,), 64)):
        ctot = sum(pad16(c) for c in cins)
        w2 = L.rnr_packed_weight_floats(ctypes.byref(desc(0, cins, c_out, _lib.CONV_WINOGRAD)))
        w4 = L.rnr_packed_weight_floats(ctypes.byref(desc(0, cins, c_out, both)))
        assert w4 == w2 + (pad16(c_out) // 64) * (ctot // 2 + 2) * 4608       # + the look-ahead padding: two K steps since r05 (W4_BDIST_K)
    for kind, cins, c_out in ((0, (64, 64), 78), (1, (64,), 128), (2, (128, 128), 64)):
        assert (L.rnr_packed_weight_floats(ctypes.byref(desc(kind, cins, c_out, both))) ==
                L.rnr_packed_weight_floats(ctypes.byref(desc(kind, cins, c_out, _lib.CONV_WINOGRAD))))
    d = desc(0, (128,), 128, both)
    assert L.rnr_conv_algorithm(ctypes.byref(d), 8, 256, 256) == 4           # 128 tiles x 2 column tiles x 8 views
    assert L.rnr_conv_algorithm(ctypes.byref(d), 1, 256, 256) == 4           # 256 workgroups: one per CU
    assert L.rnr_conv_algorithm(ctypes.byref(d), 1, 128, 128) == 1           # 64 workgroups: F(2x2, 3x3) (256 of its tiles)
    assert L.rnr_conv_algorithm(ctypes.byref(d), 8, 256, 240) == 1           # width not a multiple of 32
    assert L.rnr_conv_algorithm(ctypes.byref(desc(0, (128,), 128, _lib.CONV_WINOGRAD)), 8, 256, 256) == 1
    assert L.rnr_conv_workspace_bytes(ctypes.byref(d), 8, 256, 256) == 256   # never split over K
    # the kernel keeps BatchNorm scale / shift of every input channel in a 1024-entry LDS table (r05): more channels -> F(2x2, 3x3)
    assert L.rnr_conv_algorithm(ctypes.byref(desc(0, (512, 512), 128, both)), 8, 256, 256) == 4
    assert L.rnr_conv_algorithm(ctypes.byref(desc(0, (1024, 512), 128, both)), 8, 256, 256) == 1


def test_winograd_f4x4_split_grids():
    """Small F(4x4, 3x3) grids are cut over K when the split grid gives every CU its workgroup again: slices of >= 4 chunks, no
    empty slice, workspace = one partial-output slab per slice."""
    L = _lib.load()
    both = _lib.CONV_WINOGRAD | _lib.CONV_WINOGRAD4
    for hw, c, views, want_sk in ((64, 512, 1, 4), (128, 256, 1, 2), (64, 512, 2, 2), (32, 512, 8, 2), (64, 512, 4, 1)):
        d = desc(0, (c,), c, both)
        assert L.rnr_conv_algorithm(ctypes.byref(d), views, hw, hw) == 4, (hw, c, views)
        ws = L.rnr_conv_workspace_bytes(ctypes.byref(d), views, hw, hw)
        sk = 1 if ws == 256 else ws // (views * hw * hw * c * 4)
        assert sk == want_sk, (hw, c, views, sk)
    # 32 x 32 at one view: 16 workgroups, 8 slices would give 128 < 256 -> F(2x2, 3x3) (64 of its tiles, split four ways)
    assert L.rnr_conv_algorithm(ctypes.byref(desc(0, (512,), 512, both)), 1, 32, 32) == 1
    for chunks in range(4, 40):
        d = desc(0, (16 * chunks,), 64, both)
        if L.rnr_conv_algorithm(ctypes.byref(d), 1, 128, 128) != 4:          # 32 workgroups
            continue
        sk = L.rnr_conv_workspace_bytes(ctypes.byref(d), 1, 128, 128) // (128 * 128 * 64 * 4)
        per = -(-chunks // sk)
        assert sk >= 2 and (sk - 1) * per < chunks and per >= 4 and 32 * sk >= 256, (chunks, sk)
