"""Wall time of single conv layers of the benchmarked RenderingNet (the 22 live shapes at 512 x 512), one rnr_conv2d call
per launch as the U-Net plan issues it (statistics on, BatchNorm prologue on), V views per call.
Usage (GPU box): python scripts/layer_time.py [--views 8] [--precision f32|bf16x6|f16x3] [--layers 1,2,22] [--iters 30]
Prints per layer: us per call, TFLOP/s of the convolution (2 MACs), fraction of the matrix-core peak of that precision."""
import argparse
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'relightable-nr_amd'))
import torch  # noqa: E402

from emu_layer_table import LAYERS  # noqa: E402
from rnr_amd import _lib  # noqa: E402
from rnr_amd.ops import _ptr, _stream  # noqa: E402

DEV = torch.device('cuda:0')


def time_layer(L, idx, kind, H, cins, cout, V, flags, iters, stats_on=True, zero='', unfused=False):
    pad16 = lambda c: (c + 15) // 16 * 16
    keep, csrc = [], []
    for j, C in enumerate(cins):
        cp = pad16(C)
        d = torch.randn(V, H, H, cp, device=DEV)
        if 'a' in zero:
            d.zero_()
        sc = torch.rand(V, cp, device=DEV) * 0.5 + 0.75
        sh = torch.randn(V, cp, device=DEV) * (0.0 if 'a' in zero else 0.25)
        keep += [d, sc, sh]
        csrc.append(_lib.RnrConvSrc(d.data_ptr(), sc.data_ptr(), sh.data_ptr(), cp, 1 if kind != 2 and j == 0 else 2))
    desc = _lib.RnrConvDesc(kind, cins[0], pad16(cins[0]), cins[1] if len(cins) > 1 else 0,
                            pad16(cins[1]) if len(cins) > 1 else 0, cout, pad16(cout), flags | _lib.CONV_STATS_PREZEROED)
    cin = sum(cins)
    k = 3 if kind == 0 else 4
    shape = (cin, cout, 4, 4) if kind == 2 else (cout, cin, k, k)
    w = (torch.rand(shape, device=DEV) * 2 - 1) / (cin * k * k) ** 0.5
    if 'w' in zero:
        w.zero_()
    packed = torch.empty(L.rnr_packed_weight_floats(ctypes.byref(desc)), device=DEV)
    _lib.check(L.rnr_pack_conv_weight(ctypes.byref(desc), _ptr(w), _ptr(packed), _stream()))
    oh = H if kind == 0 else (H // 2 if kind == 1 else 2 * H)
    out = torch.empty(V, oh, oh, desc.c_out_pad, device=DEV)
    stats = torch.zeros(V, desc.c_out_pad, 2, dtype=torch.float64, device=DEV)
    wsb = L.rnr_conv_workspace_bytes(ctypes.byref(desc), V, H, H)
    ws = torch.empty(max(wsb, 1), dtype=torch.uint8, device=DEV)

    # the product path (rnr_amd.unet.UNetPlan): rnr_conv2d_fused, BatchNorm finalised by the convolution's own launch where the
    # network has one (layers 10-13 and 22 carry a bias instead); --unfused times the legacy rnr_conv2d (statistics only)
    has_bn = stats_on and idx not in (10, 11, 12, 13, 22)
    sync = torch.zeros(L.rnr_conv_sync_bytes(ctypes.byref(desc), V, H, H), dtype=torch.uint8, device=DEV)
    gamma, beta = torch.rand(cout, device=DEV) + 0.5, torch.randn(cout, device=DEV)
    scale, shift = torch.empty(V, desc.c_out_pad, device=DEV), torch.empty(V, desc.c_out_pad, device=DEV)
    cbn = _lib.RnrConvBn(gamma.data_ptr(), beta.data_ptr(), scale.data_ptr(), shift.data_ptr(), 1e-5)
    keep += [sync, gamma, beta, scale, shift]

    def call():
        if unfused:
            _lib.check(L.rnr_conv2d(ctypes.byref(desc), ctypes.byref(csrc[0]), ctypes.byref(csrc[1]) if len(csrc) > 1 else None,
                                    _ptr(packed), _ptr(out), _ptr(stats) if stats_on else None, V, H, H, _ptr(ws), wsb, _stream()))
        else:
            _lib.check(L.rnr_conv2d_fused(ctypes.byref(desc), ctypes.byref(csrc[0]), ctypes.byref(csrc[1]) if len(csrc) > 1 else None,
                                          _ptr(packed), _ptr(out), ctypes.byref(cbn) if has_bn else None, V, H, H, _ptr(ws), wsb,
                                          _ptr(sync), sync.numel(), None, _stream()))
    for _ in range(3):
        call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        call()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    taps = 9 if kind == 0 else (16 if kind == 1 else 4)
    flops = 2.0 * taps * cin * cout * oh * oh * V
    return us, flops


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--views', type=int, default=8)
    ap.add_argument('--precision', default='f32')
    ap.add_argument('--layers', default='')
    ap.add_argument('--iters', type=int, default=30)
    ap.add_argument('--winograd', action='store_true', help='RNR_CONV_WINOGRAD: F(2x2, 3x3) for the 3x3 layers it covers')
    ap.add_argument('--winograd4', action='store_true', help='+ RNR_CONV_WINOGRAD4: F(4x4, 3x3) where the shape allows')
    ap.add_argument('--no-stats', action='store_true')
    ap.add_argument('--unfused', action='store_true', help='legacy rnr_conv2d (separate split-K reduce, no BatchNorm finalise)')
    ap.add_argument('--zero', default='', help="'w' zero weights, 'a' zero activations, 'wa' both: data-dependent power")
    a = ap.parse_args()
    L = _lib.load()
    flags = (_lib.EMU_FLAGS[a.precision] | (_lib.CONV_WINOGRAD if (a.winograd or a.winograd4) else 0) |
             (_lib.CONV_WINOGRAD4 if a.winograd4 else 0))
    peak = {'f32': 157.3e12, 'bf16x6': 2.5e15 / 6, 'f16x3': 2.5e15 / 3}[a.precision]
    want = [int(x) for x in a.layers.split(',')] if a.layers else [l[0] for l in LAYERS]
    tot_us = tot_fl = 0.0
    for idx, kind, H, cins, cout in LAYERS:
        if idx not in want:
            continue
        us, fl = time_layer(L, idx, kind, H, cins, cout, a.views, flags, a.iters, not a.no_stats, a.zero, a.unfused)
        tot_us += us
        tot_fl += fl
        print('L%-2d kind %d %4d^2 %-9s -> %3d  %8.1f us  %6.1f TF/s  %.3f of peak' % (
            idx, kind, H, '+'.join(map(str, cins)), cout, us, fl / us * 1e-6, fl / us * 1e6 / peak))
        sys.stdout.flush()
    print('sum %.1f us  %.1f TF/s  %.3f of peak  (%s, %d views)' % (tot_us, tot_fl / tot_us * 1e-6, tot_fl / tot_us * 1e6 / peak,
                                                               a.precision, a.views))


if __name__ == '__main__':
    main()
