"""Winograd kernels (F(2x2, 3x3) for the 3x3 layers, F(2x2, 2x2) for the 4x4 stride-2 ones) against the direct exact-fp32
kernels and a float64 convolution on U-Net layer shapes
(GPU box): max |difference| relative to the rms of the output, statistics / BatchNorm scale-shift difference.
Usage: python scripts/wino_check.py [--views 2] [--f4x4]"""
import argparse
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'relightable-nr_amd'))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from rnr_amd import _lib  # noqa: E402
from rnr_amd.ops import _ptr, _stream  # noqa: E402

DEV = torch.device('cuda:0')
pad16 = lambda c: (c + 15) // 16 * 16


def run(L, flags, kind, H, cins, cout, V, data, w, gamma, beta):
    csrc, keep = [], []
    for j, C in enumerate(cins):
        d, sc, sh = data[j]
        csrc.append(_lib.RnrConvSrc(d.data_ptr(), sc.data_ptr(), sh.data_ptr(), pad16(C), 1 if kind != 2 and j == 0 else 2))
    desc = _lib.RnrConvDesc(kind, cins[0], pad16(cins[0]), cins[1] if len(cins) > 1 else 0,
                            pad16(cins[1]) if len(cins) > 1 else 0, cout, pad16(cout), flags)
    packed = torch.empty(L.rnr_packed_weight_floats(ctypes.byref(desc)), device=DEV)
    _lib.check(L.rnr_pack_conv_weight(ctypes.byref(desc), _ptr(w), _ptr(packed), _stream()))
    oh = H if kind == 0 else (H // 2 if kind == 1 else 2 * H)
    out = torch.full((V, oh, oh, desc.c_out_pad), float('nan'), device=DEV)
    wsb = L.rnr_conv_workspace_bytes(ctypes.byref(desc), V, H, H)
    ws = torch.empty(max(wsb, 1), dtype=torch.uint8, device=DEV)
    sync = torch.zeros(L.rnr_conv_sync_bytes(ctypes.byref(desc), V, H, H), dtype=torch.uint8, device=DEV)
    scale, shift = torch.empty(V, desc.c_out_pad, device=DEV), torch.empty(V, desc.c_out_pad, device=DEV)
    cbn = _lib.RnrConvBn(gamma.data_ptr(), beta.data_ptr(), scale.data_ptr(), shift.data_ptr(), 1e-5)
    _lib.check(L.rnr_conv2d_fused(ctypes.byref(desc), ctypes.byref(csrc[0]), ctypes.byref(csrc[1]) if len(csrc) > 1 else None,
                                  _ptr(packed), _ptr(out), ctypes.byref(cbn), V, H, H, _ptr(ws), wsb, _ptr(sync), sync.numel(),
                                  None, _stream()))
    torch.cuda.synchronize()
    assert int(sync.to(torch.int32).abs().sum()) == 0, 'sync buffer not left at zero'
    return out, scale, shift


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--views', type=int, default=2)
    ap.add_argument('--f4x4', action='store_true', help='RNR_CONV_WINOGRAD | RNR_CONV_WINOGRAD4: F(4x4, 3x3) where the plan takes it (set RNR_WINO4_MIN_WGS=1 to force it)')
    a = ap.parse_args()
    L = _lib.load()
    torch.manual_seed(1)
    V = a.views
    global WFLAGS
    WFLAGS = _lib.CONV_WINOGRAD | (_lib.CONV_WINOGRAD4 if a.f4x4 else 0)
    for kind, H, cins, cout in [(0, 64, (64,), 64), (0, 128, (108,), 64), (0, 64, (128,), 128), (0, 32, (256,), 256),
                                (0, 32, (512,), 512), (0, 64, (64, 64), 64), (0, 128, (64, 64), 78), (0, 64, (112,), 78), (0, 16, (512,), 512), (0, 48, (32,), 192),
                                (2, 16, (512,), 512), (2, 32, (64, 64), 64), (2, 64, (128, 128), 64), (2, 32, (256, 256), 128),
                                (2, 16, (48,), 192), (1, 64, (64,), 128), (1, 128, (128,), 256), (1, 32, (512,), 512),
                                (1, 96, (32,), 128)]:
        data = []
        for j, C in enumerate(cins):
            cp = pad16(C)
            d = torch.randn(V, H, H, cp, device=DEV)
            d[..., C:] = 0
            sc = torch.rand(V, cp, device=DEV) * 0.5 + 0.75
            sh = torch.randn(V, cp, device=DEV) * 0.25
            data.append((d, sc, sh))
        cin = sum(cins)
        k = 3 if kind == 0 else 4
        shape = (cin, cout, 4, 4) if kind == 2 else (cout, cin, k, k)
        w = (torch.rand(shape, device=DEV) * 2 - 1) / (cin * k * k) ** 0.5
        gamma, beta = torch.rand(cout, device=DEV) + 0.5, torch.randn(cout, device=DEV)
        o_d, sc_d, sh_d = run(L, 0, kind, H, cins, cout, V, data, w, gamma, beta)
        o_w, sc_w, sh_w = run(L, WFLAGS, kind, H, cins, cout, V, data, w, gamma, beta)
        # float64 reference
        xs = []
        for j, C in enumerate(cins):
            d, sc, sh = data[j]
            slope = 0.2 if kind != 2 and j == 0 else 0.0
            v = d.double() * sc.double()[:, None, None, :] + sh.double()[:, None, None, :]
            v = torch.maximum(v, slope * v)[..., :C]
            xs.append(v)
        x = torch.cat(xs, -1).permute(0, 3, 1, 2)
        if kind == 0:
            ref = F.conv2d(F.pad(x, (1, 1, 1, 1), mode='reflect'), w.double())
        elif kind == 1:
            ref = F.conv2d(F.pad(x, (1, 1, 1, 1), mode='reflect'), w.double(), stride=2)
        else:
            ref = F.conv_transpose2d(x, w.double(), stride=2, padding=1)
        ref = ref.permute(0, 2, 3, 1)
        rms = float(ref.pow(2).mean().sqrt())
        e_d = float((o_d[..., :cout].double() - ref).abs().max()) / rms
        e_w = float((o_w[..., :cout].double() - ref).abs().max()) / rms
        r_d = float((o_d[..., :cout].double() - ref).pow(2).mean().sqrt()) / rms
        r_w = float((o_w[..., :cout].double() - ref).pow(2).mean().sqrt()) / rms
        pad_ok = bool((o_w[..., cout:] == 0).all()) if pad16(cout) > cout else True
        dsc = _lib.RnrConvDesc(kind, cins[0], pad16(cins[0]), cins[1] if len(cins) > 1 else 0, pad16(cins[1]) if len(cins) > 1 else 0,
                               cout, pad16(cout), WFLAGS)
        algo = L.rnr_conv_algorithm(ctypes.byref(dsc), V, H, H)
        print('algo %d  kind %d %4d^2 %-9s -> %3d  V=%d  max err / rms: direct %.2e winograd %.2e   rms err / rms: direct %.2e winograd %.2e   '
              'scale diff %.1e shift diff %.1e  pad cols zero %s  finite %s' % (
                  algo, kind, H, '+'.join(map(str, cins)), cout, V, e_d, e_w, r_d, r_w, float((sc_d - sc_w).abs().max()),
                  float((sh_d - sh_w).abs().max()), pad_ok, bool(torch.isfinite(o_w).all())))
        sys.stdout.flush()


if __name__ == '__main__':
    main()
