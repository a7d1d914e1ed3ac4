"""Per-layer accuracy of the conv kernels on ALL 22 live layer shapes of the benchmarked RenderingNet (108 -> 78, nf0 = 64,
512 x 512) against a float64 convolution (torch CPU): exact-fp32 MFMA, bf16x6 and f16x3 emulation.
Inputs: seeded N(0,1) raw activations with a BatchNorm-like affine + (Leaky)ReLU prologue, weights U(+-1/sqrt(fan_in)).
Prints a markdown table (rms error relative to the rms of the exact result, and the ratio to the exact-fp32 kernel).
Usage (GPU box): python scripts/emu_layer_table.py > gpurun_out/emu_layer_table.md"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'relightable-nr_amd'))
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch  # noqa: E402

import test_gpu_unet as tu  # noqa: E402
from rnr_amd import _lib  # noqa: E402

# (#, kind, H(in), [Cin per source], Cout, act of the sources)
LAYERS = [(1, 0, 512, [108], 64), (2, 0, 512, [64], 64), (3, 1, 512, [64], 128), (4, 0, 256, [128], 128), (5, 1, 256, [128], 256),
          (6, 0, 128, [256], 256), (7, 1, 128, [256], 512), (8, 0, 64, [512], 512), (9, 1, 64, [512], 512), (10, 0, 32, [512], 512),
          (11, 1, 32, [512], 512), (12, 2, 16, [512], 512), (13, 0, 32, [512], 512), (14, 2, 32, [512, 512], 512),
          (15, 0, 64, [512], 512), (16, 2, 64, [512, 512], 256), (17, 0, 128, [256], 256), (18, 2, 128, [256, 256], 128),
          (19, 0, 256, [128], 128), (20, 2, 256, [128, 128], 64), (21, 0, 512, [64], 64), (22, 0, 512, [64, 64], 78)]


def main():
    torch.set_num_threads(os.cpu_count() or 8)
    print('| # | op | in HxW | Cin -> Cout | K | split-K f32 / emu | rel rms f32 MFMA | bf16x6 | ratio | f16x3 | ratio |')
    print('|--:|---|---|---|--:|---|--:|--:|--:|--:|--:|')
    worst = {'bf16x6': 0.0, 'f16x3': 0.0}
    for idx, kind, H, cins, cout in LAYERS:
        g = torch.Generator().manual_seed(100 + idx)
        srcs = []
        for j, C in enumerate(cins):
            raw = torch.randn(1, C, H, H, generator=g)
            sc = torch.rand(1, C, generator=g) * 0.5 + 0.75
            sh = torch.randn(1, C, generator=g) * 0.25
            srcs.append((raw, sc, sh, 1 if kind != 2 and j == 0 else 2))
        cin = sum(cins)
        k = 3 if kind == 0 else 4
        fan = cin * k * k if kind != 2 else cout * 16
        shape = (cin, cout, 4, 4) if kind == 2 else (cout, cin, k, k)
        w = (torch.rand(shape, generator=g) * 2 - 1) / fan ** 0.5
        ref = tu.ref_conv(kind, srcs, w).permute(0, 2, 3, 1)
        den = ref.pow(2).mean().sqrt()
        rel = {}
        for name, flag in [('f32', 0), ('bf16x6', _lib.CONV_F32_EMU_BF16X6), ('f16x3', _lib.CONV_F32_EMU_F16X3)]:
            out, _ = tu.run_conv(kind, srcs, w, cout, 1, H, H, flags=flag)
            rel[name] = float((out[..., :cout].double() - ref).pow(2).mean().sqrt() / den)
        K = cin * (9 if kind == 0 else (16 if kind == 1 else 4))
        # split-K depth each path uses at this batch of 1 (workspace = splitk x output bytes + 256): partial sums in
        # separate slabs shorten the fp32 accumulation chains, which lowers the error of whichever path splits
        import ctypes
        L = _lib.load()
        pad16 = lambda c: (c + 15) // 16 * 16
        oh = H if kind == 0 else (H // 2 if kind == 1 else 2 * H)
        sk = {}
        for name, flag in [('f32', 0), ('emu', _lib.CONV_F32_EMU_F16X3)]:
            desc = _lib.RnrConvDesc(kind, cins[0], pad16(cins[0]), cins[1] if len(cins) > 1 else 0,
                                    pad16(cins[1]) if len(cins) > 1 else 0, cout, pad16(cout), flag)
            ws = L.rnr_conv_workspace_bytes(ctypes.byref(desc), 1, H, H)
            sk[name] = max(1, (ws - 256) // (oh * oh * pad16(cout) * 4)) if ws > 256 else 1
        op = ['conv3x3', 'conv4x4 s2', 'convT4x4 s2'][kind]
        for n in worst:
            worst[n] = max(worst[n], rel[n] / rel['f32'])
        print('| %d | %s | %d^2 | %s -> %d | %d | %d / %d | %.3e | %.3e | %.2f | %.3e | %.2f |' % (
            idx, op, H, '+'.join(map(str, cins)), cout, K, sk['f32'], sk['emu'], rel['f32'], rel['bf16x6'], rel['bf16x6'] / rel['f32'], rel['f16x3'],
            rel['f16x3'] / rel['f32']))
        sys.stdout.flush()
    print()
    print('worst ratio to the exact-fp32 kernel: bf16x6 %.2f, f16x3 %.2f' % (worst['bf16x6'], worst['f16x3']))


if __name__ == '__main__':
    main()
