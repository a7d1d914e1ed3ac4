"""All 720 spiral_step720 views of the bench scene rendered with the exact-fp32 convolutions and with the two fp32
emulations (f16x3, bf16x6): per-view max |difference| and PSNR of the emulated frame against the exact one — the data
behind the question "may an emulated configuration stand in for the exact one".  (Frames take values in about [0, 2];
PSNR uses peak 1 like oracle.psnr.)
Usage (GPU box): python scripts/emu_frames_720.py > profiles/rNN_emu_vs_f32_720views.json"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'relightable-nr_amd'))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from rnr_amd import scene  # noqa: E402


def main():
    args = bench.parse([])
    sc = bench.build_scene(args)
    dev = torch.device('cuda', 0)
    V = 8
    pv = {k: torch.from_numpy(v).to(dev) for k, v in scene.spiral_views(512, np.arange(720)).items()}
    pipes = {p: bench.make_pipeline(sc, args, dev, V, precision=p, skip_background_tiles=False) for p in ['f32', 'f16x3', 'bf16x6']}
    stats = {p: {'max_abs': [], 'psnr': []} for p in ['f16x3', 'bf16x6']}
    run2 = []          # the exact path against itself (BatchNorm statistics are float64 atomics: run-to-run noise floor)
    for lo in range(0, 720, V):
        sl = slice(lo, lo + V)
        a = [pv[k][sl] for k in ('proj', 'pose', 'proj_inv', 'R_inv')]
        ref = pipes['f32'].render(*a).clone()
        again = pipes['f32'].render(*a)
        run2 += (again - ref).abs().flatten(1).max(1).values.tolist()
        for p in stats:
            img = pipes[p].render(*a)
            d = (img - ref)
            stats[p]['max_abs'] += d.abs().flatten(1).max(1).values.tolist()
            mse = d.pow(2).flatten(1).mean(1).clamp_min(1e-30)
            stats[p]['psnr'] += (10.0 * torch.log10(1.0 / mse)).tolist()
    edges = [0, 1e-7, 2e-7, 5e-7, 1e-6, 2e-6, 5e-6, 1e-5, 1e-4, 1.0]
    out = {'views': 720, 'img_size': 512, 'scene': 'bench.py build_scene (UV sphere 65 536 faces, nf0 = 64)',
           'exact_f32_run_to_run_max_abs': {'max': float(np.max(run2)), 'median': float(np.median(run2))},
           'histogram_edges_max_abs': edges}
    for p, s in stats.items():
        m, q = np.asarray(s['max_abs']), np.asarray(s['psnr'])
        out[p] = {'max_abs_diff_vs_exact_f32': {'max': float(m.max()), 'median': float(np.median(m)), 'p99': float(np.percentile(m, 99)),
                                                'histogram': np.histogram(m, bins=edges)[0].tolist()},
                  'psnr_db_vs_exact_f32': {'min': float(q.min()), 'median': float(np.median(q)), 'p1': float(np.percentile(q, 1))},
                  'worst_view': int(m.argmax())}
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
