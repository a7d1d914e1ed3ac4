"""All 720 spiral_step720 views of the bench scene rendered with the Winograd convolutions (the product path) and with the
direct ones, 8 views per call and one view per call (different plans: at one view ten layers are split over K): per-view
max |difference| and PSNR of the Winograd frame against the direct one, and the run-to-run floor of each path (BatchNorm
statistics are float64 atomics).  Frames take values in about [0, 2]; PSNR uses peak 1 like oracle.psnr.
Usage (GPU box): python scripts/wino_frames_720.py [winograd|winograd4] > profiles/rNN_winograd_vs_direct_720views.json"""
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
    algo = sys.argv[1] if len(sys.argv) > 1 else 'winograd'
    args = bench.parse([])
    sc = bench.build_scene(args)
    dev = torch.device('cuda', 0)
    pv = {k: torch.from_numpy(v).to(dev) for k, v in scene.spiral_views(512, np.arange(720)).items()}
    edges = [0, 1e-7, 2e-7, 5e-7, 1e-6, 2e-6, 5e-6, 1e-5, 1e-4, 1.0]
    out = {'views': 720, 'img_size': 512, 'scene': 'bench.py build_scene (UV sphere 65 536 faces, nf0 = 64)',
           'histogram_edges_max_abs': edges, 'conv_algo': algo}
    for V in (8, 1):
        pw = bench.make_pipeline(sc, args, dev, V, conv_algo=algo, skip_background_tiles=False)
        pd = bench.make_pipeline(sc, args, dev, V, conv_algo='direct', skip_background_tiles=False)
        dmax, psnr, again_w, again_d = [], [], [], []
        for lo in range(0, 720, V):
            a = [pv[k][lo:lo + V] for k in ('proj', 'pose', 'proj_inv', 'R_inv')]
            ref = pd.render(*a).clone()
            again_d += (pd.render(*a) - ref).abs().flatten(1).max(1).values.tolist()
            img = pw.render(*a).clone()
            again_w += (pw.render(*a) - img).abs().flatten(1).max(1).values.tolist()
            d = img - ref
            dmax += d.abs().flatten(1).max(1).values.tolist()
            mse = d.pow(2).flatten(1).mean(1).clamp_min(1e-30)
            psnr += (10.0 * torch.log10(1.0 / mse)).tolist()
        m, q = np.asarray(dmax), np.asarray(psnr)
        algos = [pw.unet.L.rnr_conv_algorithm(__import__('ctypes').byref(s['desc']), V, *s['in_hw']) for s in pw.unet.steps]
        out['views_per_call_%d' % V] = {
            'layers_on_winograd_kernels': int(sum(x > 0 for x in algos)), 'layers_on_f4x4_3x3': int(sum(x == 4 for x in algos)),
            'max_abs_diff_winograd_vs_direct': {'max': float(m.max()), 'median': float(np.median(m)), 'p99': float(np.percentile(m, 99)),
                                                'histogram': np.histogram(m, bins=edges)[0].tolist(), 'worst_view': int(m.argmax())},
            'psnr_db_winograd_vs_direct': {'min': float(q.min()), 'median': float(np.median(q)), 'p1': float(np.percentile(q, 1))},
            'run_to_run_max_abs': {'winograd': float(np.max(again_w)), 'direct': float(np.max(again_d))}}
        del pw, pd
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
