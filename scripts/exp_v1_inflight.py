"""One view per call (the reference's mode, test_rnr.py:265), D calls in flight: D independent RNRPipeline(max_views=1)
objects, each on its own HIP stream, fed round-robin.  frames/s and ms per frame for D = 1..4.
Usage (GPU box): python scripts/exp_v1_inflight.py [precision]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'relightable-nr_amd'))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from rnr_amd import scene  # noqa: E402
from rnr_amd.pipeline import RNRPipeline  # noqa: E402


def main():
    prec = sys.argv[1] if len(sys.argv) > 1 else 'f32'
    sys.argv = sys.argv[:1]
    args = bench.parse()
    sc = bench.build_scene(args)
    dev = torch.device('cuda', 0)
    n, warm = 96, 8
    ids = (np.arange(n + warm) * 7) % 720
    poses = {k: torch.from_numpy(v).to(dev) for k, v in scene.spiral_views(512, ids).items()}
    for D in [1, 2, 3, 4]:
        pipes = [RNRPipeline(sc['mesh'], 512, sc['textures'], sc['unet_sd'], sc['pivots_spec'], sc['pivots_diff'], None,
                             nf0=64, max_views=1, device=dev, sh_coeff=sc['sh_coeff'], sh_lmax=10,
                             skip_background_tiles=False, precision=prec) for _ in range(D)]
        streams = [torch.cuda.Stream(device=dev) for _ in range(D)]

        def one(s):
            with torch.cuda.stream(streams[s % D]):
                return pipes[s % D].render(poses['proj'][s:s + 1], poses['pose'][s:s + 1], poses['proj_inv'][s:s + 1],
                                           poses['R_inv'][s:s + 1])
        for s in range(warm):
            one(s)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for s in range(warm, warm + n):
            one(s)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print('%s  in flight %d: %.1f frames/s  %.3f ms/frame' % (prec, D, n / dt, dt / n * 1e3))
        sys.stdout.flush()
        del pipes


if __name__ == '__main__':
    main()
