"""One view per call (the reference's mode, test_rnr.py:265) with D calls in flight: RNRPipeline(inflight=D).submit.
frames/s and ms per frame for D = 1..4; frames are compared with the one-call-at-a-time pipeline (bitwise apart from the
BatchNorm statistics' atomic summation order).
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
    ref = None
    for D in [1, 2, 3, 4]:
        pipe = RNRPipeline(sc['mesh'], 512, sc['textures'], sc['unet_sd'], sc['pivots_spec'], sc['pivots_diff'], None,
                           nf0=64, max_views=1, device=dev, sh_coeff=sc['sh_coeff'], sh_lmax=10,
                           skip_background_tiles=False, precision=prec, inflight=D)

        def one(s):
            return pipe.submit(poses['proj'][s:s + 1], poses['pose'][s:s + 1], poses['proj_inv'][s:s + 1],
                               poses['R_inv'][s:s + 1])
        for s in range(warm):
            one(s)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        hs = [one(s) for s in range(warm, warm + n)]
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        last = [h.image.clone() for h in hs[-D:]]
        if ref is None:
            ref = pipe.render(poses['proj'][warm + n - 1:warm + n], poses['pose'][warm + n - 1:warm + n],
                              poses['proj_inv'][warm + n - 1:warm + n], poses['R_inv'][warm + n - 1:warm + n]).clone()
        print('%s  in flight %d: %.1f frames/s  %.3f ms/frame   max |last frame - sequential| = %.2e' % (
            prec, D, n / dt, dt / n * 1e3, float((last[-1] - ref).abs().max())))
        sys.stdout.flush()
        del pipe


if __name__ == '__main__':
    main()
