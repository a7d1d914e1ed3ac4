"""frames/s of RNRPipeline for precision x streams x dead-tile elimination (bench scene, 8 views per step).
Usage (GPU box): python scripts/exp_streams.py [views_per_step]"""
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
    V = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    sys.argv = sys.argv[:1]
    args = bench.parse()
    sc = bench.build_scene(args)
    dev = torch.device('cuda', 0)
    steps, warm = 12, 3
    ids = (np.arange((steps + warm) * V) * 7) % 720
    poses = {k: torch.from_numpy(v).to(dev) for k, v in scene.spiral_views(512, ids).items()}
    for prec in ['f32', 'bf16x6', 'f16x3']:
        for streams in [1, 2, 4]:
            for skip in [False, True]:
                if streams > V:
                    continue
                pipe = RNRPipeline(sc['mesh'], 512, sc['textures'], sc['unet_sd'], sc['pivots_spec'], sc['pivots_diff'], None,
                                   nf0=64, max_views=V, device=dev, sh_coeff=sc['sh_coeff'], sh_lmax=10,
                                   skip_background_tiles=skip, streams=streams, precision=prec)

                def st(s):
                    sl = slice(s * V, (s + 1) * V)
                    return pipe.render(poses['proj'][sl], poses['pose'][sl], poses['proj_inv'][sl], poses['R_inv'][sl])
                for s in range(warm):
                    st(s)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for s in range(warm, warm + steps):
                    st(s)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                print('%-7s streams=%d tile_skip=%d  %.1f frames/s  %.2f ms/step' % (prec, streams, int(skip), steps * V / dt, dt / steps * 1e3))
                sys.stdout.flush()
                del pipe


if __name__ == '__main__':
    main()
