"""The ray renderer folded into the out layer (RNRPipeline(fuse_ray=True): ops.ray_weights + rnr_conv2d_ray) against the
separate ray_render_kernel: frames/s at 8 views per step and one view per call, per-stage times, frame difference.
Usage (GPU box): python scripts/exp_fuse_ray.py"""
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


def main():
    args = bench.parse([])
    sc = bench.build_scene(args)
    dev = torch.device('cuda', 0)
    ids = (np.arange(8 * 24) * 7) % 720
    pv = {k: torch.from_numpy(v).to(dev) for k, v in scene.spiral_views(512, ids).items()}
    a = lambda sl: [pv[k][sl] for k in ('proj', 'pose', 'proj_inv', 'R_inv')]
    for V in (8, 1):
        frames = {}
        for skip in (False, True):
            for fuse in (False, True):
                pipe = bench.make_pipeline(sc, args, dev, V, fuse_ray=fuse, skip_background_tiles=skip)
                for s in range(3):
                    pipe.render(*a(slice(s * V, (s + 1) * V)))
                torch.cuda.synchronize()
                n = 20 if V == 8 else 160
                t0 = time.perf_counter()
                for s in range(n):
                    img = pipe.render(*a(slice((s % 24) * V, (s % 24) * V + V)))
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / n
                evs = []
                pipe.render(*a(slice(0, V)), stage_events=evs)
                torch.cuda.synchronize()
                st = {name: e0.elapsed_time(e1) for (_, e0), (name, e1) in zip(evs[:-1], evs[1:])}
                frames[(skip, fuse)] = pipe.render(*a(slice(0, V))).clone()
                print('views/call %d  tile-skip %d  fuse_ray %d: %.1f frames/s  %.3f ms/call   stages %s' % (
                    V, skip, fuse, V / dt, dt * 1e3, {k: round(v, 3) for k, v in st.items()}), flush=True)
                del pipe
        for skip in (False, True):
            d = (frames[(skip, True)] - frames[(skip, False)]).abs().max()
            print('   max |fused - separate| (tile-skip %d) = %.3e;  frame max %.3f' % (skip, float(d), float(frames[(skip, False)].max())))
        print('   max |skip - noskip| fused = %.3e' % float((frames[(True, True)] - frames[(False, True)]).abs().max()))


if __name__ == '__main__':
    main()
