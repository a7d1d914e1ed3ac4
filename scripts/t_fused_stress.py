"""Race hunt for the in-launch BatchNorm finalise / split-K combine of rnr_conv2d_fused: the same poses rendered again
and again — one view per call (every layer one or two rounds of workgroups that finish together: the worst case for the
arrival tickets), 3 views per call, and two calls in flight — must give identical frames every time.  A stale statistics
read, a lost ticket or a slab read before it landed would show up as a frame that differs from its first rendering.
Usage (GPU box): python scripts/t_fused_stress.py [rounds]"""
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
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    args = bench.parse([])
    sc = bench.build_scene(args)
    dev = torch.device('cuda', 0)
    ids = (np.arange(48) * 15) % 720
    pv = {k: torch.from_numpy(v).to(dev) for k, v in scene.spiral_views(512, ids).items()}
    a = lambda sl: [pv[k][sl] for k in ('proj', 'pose', 'proj_inv', 'R_inv')]
    bad = 0
    worst = 0.0
    for V, inflight in [(1, 1), (3, 1), (1, 2), (2, 3)]:
        pipe = bench.make_pipeline(sc, args, dev, V, inflight=inflight, skip_background_tiles=False)
        first = {}
        n = 0
        for r in range(rounds):
            hs = []
            for lo in range(0, 48 - V + 1, V):
                hs.append((lo, pipe.submit(*a(slice(lo, lo + V)))))
                if len(hs) >= inflight:
                    lo0, h = hs.pop(0)
                    img = h.synchronize().clone()
                    n += 1
                    if lo0 not in first:
                        first[lo0] = img
                    else:
                        d = float((img - first[lo0]).abs().max())
                        worst = max(worst, d)
                        bad += d != 0.0
            for lo0, h in hs:
                h.synchronize()
        print('views/call %d, calls in flight %d: %d frames-groups rendered, %d differ from their first rendering (max |d| %.3e)'
              % (V, inflight, n, bad, worst), flush=True)
        del pipe
    print('OK' if bad == 0 else 'MISMATCH (float64 atomics may legitimately reorder; anything above 1e-6 is a race)')
    return 0 if worst < 1e-6 else 1


if __name__ == '__main__':
    sys.exit(main())
