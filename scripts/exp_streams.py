"""Experiment: S pipelines of V/S views each on separate HIP streams (tail overlap between kernels of different streams)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'relightable-nr_amd'))
import numpy as np, torch
import bench
from rnr_amd import scene
from rnr_amd.pipeline import RNRPipeline

class A: pass
args = A(); args.img_size = 512; args.nf0 = 64; args.tex_ch = 24
sc = bench.build_scene(args)
dev = torch.device('cuda:0')
for S, V in [(1, 8), (2, 8), (4, 8), (2, 16), (1, 16)]:
    v = V // S
    pipes = [RNRPipeline(sc['mesh'], 512, sc['textures'], sc['unet_sd'], sc['pivots_spec'], sc['pivots_diff'], None,
                         nf0=64, max_views=v, device=dev, sh_coeff=sc['sh_coeff'], sh_lmax=10) for _ in range(S)]
    streams = [torch.cuda.Stream() for _ in range(S)]
    steps, warm = 10, 2
    ids = (np.arange((steps + warm) * V) * 7) % 720
    poses = {k: torch.from_numpy(x).to(dev) for k, x in scene.spiral_views(512, ids).items()}
    def step(s):
        cur = torch.cuda.current_stream()
        for i, (p, st) in enumerate(zip(pipes, streams)):
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                lo = s * V + i * v
                sl = slice(lo, lo + v)
                p.render(poses['proj'][sl], poses['pose'][sl], poses['proj_inv'][sl], poses['R_inv'][sl])
        for st in streams:
            cur.wait_stream(st)
    for s in range(warm): step(s)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for s in range(warm, warm + steps): step(s)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print('streams %d x %d views: %.1f frames/s' % (S, v, steps * V / dt), flush=True)
    del pipes
    torch.cuda.empty_cache()
