"""HIP-graph replay of RNRPipeline.render vs eager launches: frames/s at V views per step (V = 1: the reference's mode)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'relightable-nr_amd'))
import torch
import bench
from rnr_amd import scene
from rnr_amd.pipeline import RNRPipeline

class A: pass
args = A(); args.img_size = 512; args.nf0 = 64; args.tex_ch = 24
sc = bench.build_scene(args)
dev = torch.device('cuda:0')
for V in (1, 8):
    for prec in ('f32', 'f16x3'):
        pipe = RNRPipeline(sc['mesh'], 512, sc['textures'], sc['unet_sd'], sc['pivots_spec'], sc['pivots_diff'], None, nf0=64,
                           max_views=V, device=dev, sh_coeff=sc['sh_coeff'], sh_lmax=10, skip_background_tiles=False,
                           precision=prec)
        ids = [(37 * i) % 720 for i in range(V)]
        poses = {k: torch.from_numpy(x).to(dev) for k, x in scene.spiral_views(512, ids).items()}
        static = {k: v.clone() for k, v in poses.items()}
        for _ in range(3):
            ref = pipe.render(static['proj'], static['pose'], static['proj_inv'], static['R_inv']).clone()
        torch.cuda.synchronize()
        print('warm ok', flush=True)
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            pipe.render(static['proj'], static['pose'], static['proj_inv'], static['R_inv'])
        torch.cuda.current_stream().wait_stream(s)
        with torch.cuda.graph(g):
            out = pipe.render(static['proj'], static['pose'], static['proj_inv'], static['R_inv'])
        print('captured', flush=True)
        g.replay(); torch.cuda.synchronize()
        print('replayed', flush=True)
        print('V=%d %s  graph vs eager max diff %.3e' % (V, prec, (out - ref).abs().max().item()))

        def timeit(fn, n=40):
            for _ in range(5): fn()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(n): fn()
            torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
        print('diff ok', flush=True)
        te = 1.0
        if os.environ.get('NO_EAGER') != '1':
            te = timeit(lambda: pipe.render(static['proj'], static['pose'], static['proj_inv'], static['R_inv']))
        print('eager timed', te, flush=True)
        tg = timeit(g.replay, int(os.environ.get('N_REPLAY', '40')))
        print('V=%d %s  eager %.3f ms/step (%.1f frames/s)   graph %.3f ms/step (%.1f frames/s)' % (V, prec, te, V / te * 1e3, tg, V / tg * 1e3))
        del g, pipe
