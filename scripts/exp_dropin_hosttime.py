"""Is the drop-in view loop (device variant) host-bound?  Host time to ENQUEUE n views (no synchronize inside) vs the wall time including
the final drain, plus a cProfile of the enqueue."""
import os, sys, time, tempfile, cProfile, pstats
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'relightable-nr_amd')]
import numpy as np, torch
import bench
from rnr_amd import scene
from rnr_amd.view_loop import DropinViewLoop
dev = torch.device('cuda:0')
args = bench.parse([]); sc = bench.build_scene(args)
with tempfile.TemporaryDirectory() as td:
    obj = os.path.join(td, 'm.obj'); scene.write_obj(obj, sc['mesh'])
    loop = DropinViewLoop(obj, 512, sc['textures'], sc['unet_sd'], sc['sh_coeff'], nf0=64, device=dev, sh_on_device=True)
pv = {k: torch.from_numpy(v).to(dev) for k, v in scene.spiral_views(512, np.arange(120)).items()}
pose = lambda i: (pv['proj'][i:i + 1], pv['pose'][i:i + 1], pv['proj_inv'][i:i + 1], pv['R_inv'][i:i + 1])
for i in range(10): loop.view(*pose(i))
torch.cuda.synchronize()
n = 100
t0 = time.perf_counter()
for i in range(n): loop.view(*pose(i))
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print('host enqueue %.3f ms per view, wall %.3f ms per view' % (t_host / n * 1e3, t_all / n * 1e3))
pr = cProfile.Profile(); pr.enable()
for i in range(30): loop.view(*pose(i))
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('tottime').print_stats(18)
