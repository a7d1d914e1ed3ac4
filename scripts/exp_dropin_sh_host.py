"""Host wall time of the statements of the numpy SH contract inside the drop-in view loop (test_rnr.py:324-328), device drained at
every boundary: where the 1.6 ms per view go."""
import os, sys, time, tempfile, collections
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
    loop = DropinViewLoop(obj, 512, sc['textures'], sc['unet_sd'], sc['sh_coeff'], nf0=64, device=dev, sh_on_device=False)
pv = {k: torch.from_numpy(v).to(dev) for k, v in scene.spiral_views(512, np.arange(60)).items()}
pose = lambda i: (pv['proj'][i:i + 1], pv['pose'][i:i + 1], pv['proj_inv'][i:i + 1], pv['R_inv'][i:i + 1])
for i in range(5): loop.view(*pose(i))
acc = collections.OrderedDict()
n = 40
for i in range(n):
    ht = []
    loop.view(*pose(i), host_times=ht)
    for (a, ta), (b, tb) in zip(ht[:-1], ht[1:]):
        acc[b] = acc.get(b, 0.0) + (tb - ta)
for k, v in acc.items():
    print('%-36s %.3f ms' % (k, v / n * 1e3))
print('sum %.3f ms' % (sum(acc.values()) / n * 1e3))
