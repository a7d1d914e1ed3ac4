import os, sys, time, json, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'relightable-nr_amd')]
import numpy as np, torch
import sph_harm
from rnr_amd import ops, scene
from rnr_amd.view_loop import DropinViewLoop
import bench
dev = torch.device('cuda:0')
args = bench.parse([])
sc = bench.build_scene(args)
with tempfile.TemporaryDirectory() as td:
    obj = os.path.join(td, 'm.obj'); scene.write_obj(obj, sc['mesh'])
    loop = DropinViewLoop(obj, 512, sc['textures'], sc['unet_sd'], sc['sh_coeff'], nf0=64, device=dev)
pv = {k: torch.from_numpy(v).to(dev) for k, v in scene.spiral_views(512, np.arange(60)).items()}
pose = lambda i: (pv['proj'][i:i + 1], pv['pose'][i:i + 1], pv['proj_inv'][i:i + 1], pv['R_inv'][i:i + 1])
T = {}
def tick(name, t0):
    T.setdefault(name, []).append((time.perf_counter() - t0) * 1e3)
def timed_eval(mode):
    def f(lmax=0, azi=None, pol=None, directions=None, device=None, as_tensor=False):
        t0 = time.perf_counter()
        d = torch.as_tensor(np.ascontiguousarray(directions, dtype=np.float32)).to(dev)
        tick('e:upload', t0); t0 = time.perf_counter()
        out = ops.sh_basis(d, int(lmax))
        tick('e:kernel launch', t0); t0 = time.perf_counter()
        if mode == 'pinned':
            host = torch.empty(out.shape, dtype=torch.float64, pin_memory=True)
            tick('e:pinned alloc', t0); t0 = time.perf_counter()
            o64 = out.double()
            tick('e:double() launch', t0); t0 = time.perf_counter()
            host.copy_(o64)
            tick('e:copy_ (blocking D2H)', t0); t0 = time.perf_counter()
            r = host.numpy()
        elif mode == 'pinned32':
            host = torch.empty(out.shape, dtype=torch.float32, pin_memory=True)
            host.copy_(out)
            tick('e:copy_ f32 (blocking D2H)', t0); t0 = time.perf_counter()
            r = host.numpy().astype(np.float64)
            tick('e:host astype f64', t0); t0 = time.perf_counter()
        else:
            c = out.cpu()
            tick('e:.cpu()', t0); t0 = time.perf_counter()
            r = c.numpy().astype(np.float64)
            tick('e:host astype f64', t0); t0 = time.perf_counter()
        return r
    return f
class TimedArr:
    pass
orig_astype = None
def run(tag, n=20):
    T.clear()
    for i in range(5):
        loop.view(*pose(i))
    T.clear()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(5, 5 + n):
        loop.view(*pose(i))
    torch.cuda.synchronize()
    print('%-50s %.2f ms per view   ' % (tag, (time.perf_counter() - t0) / n * 1e3), {k: round(float(np.median(v)), 3) for k, v in T.items()}, flush=True)
for mode in ('pinned', 'pinned32', 'pageable', 'pinned'):
    sph_harm.evaluate_sh_basis = timed_eval(mode)
    run(mode)
