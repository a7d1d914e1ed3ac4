"""Diagnosis of the drop-in loop's host time in the numpy-contract mode (free running: where does the HOST block?)."""
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

def run(tag, n=20):
    for i in range(5):
        loop.view(*pose(i))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(5, 5 + n):
        loop.view(*pose(i))
    torch.cuda.synchronize()
    print('%-60s %.2f ms per view' % (tag, (time.perf_counter() - t0) / n * 1e3), flush=True)

def table(tag):
    acc = {}
    for i in range(5, 25):
        ht = []
        loop.view(*pose(i), host_times=ht)
        for (_, t0), (name, t1) in zip(ht[:-1], ht[1:]):
            acc.setdefault(name, []).append((t1 - t0) * 1e3)
    tab = {k: round(float(np.median(v)), 3) for k, v in acc.items()}
    tab['SUM'] = round(float(sum(tab.values())), 3)
    print(tag, json.dumps(tab, indent=1), flush=True)

run('V0 numpy contract (pinned f64 D2H)')
loop.drain = False
table('host time per stage, NOT drained:')
loop.drain = True
orig = sph_harm.evaluate_sh_basis
def r04_form(lmax=0, azi=None, pol=None, directions=None, device=None, as_tensor=False):
    d = torch.as_tensor(np.asarray(directions, np.float32)).contiguous().to(dev)
    return ops.sh_basis(d, int(lmax)).cpu().numpy().astype(np.float64)
sph_harm.evaluate_sh_basis = r04_form
run('V1 r04 form (.cpu().numpy().astype(float64))')
_buf = {}
def persistent(lmax=0, azi=None, pol=None, directions=None, device=None, as_tensor=False):
    d = torch.as_tensor(np.ascontiguousarray(directions, dtype=np.float32)).to(dev)
    out = ops.sh_basis(d, int(lmax))
    if out.shape not in _buf:
        _buf[out.shape] = torch.empty(out.shape, dtype=torch.float64, pin_memory=True)
    h = _buf[out.shape]; h.copy_(out); return h.numpy()
sph_harm.evaluate_sh_basis = persistent
run('V2 one persistent pinned block')
def with_sync(*a, **k):
    torch.cuda.synchronize(); return orig(*a, **k)
sph_harm.evaluate_sh_basis = with_sync
run('V3 V0 + synchronize before')
sph_harm.evaluate_sh_basis = orig
loop.sh_on_device = True
run('V4 as_tensor=True')
loop.sh_on_device = False
os.environ['X'] = '1'
torch.cuda.empty_cache()
run('V0 again after empty_cache')
print(torch.cuda.memory_summary(abbreviated=True)[:1500])
