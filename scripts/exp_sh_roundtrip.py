"""Where the time of test_rnr.py:322-328 goes at 512 x 512 (the SH-basis host round trip of the drop-in loop): each step timed
with a device synchronize around it, 20 repetitions, median ms.  Run on the GPU box: python scripts/exp_sh_roundtrip.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'relightable-nr_amd')]
import numpy as np, torch
import sph_harm
from rnr_amd import ops

dev = torch.device('cuda:0')
S = 512
g = torch.Generator().manual_seed(0)
vd = torch.nn.functional.normalize(torch.randn(1, S, S, 3, generator=g), dim=-1).to(dev)


def med(f, n=20):
    ts = []
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter(); r = f(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    return float(np.median(ts)), r

out = {}
out['user: view_dir_map.reshape(-1,3).cpu().detach().numpy()'], d_np = med(lambda: vd.reshape((-1, 3)).cpu().detach().numpy())
out['ours: upload directions (pageable H2D 3 MB)'], d_dev = med(lambda: torch.as_tensor(np.ascontiguousarray(d_np, dtype=np.float32)).to(dev))
out['ours: rnr_sh_basis kernel'], b = med(lambda: ops.sh_basis(d_dev, 2))
def pinned():
    h = torch.empty(b.shape, dtype=torch.float64, pin_memory=True); h.copy_(b); return h.numpy()
out['ours: device f64 cast + D2H into a pinned block (19 MB)'], h64 = med(pinned)
out['r04 form: .cpu().numpy().astype(float64)'], _ = med(lambda: b.cpu().numpy().astype(np.float64))
def pinned32():
    h = torch.empty(b.shape, dtype=torch.float32, pin_memory=True); h.copy_(b); return h.numpy().astype(np.float64)
out['alt: D2H f32 pinned + host astype(float64)'], _ = med(pinned32)
out['ours: whole evaluate_sh_basis(numpy in, numpy out)'], sh64 = med(lambda: sph_harm.evaluate_sh_basis(lmax=2, directions=d_np))
out['user: .reshape(1,S,S,9).astype(np.float32)'], sh32 = med(lambda: sh64.reshape((1, S, S, -1)).astype(np.float32))
pageable64 = np.array(sh64)
out['user: same astype from a PAGEABLE float64 array'], _ = med(lambda: pageable64.reshape((1, S, S, -1)).astype(np.float32))
out['user: torch.from_numpy(sh).to(device) (pageable H2D 9.4 MB)'], _ = med(lambda: torch.from_numpy(sh32).to(dev))
out['ours: evaluate_sh_basis(..., as_tensor=True) on the device tensor'], _ = med(lambda: sph_harm.evaluate_sh_basis(lmax=2, directions=vd.reshape((-1, 3)), as_tensor=True))
tbn = torch.randn(1, S, S, 3, 3, device=dev)
out['user: torch.matmul(TBN^T, view_dir) + normalize (test_rnr.py:314-315)'], _ = med(lambda: torch.nn.functional.normalize(torch.matmul(tbn.reshape((-1, 3, 3)).transpose(-2, -1), vd.reshape((-1, 3, 1)))[..., 0].reshape(vd.shape), dim=-1))
import json
print(json.dumps(out, indent=1))

# ---- the same statements INSIDE the drop-in loop at the bench configuration: host wall clock per stage (device drained at
# every boundary), median over 20 views
import tempfile
from rnr_amd import scene
from rnr_amd.view_loop import DropinViewLoop
sys.path.insert(0, ROOT)
import bench
args = bench.parse([])
sc = bench.build_scene(args)
with tempfile.TemporaryDirectory() as td:
    obj = os.path.join(td, 'm.obj'); scene.write_obj(obj, sc['mesh'])
    loop = DropinViewLoop(obj, 512, sc['textures'], sc['unet_sd'], sc['sh_coeff'], nf0=64, device=dev)
pv = {k: torch.from_numpy(v).to(dev) for k, v in scene.spiral_views(512, np.arange(40)).items()}
pose = lambda i: (pv['proj'][i:i + 1], pv['pose'][i:i + 1], pv['proj_inv'][i:i + 1], pv['R_inv'][i:i + 1])
for i in range(5):
    loop.view(*pose(i))
acc = {}
for i in range(5, 25):
    ht = []
    loop.view(*pose(i), host_times=ht)
    for (_, t0), (name, t1) in zip(ht[:-1], ht[1:]):
        acc.setdefault(name, []).append((t1 - t0) * 1e3)
tab = {k: float(np.median(v)) for k, v in acc.items()}
tab['SUM'] = float(sum(tab.values()))
print(json.dumps({'host_ms_per_stage_drained': tab}, indent=1))
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(5, 25):
    loop.view(*pose(i))
torch.cuda.synchronize(); print('numpy contract, free running: %.2f ms per view' % ((time.perf_counter() - t0) / 20 * 1e3))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for i in range(5, 15):
    loop.view(*pose(i))
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(25)
