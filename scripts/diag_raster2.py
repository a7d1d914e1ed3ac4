import sys, os, numpy as np, torch
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,'relightable-nr_amd')); sys.path.insert(0, os.path.join(ROOT,'tests'))
from test_gpu_raster import run_hip_raster
from oracle import raster as oras
for (S,nf,seed) in [(17,50,1),(33,1,4)]:
    rng = np.random.RandomState(seed)
    f = rng.uniform(-1.5, 1.5, size=(2, nf, 3, 3)).astype(np.float32)
    f[..., 2] = rng.uniform(0.2, 9.0, size=(2, nf, 3))
    small = rng.rand(2, nf) < 0.7
    c = f[:, :, :1, :2].copy()
    f[..., :2] = np.where(small[..., None, None], c + (f[..., :2] - c) * 0.03, f[..., :2])
    if nf > 10:
        f[0, 3, 1] = f[0, 3, 0]; f[0, 4, 2, :2] = f[0, 4, 0, :2] * 0.25 + f[0, 4, 1, :2] * 0.75
        f[1, 5, :, :2] *= 1e4; f[1, 6, 0, 0] = np.nan; f[0, 7, :, 2] = [1e-30, 2.0, 3.0]
    g = oras.face_index_map(f, S, 0.0, 1e5)
    r = run_hip_raster(f, S, 0.0, 1e5)
    d = np.argwhere(r['face_index_map']!=g['face_index_map'])
    print(S,nf,'mismatch count',len(d)); print(d[:40].tolist())
    for i in d[:6]:
        i=tuple(i); print(i, 'hip',r['face_index_map'][i], r['depth_map'][i], 'ora', g['face_index_map'][i], g['depth_map'][i])
