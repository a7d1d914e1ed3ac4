import sys, os, numpy as np, torch
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,'relightable-nr_amd')); sys.path.insert(0, os.path.join(ROOT,'tests'))
from test_gpu_raster import run_hip_raster
for name in ['raster_soup64','raster_soup50','raster_soup64_nearfar','raster_sphere128']:
    g=np.load(os.path.join(ROOT,'tests/golden',name+'.npz'))
    r=run_hip_raster(g['faces'], int(g['image_size']), float(g['near']), float(g['far']))
    print(name, 'idx mismatches', int((r['face_index_map']!=g['face_index_map']).sum()))
    for k in ['faces_inv','weight_map','depth_map','face_inv_map']:
        a=r[k]; b=np.asarray(g[k]).reshape(a.shape)
        neq = a.view(np.uint32)!=b.view(np.uint32)
        bothnan = np.isnan(a)&np.isnan(b)
        real = neq & ~bothnan
        print('  ',k,'bit-diff',int(neq.sum()),'of which both-NaN',int((neq&bothnan).sum()),'real',int(real.sum()))
        if real.sum():
            ii=np.argwhere(real)[:8]
            for i in ii:
                i=tuple(i); print('     ',i,a[i],b[i], hex(a.view(np.uint32)[i]), hex(b.view(np.uint32)[i]))
