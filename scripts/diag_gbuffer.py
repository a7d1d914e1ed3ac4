import sys, os, numpy as np, torch
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,'relightable-nr_amd'))
from oracle import rnr_oracle as orc
from rnr_amd import ops
g=np.load(os.path.join(ROOT,'tests/golden/rasterizer_module64.npz'))
dev='cuda:0'
mesh = ops.DeviceMesh(g['buf_vertices'][0], g['mesh_vt'], g['buf_vertices_normals'][0], g['mesh_f_v_idx'], g['mesh_f_vt_idx'], g['mesh_f_vn_idx'], dev)
S=64
proj=torch.from_numpy(g['proj']); pose=torch.from_numpy(g['pose'])
i=0
v_cpu = orc.projection(torch.from_numpy(g['buf_vertices']), proj[i:i+1], pose[i:i+1,:3,:3], pose[i:i+1,:3,3][:,None,:], torch.zeros(1,5), S)
gb = ops.rasterize_gbuffer(mesh, v_cpu.contiguous().to(dev), pose[i:i+1].to(dev), S)
mesh_t={k:torch.from_numpy(g['mesh_'+k]) for k in ['v','vt','vn','f_v_idx','f_vt_idx','f_vn_idx']}
mesh_t['v']=torch.from_numpy(g['buf_vertices'][0]); mesh_t['vn']=torch.from_numpy(g['buf_vertices_normals'][0])
o=orc.rasterizer_forward(mesh_t, proj[i:i+1], pose[i:i+1], S)
ref=g['view0_weight_map'][0][...,0]
got=gb['weight_map'][0].cpu().numpy()
orw=o['weight_map'][0,...,0].numpy()
d=np.abs(got-ref); j=np.unravel_index(d.argmax(), d.shape)
print('max diff', d.max(), 'at', j, 'got', got[j[:2]], 'ref', ref[j[:2]], 'oracle', orw[j[:2]])
print('raw hip', gb['raw_weight_map'][0].cpu().numpy()[j[:2]], 'raw oracle', o['raw_weight_map'][0].numpy()[j[:2]])
print('depth hip', gb['depth'][0].cpu().numpy()[j[:2]], 'oracle', o['depth'][0].numpy()[j[:2]], 'ref', g['view0_depth'][0][j[:2]])
fi=int(gb['face_index_map'][0].cpu().numpy()[j[:2]]); print('face', fi, 'ref face', g['view0_face_index_map'][0][j[:2]])
print('faces z oracle', o['faces_v_uvz'][0,fi,:,2].numpy())
print('diff oracle vs ref', np.abs(orw-ref).max(), 'hip vs oracle', np.abs(got-orw).max())
print('raw diff bits', (gb['raw_weight_map'][0].cpu().numpy().view(np.uint32)!=o['raw_weight_map'][0].numpy().view(np.uint32)).sum())
print('depth diff bits', (gb['depth'][0].cpu().numpy().view(np.uint32)!=o['depth'][0,...,0].numpy().view(np.uint32)).sum())
