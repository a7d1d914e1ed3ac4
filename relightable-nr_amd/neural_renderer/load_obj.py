"""nr.load_obj (reference: neural_renderer/load_obj.py:108-209), `load_texture=False` path only.

One pass over the file instead of the reference's four; same outputs: float32 v / vn / vt and 0-based int32
f_v_idx / f_vn_idx / f_vt_idx, triangles with `v/vt/vn` triplets (load_obj.py:168-175)."""
import numpy as np
import torch


def load_obj(filename_obj, normalization=True, texture_size=4, load_texture=False, texture_wrapping='REPEAT',
             use_bilinear=True, use_cuda=True):
    if load_texture:
        raise NotImplementedError('load_obj(load_texture=True) is out of scope of the hot-path build (SURVEY.md §2.1)')
    v, vn, vt, fv, fvt, fvn = [], [], [], [], [], []
    with open(filename_obj) as fh:
        for line in fh:
            tok = line.split()
            if not tok:
                continue
            key = tok[0]
            if key == 'v':
                v.append((float(tok[1]), float(tok[2]), float(tok[3])))
            elif key == 'vn':
                vn.append((float(tok[1]), float(tok[2]), float(tok[3])))
            elif key == 'vt':
                vt.append((float(tok[1]), float(tok[2])))
            elif key == 'f':
                parts = [p.split('/') for p in tok[1:]]
                fv.append([int(p[0]) for p in parts])
                if len(parts[0]) > 1 and parts[0][1] != '':
                    fvt.append([int(p[1]) for p in parts])
                if len(parts[0]) > 2:
                    fvn.append([int(p[-1]) for p in parts])
    dev = 'cuda' if use_cuda else 'cpu'
    f32 = lambda a, w: torch.from_numpy(np.asarray(a, np.float32).reshape(-1, w)).to(dev)
    i32 = lambda a: (torch.from_numpy(np.asarray(a, np.int32).reshape(-1, 3)) - 1).to(dev)
    vertices = f32(v, 3)
    if normalization:   # load_obj.py:196-201
        vertices = vertices - vertices.min(0)[0][None, :]
        vertices = vertices / torch.abs(vertices).max()
        vertices = vertices * 2
        vertices = vertices - vertices.max(0)[0][None, :] / 2
    v_attr = {'v': vertices, 'vn': f32(vn, 3) if vn else [], 'vt': f32(vt, 2) if vt else []}
    f_attr = {'f_v_idx': i32(fv), 'f_vn_idx': i32(fvn), 'f_vt_idx': i32(fvt)}
    return v_attr, f_attr
