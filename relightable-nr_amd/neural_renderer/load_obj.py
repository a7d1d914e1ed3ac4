"""nr.load_obj / load_mtl / load_textures (reference: neural_renderer/load_obj.py:1-209).

One pass over the file instead of the reference's four; same outputs: float32 v / vn / vt and 0-based int32
f_v_idx / f_vn_idx / f_vt_idx, triangles with `v/vt/vn` triplets (load_obj.py:168-175).  With load_texture=True the
material colours / texture images of the .mtl are baked into per-face texture cubes by the HIP kernel behind
neural_renderer.cuda.load_textures (images are read with PIL; the reference uses skimage.io.imread)."""
import os

import numpy as np
import torch

texture_wrapping_dict = {'REPEAT': 0, 'MIRRORED_REPEAT': 1, 'CLAMP_TO_EDGE': 2, 'CLAMP_TO_BORDER': 3}


def load_mtl(filename_mtl):
    """colour (Kd) and texture file name (map_Kd) per material (load_obj.py:13-30)."""
    texture_filenames, colors, name = {}, {}, ''
    with open(filename_mtl) as fh:
        for line in fh:
            tok = line.split()
            if not tok:
                continue
            if tok[0] == 'newmtl':
                name = tok[1]
            elif tok[0] == 'map_Kd':
                texture_filenames[name] = tok[1]
            elif tok[0] == 'Kd':
                colors[name] = np.array(list(map(float, tok[1:4])))
    return colors, texture_filenames


def _imread(path):
    from PIL import Image
    return np.asarray(Image.open(path))


def load_textures(filename_obj, filename_mtl, texture_size, texture_wrapping='REPEAT', use_bilinear=True):
    """Per-face texture cubes [nf, ts, ts, ts, 3] on the GPU (load_obj.py:33-106): faces default to 0.5 grey, take
    their material's Kd colour, and are overwritten by the material's texture image where there is one."""
    import neural_renderer.cuda.load_textures as load_textures_cuda
    vt, faces, material_names, material = [], [], [], ''
    with open(filename_obj) as fh:
        for line in fh:
            tok = line.split()
            if not tok:
                continue
            if tok[0] == 'vt':
                vt.append([float(x) for x in tok[1:3]])
            elif tok[0] == 'f':
                idx = [int(p.split('/')[1]) if ('/' in p and '//' not in p) else 0 for p in tok[1:]]
                for i in range(len(idx) - 2):            # fan triangulation (load_obj.py:54-70)
                    faces.append((idx[0], idx[i + 1], idx[i + 2]))
                    material_names.append(material)
            elif tok[0] == 'usemtl':
                material = tok[1]
    vt = np.vstack(vt).astype(np.float32)
    faces = torch.from_numpy(vt[np.vstack(faces).astype(np.int32) - 1]).cuda().contiguous()      # [nf,3,2]
    colors, texture_filenames = load_mtl(filename_mtl)
    nf = faces.shape[0]
    textures = torch.full((nf, texture_size, texture_size, texture_size, 3), 0.5, dtype=torch.float32, device='cuda')
    names = np.array(material_names)
    for name, color in colors.items():
        sel = torch.from_numpy(names == name).cuda()
        textures[sel] = torch.from_numpy(color.astype(np.float32)).cuda()[None, None, None, None, :]
    for name, filename_texture in texture_filenames.items():
        image = _imread(os.path.join(os.path.dirname(filename_obj), filename_texture)).astype(np.float32) / 255.
        if image.ndim == 2:
            image = np.stack((image,) * 3, -1)
        if image.shape[2] == 4:
            image = image[:, :, :3]
        image = torch.from_numpy(image[::-1, :, :].copy()).cuda()
        is_update = torch.from_numpy((names == name).astype(np.int32)).cuda()
        textures = load_textures_cuda.load_textures(image, faces, textures, is_update,
                                                    texture_wrapping_dict[texture_wrapping], use_bilinear)
    return textures


def load_obj(filename_obj, normalization=True, texture_size=4, load_texture=False, texture_wrapping='REPEAT',
             use_bilinear=True, use_cuda=True):
    v, vn, vt, fv, fvt, fvn = [], [], [], [], [], []
    mtllib = None
    with open(filename_obj) as fh:
        for line in fh:
            tok = line.split()
            if not tok:
                continue
            key = tok[0]
            if key == 'mtllib':
                mtllib = tok[1]
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
    if load_texture:
        if mtllib is None:
            raise Exception('Failed to load textures.')
        textures = load_textures(filename_obj, os.path.join(os.path.dirname(filename_obj), mtllib), texture_size,
                                 texture_wrapping=texture_wrapping, use_bilinear=use_bilinear)
        return v_attr, f_attr, textures
    return v_attr, f_attr
