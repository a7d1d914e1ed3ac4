"""nr.load_obj / load_mtl / load_textures (reference: neural_renderer/load_obj.py:1-209).

One native pass over the file bytes (csrc/objparse.hip behind the C ABI) instead of the reference's four Python passes;
same outputs: float32 v / vn / vt and 0-based int32 f_v_idx / f_vn_idx / f_vt_idx, triangles (load_obj.py:168-175).  With load_texture=True the
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


def parse_obj_bytes(data):
    """OBJ text (bytes) -> numpy arrays via the native reader of librnr_hip.so (rnr_obj_scan / rnr_obj_parse,
    include/rnr_hip.h §4): v [nv,3], vn [nvn,3], vt [nvt,2] float32; f_v_idx, f_vt_idx, f_vn_idx [nf,3] int32 0-based
    (index arrays of absent attributes are empty)."""
    import ctypes
    from rnr_amd import _lib
    L = _lib.load()
    cnt = _lib.RnrObjCounts()
    _lib.check(L.rnr_obj_scan(data, len(data), ctypes.byref(cnt)))
    nv, nvn, nvt, nf = cnt.num_vertices, cnt.num_normals, cnt.num_texcoords, cnt.num_faces
    v, vn, vt = np.empty((nv, 3), np.float32), np.empty((nvn, 3), np.float32), np.empty((nvt, 2), np.float32)
    fv = np.empty((nf, 3), np.int32)
    fvt = np.empty((nf if nvt else 0, 3), np.int32)
    fvn = np.empty((nf if nvn else 0, 3), np.int32)
    ptr = lambda a: a.ctypes.data_as(ctypes.c_void_p) if a.size else None
    try:
        _lib.check(L.rnr_obj_parse(data, len(data), ctypes.byref(cnt), ptr(v), ptr(vn), ptr(vt), ptr(fv), ptr(fvt), ptr(fvn)))
    except _lib.RnrError as e:
        raise ValueError(str(e)) from None          # the reference's parser raises ValueError on malformed numbers
    return v, vn, vt, fv, fvt, fvn


def load_obj(filename_obj, normalization=True, texture_size=4, load_texture=False, texture_wrapping='REPEAT',
             use_bilinear=True, use_cuda=True):
    """load_obj.py:108-209.  Geometry comes from the native one-pass reader (no Python loop over lines)."""
    with open(filename_obj, 'rb') as fh:
        data = fh.read()
    v, vn, vt, fv, fvt, fvn = parse_obj_bytes(data)
    mtllib = None
    if load_texture:
        for line in data.split(b'\n'):
            if line.startswith(b'mtllib'):
                mtllib = line.split()[1].decode()
    dev = 'cuda' if use_cuda else 'cpu'
    f32 = lambda a: torch.from_numpy(a).to(dev)
    vertices = f32(v)
    if normalization:   # load_obj.py:196-201
        vertices = vertices - vertices.min(0)[0][None, :]
        vertices = vertices / torch.abs(vertices).max()
        vertices = vertices * 2
        vertices = vertices - vertices.max(0)[0][None, :] / 2
    v_attr = {'v': vertices, 'vn': f32(vn) if len(vn) else [], 'vt': f32(vt) if len(vt) else []}
    f_attr = {'f_v_idx': f32(fv), 'f_vn_idx': f32(fvn), 'f_vt_idx': f32(fvt)}
    if load_texture:
        if mtllib is None:
            raise Exception('Failed to load textures.')
        textures = load_textures(filename_obj, os.path.join(os.path.dirname(filename_obj), mtllib), texture_size,
                                 texture_wrapping=texture_wrapping, use_bilinear=use_bilinear)
        return v_attr, f_attr, textures
    return v_attr, f_attr
