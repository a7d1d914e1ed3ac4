"""nr.save_obj / create_texture_image on the HIP atlas kernels (neural_renderer.cuda.create_texture_image).

Same outputs as the reference module (neural_renderer/save_obj.py:10-82): `name.obj` with per-corner `vt` indices,
`name.mtl` pointing at `name.png`, the atlas holding one right-triangle tile per face.  Written from scratch: the tile
corners come from one numpy expression, the files are assembled as text blocks; PNG output uses PIL."""
import os

import numpy as np
import torch

_MATERIAL = 'material_1'


def _tile_grid(num_faces):
    """Tiles per row / per column of the atlas (save_obj.py:12-13)."""
    per_row = int((num_faces - 1.) ** 0.5) + 1
    return per_row, int((num_faces - 1.) / per_row) + 1


def _tile_corners(num_faces, per_row, size):
    """[nf, 3, 2] pixel coordinates of each face's tile triangle: top-left, bottom-left, bottom-right of the tile
    (save_obj.py:16-24)."""
    k = np.arange(num_faces)
    left, top = (k % per_row) * size, (k // per_row) * size
    right, bottom = left + size - 1, top + size - 1
    return np.stack([np.stack([left, top], -1), np.stack([left, bottom], -1), np.stack([right, bottom], -1)],
                    1).astype(np.float32)


def create_texture_image(textures, texture_size_out=16):
    """textures [nf, ts, ts, ts, 3] -> (atlas [H, W, 3] float numpy with row 0 at the bottom like an image file,
    vt [nf, 3, 2] in [0, 1])."""
    import neural_renderer.cuda.create_texture_image as ext
    nf = textures.shape[0]
    per_row, rows = _tile_grid(nf)
    corners = torch.from_numpy(_tile_corners(nf, per_row, texture_size_out)).cuda()
    atlas = torch.zeros(rows * texture_size_out, per_row * texture_size_out, 3, device='cuda')
    ext.create_texture_image(corners, textures.detach().float().contiguous().cuda(), atlas, 1e-5)
    vt = corners.cpu().numpy() / np.array([atlas.shape[1] - 1, atlas.shape[0] - 1], np.float32)
    return atlas.cpu().numpy()[::-1], vt


def _obj_text(name, vertices, faces, vt, mtl_name):
    blocks = ['# %s\n#\n' % name]
    if vt is not None:
        blocks.append('mtllib %s\n' % mtl_name)
    blocks.append(''.join('v %.8f %.8f %.8f\n' % tuple(p) for p in vertices))
    if vt is None:
        blocks.append(''.join('f %d %d %d\n' % tuple(f + 1) for f in faces))
    else:
        blocks.append(''.join('vt %.8f %.8f\n' % tuple(p) for p in vt.reshape(-1, 2)))
        corner = 3 * np.arange(len(faces))[:, None] + np.arange(1, 4)[None, :]          # one vt per face corner
        rows = np.stack([faces + 1, corner], -1).reshape(len(faces), 6)
        blocks.append('usemtl %s\n' % _MATERIAL + ''.join('f %d/%d %d/%d %d/%d\n' % tuple(r) for r in rows))
    return '\n'.join(blocks) + '\n'


def save_obj(filename, vertices, faces, textures=None):
    """vertices [nv, 3], faces [nf, 3] (0-based), optional textures [nf, ts, ts, ts, 3]."""
    assert vertices.ndimension() == 2 and faces.ndimension() == 2
    stem = filename[:-4]
    vt = None
    if textures is not None:
        from PIL import Image
        atlas, vt = create_texture_image(textures)
        Image.fromarray((np.clip(atlas, 0, 1) * 255).round().astype(np.uint8)).save(stem + '.png')
        with open(stem + '.mtl', 'w') as fh:
            fh.write('newmtl %s\nmap_Kd %s\n' % (_MATERIAL, os.path.basename(stem + '.png')))
    with open(filename, 'w') as fh:
        fh.write(_obj_text(os.path.basename(filename), vertices.detach().cpu().numpy(), faces.detach().cpu().numpy(), vt,
                           os.path.basename(stem + '.mtl')))
