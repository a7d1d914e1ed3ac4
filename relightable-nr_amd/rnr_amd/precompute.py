"""G-buffer export with the HIP rasterizer: the `.mat` products of the reference's
precompute.py:140-257, same directory names, file names and keys, so that the reference's training scripts can read
maps produced on ROCm.  PNG previews and ground-truth image copies (cv2) are not written.
"""
import os

import numpy as np
import scipy.io
import torch

import camera
import render
import sph_harm


def _save(out_dir, sub, img_fn, **arrs):
    d = os.path.join(out_dir, sub)
    os.makedirs(d, exist_ok=True)
    scipy.io.savemat(os.path.join(d, img_fn + '.mat'), arrs)


def export_view_maps(rasterizer, view, out_dir, img_fn, only_mesh_related=False):
    """rasterizer: drop-in `network.Rasterizer` on a GPU; view: one `ViewDataset` record with a leading batch dim of 1
    (proj, pose, proj_inv, R_inv, proj_orig, dist_coeffs).  Writes <out_dir>/<kind>/<img_fn>.mat like precompute.py."""
    dev = rasterizer.vertices.device
    g = lambda k: view[k].to(dev)
    n = lambda t: t.detach().cpu().numpy()
    with torch.no_grad():
        (uv_map, alpha_map, face_index_map, weight_map, faces_v_idx, normal_map, normal_map_cam, faces_v, faces_vt,
         position_map, position_map_cam, depth, v_uvz, v_front_mask) = rasterizer(
            proj=g('proj'), pose=g('pose'), dist_coeffs=g('dist_coeffs') if 'dist_coeffs' in view else None, offset=None,
            scale=None)
        _save(out_dir, 'raster', img_fn, face_index_map=n(face_index_map[0]), weight_map=n(weight_map[0]),
              faces_v_idx=n(faces_v_idx[0]), v_uvz=n(v_uvz[0]), v_front_mask=n(v_front_mask[0]))      # precompute.py:160-165
        if only_mesh_related:
            return
        tbn = render.get_TBN_map(normal_map, face_index_map, faces_v=faces_v[0], faces_texcoord=faces_vt[0])
        _save(out_dir, 'TBN_map', img_fn, TBN_map=n(tbn[0]))
        _save(out_dir, 'pose', img_fn, pose=n(g('pose')[0]), proj_orig=n(view['proj_orig'][0]) if 'proj_orig' in view else n(g('proj')[0]))
        _save(out_dir, 'proj', img_fn, proj=n(g('proj')[0]))
        _save(out_dir, 'uv_map', img_fn, uv_map=n(uv_map[0]))
        _save(out_dir, 'normal_map', img_fn, normal_map=n(normal_map[0]))
        z_out = np.array([1, -1, -1], dtype=np.float32)                                               # precompute.py:204
        _save(out_dir, 'normal_map_cam', img_fn, normal_map_cam=n(normal_map_cam[0]) * z_out)
        _save(out_dir, 'position_map', img_fn, position_map=n(position_map[0]))
        _save(out_dir, 'position_map_cam', img_fn, position_map_cam=n(position_map_cam[0]))
        vd, vd_cam = camera.get_view_dir_map(uv_map.shape[1:3], g('proj_inv'), g('R_inv'))
        _save(out_dir, 'view_dir_map', img_fn, view_dir_map=n(vd[0]))
        _save(out_dir, 'view_dir_map_cam', img_fn, view_dir_map_cam=n(vd_cam[0]) * z_out)
        vt = torch.matmul(tbn.reshape((-1, 3, 3)).transpose(-2, -1), vd.reshape((-1, 3, 1)))[..., 0].reshape(vd.shape)
        _save(out_dir, 'view_dir_map_tangent', img_fn, view_dir_map_tangent=n(torch.nn.functional.normalize(vt, dim=-1)[0]))
        sh = sph_harm.evaluate_sh_basis(lmax=2, directions=n(vd.reshape(-1, 3))).reshape(vd.shape[1:3] + (-1,)).astype(np.float32)
        _save(out_dir, 'sh_basis_map', img_fn, sh_basis_map=sh)
        refl = camera.get_reflect_dir(vd, normal_map) * alpha_map[..., None]                         # precompute.py:244
        _save(out_dir, 'reflect_dir_map', img_fn, reflect_dir_map=n(refl[0]))


def export_dataset(rasterizer, view_dataset, out_dir, only_mesh_related=False):
    """All views of a `dataio.ViewDataset` (buffered or not); file names 00000.mat, 00001.mat, ..."""
    for i in range(len(view_dataset)):
        v = view_dataset[i][0]
        view = {k: (t[None] if torch.is_tensor(t) else t) for k, t in v.items()}
        export_view_maps(rasterizer, view, out_dir, str(i).zfill(5), only_mesh_related)
