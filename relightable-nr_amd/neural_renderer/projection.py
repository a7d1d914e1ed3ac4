"""nr.projection (reference: neural_renderer/projection.py:6-53) on the HIP kernel."""
import torch

from rnr_amd import ops


def projection(vertices, K, R, t, dist_coeffs, orig_size, offset=None, scale=None, eps=1e-9):
    """vertices [B,nv,3] (B = 1 or N), K/R [N,3,3], t [N,1,3], dist_coeffs [N or 1,5] -> [N,nv,3] (u_ndc, v_ndc, z)."""
    N = max(vertices.shape[0], K.shape[0], R.shape[0])
    ex = lambda x, tail: x.expand((N,) + tail).contiguous() if x.shape[0] != N else x.contiguous()
    K, R = ex(K.float(), (3, 3)), ex(R.float(), (3, 3))
    tt = ex(t.float().reshape(t.shape[0], 3), (3,))
    dc = ex(dist_coeffs.float(), (5,)) if dist_coeffs is not None else None
    off = ex(offset.float(), (2,)) if (offset is not None and scale is not None) else None
    sc = ex(scale.float(), (2,)) if (offset is not None and scale is not None) else None
    if vertices.shape[0] == 1:
        return ops.project_vertices(vertices[0].float().contiguous(), K, R, tt, orig_size, dc, off, sc, eps)
    outs = []
    for b in range(N):   # per-view meshes: one launch each
        s = slice(b, b + 1)
        outs.append(ops.project_vertices(vertices[b].float().contiguous(), K[s], R[s], tt[s], orig_size,
                                         dc[s] if dc is not None else None, off[s] if off is not None else None,
                                         sc[s] if sc is not None else None, eps))
    return torch.cat(outs, 0)
