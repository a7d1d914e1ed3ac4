"""nr.vertices_to_faces / nr.vertex_attrs_to_faces (reference: vertices_to_faces.py:4-46): pure index gathers.
(The fused hot path does this gather inside the face-setup kernel; these are the API-level equivalents.)"""
import torch


def _gather(attrs, faces):
    if faces.shape[0] == 1 and attrs.shape[0] != 1:
        faces = faces.expand(attrs.shape[0], -1, -1)
    assert attrs.ndimension() == 3 and faces.ndimension() == 3
    assert attrs.shape[0] == faces.shape[0] and faces.shape[2] == 3
    bs, nv = attrs.shape[:2]
    idx = faces.long() + (torch.arange(bs, device=attrs.device) * nv)[:, None, None]
    return attrs.reshape(bs * nv, attrs.shape[2])[idx]


def vertices_to_faces(vertices, faces):
    """[B,nv,3], [B or 1,nf,3] -> [B,nf,3,3]"""
    assert vertices.shape[2] == 3
    return _gather(vertices, faces)


def vertex_attrs_to_faces(vertex_attrs, faces):
    """[B,nv,A], [B,nf,3] -> [B,nf,3,A]"""
    return _gather(vertex_attrs, faces)
