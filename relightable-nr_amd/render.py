"""Drop-in `render` (reference: render.py): the functions the inference path uses."""
import os
import warnings

import numpy as np
import torch

from rnr_amd import ops


def spherical_mapping(l_dir):
    """render.py:87-93.  [3,...] -> [2,...] equirect uv (y-up)."""
    return torch.stack((torch.atan2(l_dir[2], l_dir[0]) * 0.5 / np.pi + 0.5, torch.acos(l_dir[1]) * 1.0 / np.pi), dim=0)


def spherical_mapping_batch(l_dir):
    """render.py:96-102.  [N,3,...] -> [N,2,...]."""
    return torch.stack((torch.atan2(l_dir[:, 2], l_dir[:, 0]) * 0.5 / np.pi + 0.5, torch.acos(l_dir[:, 1]) * 1.0 / np.pi),
                       dim=1)


def spherical_mapping_inv(lp_samples_uv):
    """render.py:105-121.  uv [2,n] -> unit directions [3,n] (init-time helper of LightingSH)."""
    y = torch.cos(lp_samples_uv[1] * np.pi)
    s = (1 - y ** 2).sqrt()
    a = lp_samples_uv[0] * 2 - 1
    x = s * torch.cos(a * np.pi)
    z = s * torch.sin(a * np.pi)
    z = z * ((~(a == 1.0)).to(s.dtype) * 2 - 1)
    z = z * ((~(a == -1.0)).to(s.dtype) * 2 - 1)
    return torch.nn.functional.normalize(torch.stack((x, y, z), dim=0), dim=0)


class TBNMap(torch.Tensor):
    """What `get_TBN_map` returns: the [N,H,W,3,3] tensor itself (same storage, same behaviour) with ONE overridden call — the
    batched product the reference's view loop runs next (test_rnr.py:314),

        torch.matmul(TBN_map.reshape((-1, 3, 3)).transpose(-2, -1), view_dir_map.reshape((-1, 3, 1)))

    which torch hands to rocBLAS as 262 144 batched 3 x 3 GEMMs (1.9 ms per 512 x 512 view, the second-largest item of the
    drop-in loop).  A [P,3,3] view of a TBNMap (plain or transposed in its last two dimensions) times a [P,3,1] float32 device
    tensor is answered by `rnr_tbn_matvec` (one launch, a few microseconds; fused multiply-adds, <= 1 ulp per term from the GEMM
    result).  Views (reshape / view / transpose / permute / indexing) keep the type; every other operation — and any matmul that
    does not have this exact form — is torch's own and returns plain tensors."""

    _VIEW_OPS = None
    # observability of the fast path: matmuls on a TBNMap answered by rnr_tbn_matvec ('hits') and those that fell through to
    # torch's batched GEMM ('misses': 1.97 ms per 512 x 512 view through rocBLAS); the first miss warns once per process
    stats = {'hits': 0, 'misses': 0, 'last_miss': None}
    _warned = False

    @classmethod
    def reset_stats(cls):
        cls.stats.update(hits=0, misses=0, last_miss=None)
        cls._warned = False

    @classmethod
    def _view_ops(cls):
        if cls._VIEW_OPS is None:
            T = torch.Tensor
            cls._VIEW_OPS = {T.reshape, torch.reshape, T.view, T.transpose, torch.transpose, T.permute, torch.permute,
                             T.__getitem__, T.detach, T.contiguous, T.float}
        return cls._VIEW_OPS

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        if func in (torch.matmul, torch.Tensor.matmul, torch.Tensor.__matmul__) and len(args) == 2 and not kwargs:
            with torch._C.DisableTorchFunctionSubclass():         # the checks below must not re-enter this hook per attribute
                out, why = _tbn_matvec(args[0], args[1])
            if out is not None:
                cls.stats['hits'] += 1
                return out
            if why is not None:
                cls.stats['misses'] += 1
                cls.stats['last_miss'] = why
                if not cls._warned:
                    cls._warned = True
                    warnings.warn('render.TBNMap: torch.matmul on a TBN map was NOT answered by rnr_tbn_matvec (%s); torch runs it '
                                  'as batched 3 x 3 GEMMs (about 2 ms per 512 x 512 view).  test_rnr.py:314 spells it '
                                  'torch.matmul(TBN_map.reshape((-1, 3, 3)).transpose(-2, -1), view_dir_map.reshape((-1, 3, 1))) '
                                  'on float32 device tensors; see render.TBNMap.stats' % why, RuntimeWarning, stacklevel=2)
        out = super().__torch_function__(func, types, args, kwargs)
        if func not in cls._view_ops() and isinstance(out, TBNMap):
            out = out.as_subclass(torch.Tensor)
        return out


def _tbn_matvec(a, b):
    """The overridden product -> (result, None), or (None, reason) when the operands do not have test_rnr.py:314's form
    (reason None: the override does not apply at all — opted out, or the TBN map is not the left operand)."""
    if not isinstance(a, TBNMap) or not isinstance(b, torch.Tensor):
        return None, None
    if os.environ.get('RNR_TBN_MATMUL', '1') == '0':
        return None, None
    if torch.is_autocast_enabled():
        return None, 'autocast is enabled: left to torch so that the autocast dtype rules apply'
    if not a.is_cuda:
        return None, None               # a CPU map: torch's product is the only one there is, nothing is lost
    if not (b.is_cuda and a.device == b.device):
        return None, 'operands are not on one GPU'
    if a.dtype != torch.float32 or b.dtype != torch.float32:
        return None, 'dtypes %s x %s, not float32' % (a.dtype, b.dtype)
    if a.dim() != 3 or b.dim() != 3 or tuple(a.shape[1:]) != (3, 3) or tuple(b.shape) != (a.shape[0], 3, 1) or a.shape[0] == 0:
        return None, 'shapes %s x %s, not [P,3,3] x [P,3,1]' % (tuple(a.shape), tuple(b.shape))
    if a.requires_grad or b.requires_grad:
        return None, 'an operand requires grad'
    if a.stride() == (9, 1, 3):
        transposed = True
    elif a.stride() == (9, 3, 1):
        transposed = False
    else:
        return None, 'left operand strides %s: neither the [P,3,3] records (9,3,1) nor their transpose (9,1,3)' % (tuple(a.stride()),)
    bt = b.as_subclass(torch.Tensor) if type(b) is not torch.Tensor else b
    if not bt.is_contiguous():
        return None, 'right operand is not contiguous'
    base = a.as_subclass(torch.Tensor).as_strided((a.shape[0], 3, 3), (9, 3, 1))      # the records as rnr_tbn_map wrote them
    return ops.tbn_matvec(base, bt.reshape(-1, 3), transposed=transposed).reshape(-1, 3, 1), None


def get_TBN_map(normal_map, face_index_map, faces_v=None, faces_texcoord=None, tangent=None, check_nan=False, plain=False):
    """render.py:124-168 -> [N,H,W,3,3].  `check_nan=True` restores the reference's NaN guards (3 host syncs per call,
    ValueError('nan value detected')); they are off by default to keep the stream asynchronous.  The result is a TBNMap (see
    there: one matmul form answered by one launch); `plain=True` — or RNR_TBN_MATMUL=0 in the environment — returns /
    behaves as a plain torch.Tensor for callers that compare bit for bit with torch's batched GEMM, use torch.compile, etc."""
    if tangent is None:
        assert faces_v is not None and faces_texcoord is not None
        tangent = ops.face_tangents(faces_v, faces_texcoord)
    else:
        tangent = torch.nn.functional.normalize(tangent.float(), dim=-1).contiguous()
    if check_nan and torch.isnan(tangent).sum() > 0:
        raise ValueError('nan value detected')
    tbn = ops.tbn_map(normal_map.float().contiguous(), face_index_map.int().contiguous(), tangent)
    if check_nan and torch.isnan(tbn).sum() > 0:
        raise ValueError('nan value detected')
    return tbn if plain else tbn.as_subclass(TBNMap)


def _out_of_scope(name):
    def f(*a, **k):
        raise NotImplementedError('render.%s is not used by the inference hot path (SURVEY.md §2.1)' % name)
    f.__name__ = name
    return f


interp_vertex_attr = _out_of_scope('interp_vertex_attr')
texture_mapping = _out_of_scope('texture_mapping')
lp_mapping = _out_of_scope('lp_mapping')
sample_light_dir = _out_of_scope('sample_light_dir')
get_TBN_map_perpixel = _out_of_scope('get_TBN_map_perpixel')
