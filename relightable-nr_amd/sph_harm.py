"""Drop-in `sph_harm` (reference: sph_harm.py) without pyshtools: the basis is evaluated by the HIP kernel
(real, orthonormal, no Condon-Shortley phase, columns l = 0..lmax, m = -l..l)."""
import os

import numpy as np
import torch

from rnr_amd import ops


def cart2sph(x, y, z):
    """sph_harm.py:6-19."""
    if type(x) is torch.Tensor:
        return torch.atan2(y, x), torch.atan2(z, torch.sqrt(x ** 2 + y ** 2)), torch.sqrt(x ** 2 + y ** 2 + z ** 2)
    return np.arctan2(y, x), np.arctan2(z, np.sqrt(x ** 2 + y ** 2)), np.sqrt(x ** 2 + y ** 2 + z ** 2)


def sph2cart(azimuth, elevation, r):
    """sph_harm.py:22-38."""
    m = torch if type(azimuth) is torch.Tensor else np
    ce = m.cos(elevation)
    return r * ce * m.cos(azimuth), r * ce * m.sin(azimuth), r * m.sin(elevation)


_PINNED_MAX_BYTES = int(float(os.environ.get('RNR_SH_PINNED_MAX_MB', '64')) * (1 << 20))


def evaluate_sh_basis(lmax=0, azi=None, pol=None, directions=None, device=None, as_tensor=False):
    """sph_harm.py:41-71.  directions [n,3] (numpy or tensor) or azi/pol in degrees -> np.ndarray [n,(lmax+1)^2]
    (float64 container like the reference; values carry float32 precision, which is what every caller casts to).
    Runs on the device of `directions` when that is a GPU tensor, else on `device` (default: the current GPU).
    as_tensor=True (not in the reference): return the float32 DEVICE tensor instead — the per-view loop of test_rnr.py:322-328
    then needs no host round trip (`sph_harm.evaluate_sh_basis(lmax=2, directions=view_dir_map.reshape(-1, 3), as_tensor=True)`
    in place of the `.cpu().detach().numpy()` ... `torch.from_numpy(...).to(device)` pair).
    The numpy result is written by ONE device -> pinned-host copy (converted to float64 on the device): the reference's contract
    costs a 19 MB transfer per 512 x 512 view (0.4 ms), not a pageable copy plus a host-side cast."""
    if directions is None:
        a, p = np.deg2rad(np.asarray(azi, np.float64)), np.deg2rad(np.asarray(pol, np.float64))
        directions = np.stack([np.sin(p) * np.cos(a), np.sin(p) * np.sin(a), np.cos(p)], -1)
    if torch.is_tensor(directions) and directions.is_cuda:
        d = directions.detach().float().contiguous()
    else:
        dev = torch.device(device) if device is not None else torch.device('cuda', torch.cuda.current_device())
        src = directions.detach().cpu().numpy() if torch.is_tensor(directions) else directions
        d = torch.as_tensor(np.ascontiguousarray(src, dtype=np.float32)).to(dev)
    out = ops.sh_basis(d, int(lmax))
    if as_tensor:
        return out
    # cast on the DEVICE first, then a same-dtype D2H copy: `host.copy_(out)` with float32 -> float64 across devices takes
    # torch's slow conversion path (measured in the drop-in loop at 512^2: 15.6 ms per call against 0.37 ms this way,
    # scripts/exp_dropin_host2.py)
    out64 = out.double()
    nbytes = out64.numel() * 8
    host = host32 = None
    if nbytes <= _PINNED_MAX_BYTES:
        try:
            host = torch.empty(out.shape, dtype=torch.float64, pin_memory=True)     # blocks of torch's caching pinned allocator
            host32 = torch.empty(out.shape, dtype=torch.float32, pin_memory=True)
        except RuntimeError:                                                        # page-locking refused (ulimit -l, fragmentation)
            host = host32 = None
    if host is None:
        return out64.cpu().numpy()          # pageable copy: slower, nothing stays page-locked
    host.copy_(out64, non_blocking=True)
    host32.copy_(out, non_blocking=True)    # the values as the GPU computed them, for SHBasisArray.astype(np.float32)
    torch.cuda.current_stream(out.device).synchronize()
    # The array ALIASES the page-locked block and keeps it alive: a caller that stores many results (precompute-style caches of
    # per-view bases) should store `result.copy()` — torch's caching host allocator recycles the block once the array is gone
    # but never returns it to the OS.  Results above RNR_SH_PINNED_MAX_MB (default 64) take the pageable path above.
    return SHBasisArray._wrap(host.numpy(), host32.numpy())


class SHBasisArray(np.ndarray):
    """What `evaluate_sh_basis` returns: the float64 ndarray of the reference's contract (sph_harm.py:41-71) with ONE call answered
    from a cache — the conversion the reference's view loop applies next (test_rnr.py:324),

        sph_harm.evaluate_sh_basis(...).reshape((N, H, W, -1)).astype(np.float32)

    The basis is computed in float32 on the GPU and the float64 container holds exactly those values, so `.astype(np.float32)` of
    the array (or of a C-contiguous reshape of it) equals the float32 block that came down with it bit for bit; returning that block
    saves the host-side conversion of 2.4 M doubles per 512 x 512 view, and because the block is page-locked the script's
    `torch.from_numpy(...).to(device)` that follows is a direct DMA instead of a staged pageable copy (together 0.3 - 0.5 ms of the
    1.2 ms this call costs in the loop).  The float64 array is READ-ONLY while it carries the cache (an in-place edit would make the
    cached block stale); `.copy()` gives an ordinary writable array.  Every other operation is numpy's own and returns plain
    arrays / views.  `SHBasisArray.stats` counts how often the cache answered (`fast`) and how often numpy converted (`plain`);
    RNR_SH_FAST_ASTYPE=0 turns the cache off."""

    stats = {'fast': 0, 'plain': 0}
    _f32 = None

    @classmethod
    def _wrap(cls, a64, a32):
        if os.environ.get('RNR_SH_FAST_ASTYPE', '1') == '0':
            return a64
        obj = a64.view(cls)
        obj._f32 = [a32]        # one holder shared by every same-size view: the block is handed out once
        obj.setflags(write=False)
        return obj

    def __array_finalize__(self, obj):
        # the cache follows same-size views only (reshape / view): slices, copies and results of arithmetic drop it
        f32 = getattr(obj, '_f32', None)
        if f32 is not None and f32[0] is not None and self.base is not None and self.size == f32[0].size and self.dtype == np.float64:
            self._f32 = f32
        else:
            self._f32 = None

    def astype(self, dtype, order='K', casting='unsafe', subok=True, copy=True):
        holder = self._f32
        if (holder is not None and holder[0] is not None and np.dtype(dtype) == np.float32 and order in ('K', 'C', 'A')
                and self.flags.c_contiguous and not self.flags.writeable):
            SHBasisArray.stats['fast'] += 1
            out = holder[0].reshape(self.shape)
            holder[0] = None                    # handed out once: the caller owns (and may modify) the block now
            return out
        SHBasisArray.stats['plain'] += 1
        return np.asarray(self).astype(dtype, order=order, casting=casting, subok=subok, copy=copy)

    def __array_wrap__(self, arr, context=None, return_scalar=False):
        return np.asarray(arr) if arr.ndim else arr[()]


def fit_sh_coeff(samples, sh_basis_val):
    """sph_harm.py:74-88.  samples [ns,C] or [L,ns,C], basis [ns,nb] -> [nb,C] or [L,nb,C]."""
    if not torch.is_tensor(samples):
        w = 4.0 * np.pi / samples.shape[-2]
        if samples.ndim == 2:
            return (samples[:, None, :] * sh_basis_val[:, :, None]).sum(-3) * w
        return (samples[:, :, None, :] * sh_basis_val[None, :, :, None]).sum(-3) * w
    b = sh_basis_val.float().contiguous()
    if samples.dim() == 2:
        return ops.sh_fit(samples.float().contiguous(), b)
    return torch.stack([ops.sh_fit(s.float().contiguous(), b) for s in samples])


def reconstruct_sh(sh_coeff, sh_basis_val):
    """sph_harm.py:91-102.  coeff [nb,C] or [L,nb,C], basis [ns,nb] -> [ns,C] or [L,ns,C]."""
    if not torch.is_tensor(sh_coeff):
        if sh_coeff.ndim == 2:
            return (sh_basis_val[..., None] * sh_coeff[None, :]).sum(-2)
        return (sh_basis_val[None, :, :, None] * sh_coeff[:, None, :, :]).sum(-2)
    b = sh_basis_val.float().contiguous()
    if sh_coeff.dim() == 2:
        return ops.sh_reconstruct(b, sh_coeff.float().contiguous())
    return torch.stack([ops.sh_reconstruct(b, c.float().contiguous()) for c in sh_coeff])
