"""Drop-in `misc` (reference: misc.py).  interpolate_bilinear runs the HIP sampler kernel."""
import numpy as np
import torch

from rnr_amd import ops


def interpolate_bilinear(data, sub_x, sub_y):
    """misc.py:5-42.  data [H,W,C]; sub_x, sub_y [...] -> [...,C] (device tensors)."""
    return ops.interpolate_bilinear(data.float().contiguous(), sub_x.float().contiguous(), sub_y.float().contiguous())


def interpolate_bilinear_np(data, sub_x, sub_y):
    """misc.py:45-73 (host numpy helper, not on the hot path): same tap / weight rules as the device sampler."""
    H, W = data.shape[:2]
    valid = ((sub_x >= 0) & (sub_x <= W - 1) & (sub_y >= 0) & (sub_y <= H - 1)).astype(data.dtype)
    x0 = np.floor(sub_x).astype(np.int64)
    y0 = np.floor(sub_y).astype(np.int64)
    x1 = np.clip(x0 + 1, 0, W - 1)
    y1 = np.clip(y0 + 1, 0, H - 1)
    x0 = np.clip(x0, 0, W - 1)
    y0 = np.clip(y0, 0, H - 1)
    i00, i10, i01, i11 = data[y0, x0], data[y1, x0], data[y0, x1], data[y1, x1]
    x0w = x0 - (x0 == x1)
    y0w = y0 - (y0 == y1)
    w00 = (x1 - sub_x) * (y1 - sub_y) * valid
    w10 = (x1 - sub_x) * (sub_y - y0w) * valid
    w01 = (sub_x - x0w) * (y1 - sub_y) * valid
    w11 = (sub_x - x0w) * (sub_y - y0w) * valid
    return i00 * w00[..., None] + i10 * w10[..., None] + i01 * w01[..., None] + i11 * w11[..., None]
