"""Where conv_wino4p_kernel (the persistent F(4x4, 3x3) kernel) spends a tile's time: a diagnostic build (-DW4P_TRACE:
scripts/mkvariant.sh trace4p "-DW4P_TRACE") sums, per workgroup, the wall-clock time (100 MHz) wave 0 spent in each phase of its
tile loop.  Usage (GPU box): RNR_HIP_LIB=$PWD/build_abl/librnr_trace4p.so python scripts/w4p_trace.py --layers 1,2,8 --views 16
Prints the mean over workgroups of (phase time / tiles) in us."""
import argparse
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'scripts'))
sys.path.insert(0, os.path.join(ROOT, 'relightable-nr_amd'))
import torch  # noqa: E402

from emu_layer_table import LAYERS  # noqa: E402
from layer_time import time_layer  # noqa: E402
from rnr_amd import _lib  # noqa: E402

PHASES = ['end barrier -> loop top (tile_of)', 'setup + first weight requests + BN table', 'wait raw image + barrier', 'raw -> T + barrier',
          'K loop', 'next tile DMA requests', 'A^T M A', 'statistics', 'BN arrival (vmcnt0 + barrier + ticket)', 'stores issued',
          'BN completion']


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--views', type=int, default=16)
    ap.add_argument('--layers', default='1,2,8')
    a = ap.parse_args()
    L = _lib.load()
    L.rnr_debug_w4p_trace.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    flags = _lib.EMU_FLAGS['f32'] | _lib.CONV_WINOGRAD | _lib.CONV_WINOGRAD4
    for idx, kind, H, cins, cout in LAYERS:
        if idx not in [int(x) for x in a.layers.split(',')]:
            continue
        us, _ = time_layer(L, idx, kind, H, cins, cout, a.views, flags, 5)
        torch.cuda.synchronize()
        buf = np.zeros(1024 * 16, dtype=np.uint64)
        assert L.rnr_debug_w4p_trace(buf.ctypes.data, buf.size) == 0
        t = buf.reshape(-1, 16)[:256].astype(np.float64)
        tiles = t[:, 12]
        ok = tiles > 0
        per = t[ok, :11] / tiles[ok, None] * 0.01
        print('L%d: %.1f us per launch, %d workgroups, %.1f tiles each; per tile (us): total %.2f' % (idx, us, ok.sum(), tiles[ok].mean(), per.sum(1).mean()))
        for i, name in enumerate(PHASES):
            print('    %-48s %6.2f  (min %5.2f max %5.2f over workgroups)' % (name, per[:, i].mean(), per[:, i].min(), per[:, i].max()))


if __name__ == '__main__':
    main()
