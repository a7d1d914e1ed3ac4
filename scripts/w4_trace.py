"""Per-workgroup timeline of a convolution kernel from a diagnostic build that writes wall-clock marks (100 MHz) in wave 0 of every workgroup:
  conv_wino4_kernel      scripts/experiments/conv_wino4_trace_fastdiv_r05.diff, scripts/mkvariant.sh trace "-DW4_TRACE"     --layers 1,2,8  [--fine]
  conv_wino2p_kernel<2>  scripts/experiments/conv_wino2p_trace_r05.diff,        scripts/mkvariant.sh trace2p "-DW2P_TRACE"  --layers 14,16,18,20
  conv_wino80_kernel     the same five marks in conv_wino80.inc (-DW80_TRACE)                                                --layers 22
then  RNR_HIP_LIB=$PWD/build_abl/librnr_trace.so python scripts/w4_trace.py --layers 2,8 --views 16.
Marks: 0 start, 1 first weight / halo loads requested, 2 ... landed, 3 first image staged (barrier), 4 K loop done, 5 plane rows
exchanged, 6 statistics added and stores issued (kernels without marks 1 / 2 write 1 = 2 = 3).  Prints the mean length of each phase, how
far apart the workgroups of one CU start, the idle gap between two workgroups of a CU (meaningful with one workgroup per CU) and how
synchronised the CUs are.  Output of r05: profiles/r05_wino4_tile_timeline.txt."""
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

PHASES = ['index math -> loads out', 'first loads in flight', 'first image -> barrier', 'K loop', 'plane-row exchange', 'statistics + stores']


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--views', type=int, default=16)
    ap.add_argument('--layers', default='2,8')
    ap.add_argument('--fine', action='store_true', help='the W4_TRACE build of conv_wino4_kernel also marks the inside of its prologue')
    a = ap.parse_args()
    L = _lib.load()
    L.rnr_debug_w4_trace.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    flags = _lib.EMU_FLAGS['f32'] | _lib.CONV_WINOGRAD | _lib.CONV_WINOGRAD4
    for idx, kind, H, cins, cout in LAYERS:
        if idx not in [int(x) for x in a.layers.split(',')]:
            continue
        us, _ = time_layer(L, idx, kind, H, cins, cout, a.views, flags, 5)
        torch.cuda.synchronize()
        buf = np.zeros(16384 * 16, dtype=np.uint64)
        assert L.rnr_debug_w4_trace(buf.ctypes.data, buf.size) == 0
        t = buf.reshape(-1, 16)
        keep = t[:, 0] >= t[:, 0].max() - np.uint64(int(us * 100 * 1.5))        # the last launch only (the buffer keeps older marks)
        wg = np.nonzero(keep)[0]
        t = t[keep]
        nwg = len(t)
        hw = t[:, 7]
        xcc = (hw >> np.uint64(32)) & np.uint64(0xf)
        hwid = hw & np.uint64(0xffffffff)
        cu = (hwid >> np.uint64(8)) & np.uint64(0xf)
        sh = (hwid >> np.uint64(12)) & np.uint64(0x1)
        se = (hwid >> np.uint64(13)) & np.uint64(0x7)
        cuid = ((xcc * 8 + se) * 2 + sh) * 16 + cu
        ts = t[:, :7].astype(np.int64)
        base = ts[:, 0].min()
        ts = (ts - base) * 0.01                  # us
        print('L%d  %d^2 %s -> %d, %d views: %.1f us per launch, %d workgroups on %d CUs' % (
            idx, H, '+'.join(map(str, cins)), cout, a.views, us, nwg, len(np.unique(cuid))))
        d = np.diff(ts, axis=1)
        for i, name in enumerate(PHASES):
            print('   %-24s mean %7.2f us   p10 %7.2f  p90 %7.2f' % (name, d[:, i].mean(), np.percentile(d[:, i], 10), np.percentile(d[:, i], 90)))
        print('   %-24s mean %7.2f us' % ('workgroup total', (ts[:, 6] - ts[:, 0]).mean()))
        if a.fine:
            fine = (t[:, [0, 8, 9, 10, 11, 1]].astype(np.int64) - base) * 0.01
            print('   inside "index math": tile_coords %.2f, tile origin + staging item %.2f, weight descriptor etc. %.2f, chunk_src %.2f, load requests %.2f us' %
                  tuple(np.diff(fine, axis=1).mean(0)))
            lr = np.diff(fine, axis=1)[:, 4]
            print('   load requests: first workgroup of a CU (nothing before it) %.2f us, later ones %.2f us' % (lr[wg < 256].mean(), lr[wg >= 256].mean()))
        gaps, periods = [], []
        for c in np.unique(cuid):
            w = np.where(cuid == c)[0]
            w = w[np.argsort(ts[w, 0])]
            gaps += list(ts[w[1:], 0] - ts[w[:-1], 6])
            periods += list(ts[w[1:], 0] - ts[w[:-1], 0])
        gaps, periods = np.array(gaps), np.array(periods)
        print('   tile period on a CU      mean %7.2f us   idle gap between workgroups of a CU: mean %.2f us (p10 %.2f, p90 %.2f)' % (
            periods.mean(), gaps.mean(), np.percentile(gaps, 10), np.percentile(gaps, 90)))
        # synchronisation: start phase of every workgroup modulo the tile period, circular spread (1 = all CUs in lockstep, 0 = uniform)
        ph = 2 * np.pi * (ts[:, 0] % periods.mean()) / periods.mean()
        print('   lockstep index of the workgroup starts: %.2f' % abs(np.exp(1j * ph).mean()))


if __name__ == '__main__':
    main()
