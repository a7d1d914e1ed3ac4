"""bench.py's single_view_mode block alone (one view per call over the spiral, sequential and with calls in flight).
usage: python scripts/exp_single_view.py [views]"""
import json
import os
import sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'relightable-nr_amd'))
import bench  # noqa: E402

args = bench.parse(['--no-cpu-baseline', '--single-views', sys.argv[1] if len(sys.argv) > 1 else '720'])
dev = torch.device('cuda:0')
sc = bench.build_scene(args)
r = bench.single_view_block(sc, args, dev)
print(json.dumps({k: r[k] for k in ('frames_per_s', 'ms_per_frame')} | {'unet_ms': r['roofline']['stage_ms_per_view'],
                 'two': r['two_calls_in_flight']['frames_per_s'], 'three': r['three_calls_in_flight']['frames_per_s']}))
