#!/bin/bash
# usage: scripts/single.sh [ENV=VAL ...]   frames/s at 8 views per step and ms per frame at one view per call (the reference's mode)
for e in "$@"; do
  env RNR_BENCH_FAST=1 RNR_BENCH_SINGLE=1 $e python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-parity 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$e', round(d['value'],1), 'frames/s @8; single view', round(d['single_view_mode']['ms_per_frame'],3), 'ms =', round(d['single_view_mode']['frames_per_s'],1), 'frames/s')"
done
