#!/bin/bash
# usage: scripts/abl.sh VARIANT...   ("" = in-tree build); prints frames/s and U-Net TFLOP/s per variant
for v in "$@"; do
  if [ "$v" != "base" ]; then export RNR_HIP_LIB=$PWD/build_abl/librnr_$v.so; else unset RNR_HIP_LIB; fi
  python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-parity 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['value'],1), round(d['roofline']['achieved'],1), round(d['single_view_mode']['ms_per_frame'],3))"
done
