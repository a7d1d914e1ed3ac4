#!/bin/bash
# usage: scripts/emu.sh VARIANT...  frames/s and U-Net ms with --precision bf16x6 for each library variant
for v in "$@"; do
  if [ "$v" != "base" ]; then export RNR_HIP_LIB=$PWD/build_abl/librnr_$v.so; else unset RNR_HIP_LIB; fi
  python bench.py --steps 10 --warmup 2 --no-cpu-baseline --main-loop-only --precision bf16x6 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['value'],1), round(d['roofline']['stage_ms_per_step'],2))"
done
