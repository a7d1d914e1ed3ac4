#!/bin/bash
# usage: scripts/emu.sh PRECISION[:VARIANT]...   frames/s and U-Net ms per 8-view step; VARIANT = build_abl/librnr_VARIANT.so (default: in-tree)
for pv in "$@"; do
  p=${pv%%:*}; v=${pv#*:}; [ "$v" = "$pv" ] && v=base
  if [ "$v" != "base" ]; then export RNR_HIP_LIB=$PWD/build_abl/librnr_$v.so; else unset RNR_HIP_LIB; fi
  python bench.py --steps 10 --warmup 2 --no-cpu-baseline --main-loop-only --precision $p 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$p $v', round(d['value'],1), 'frames/s, U-Net', round(d['roofline']['stage_ms_per_step'],2), 'ms/step')"
done
