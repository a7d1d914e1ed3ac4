#!/bin/bash
# usage: scripts/abl_v.sh VIEWS VARIANT...
V=$1; shift
for v in "$@"; do
  if [ "$v" != "base" ]; then export RNR_HIP_LIB=$PWD/build_abl/librnr_$v.so; else unset RNR_HIP_LIB; fi
  python bench.py --views-per-step $V --steps 10 --warmup 2 --no-cpu-baseline --no-parity --main-loop-only 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('V=$V $v', round(d['value'],1), round(d['roofline']['achieved'],1))"
done
