#!/bin/bash
# usage: scripts/stage.sh VARIANT...  -> per-stage ms of bench.py for each library variant (base = in-tree build); only the headline
# loop and the per-stage extras run (RNR_BENCH_FAST=1 skips the emulation / stream / single-view reports)
for v in "$@"; do
  if [ "$v" != "base" ]; then export RNR_HIP_LIB=$PWD/build_abl/librnr_$v.so; else unset RNR_HIP_LIB; fi
  RNR_BENCH_FAST=1 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-parity 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['value'],1), {k: round(v['ms_per_step'],3) for k,v in d['stages'].items()})"
done
