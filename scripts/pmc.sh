#!/bin/bash
# usage: scripts/pmc.sh TAG "COUNTERS A B C" ["COUNTERS ..."]...   one rocprofv3 --pmc pass per counter group
# (never combined with trace domains other than --kernel-trace); results merged by scripts/pmc_merge.py
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
i=0
for grp in "$@"; do
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/p$i -- python $GRAFT_REPO_ROOT/bench.py --views-per-step ${VIEWS:-4} --steps 2 --warmup 1 --no-cpu-baseline --no-parity --main-loop-only --precision ${PRECISION:-f32} > $OUT/p$i.log 2>&1
  i=$((i+1))
done
python $GRAFT_REPO_ROOT/scripts/pmc_merge.py $OUT
