#!/bin/bash
# usage: scripts/pmc.sh TAG "COUNTERS A B C" ["COUNTERS ..."]...   one rocprofv3 --pmc pass per counter group
# (never combined with trace domains other than --kernel-trace); results merged by scripts/pmc_merge.py
TAG=$1; shift
export VIEWS=${VIEWS:-8} PRECISION=${PRECISION:-f32} PMC_STEPS=2 PMC_WARMUP=1 CONV_ALGO=${CONV_ALGO:-winograd4}
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
i=0
for grp in "$@"; do
  timeout 400 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/p$i -- python $GRAFT_REPO_ROOT/bench.py --conv-algo $CONV_ALGO --views-per-step $VIEWS --steps $PMC_STEPS --warmup $PMC_WARMUP --no-cpu-baseline --no-parity --main-loop-only --no-calibration --prewarm-seconds 0 --windows 1 --precision $PRECISION > $OUT/p$i.log 2>&1
  i=$((i+1))
done
python $GRAFT_REPO_ROOT/scripts/pmc_merge.py $OUT
