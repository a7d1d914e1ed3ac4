#!/bin/bash
# usage (GPU box, repo root): scripts/v1_trace.sh TAG [bench args] — kernel trace (timestamps) of the one-view-per-call loop
TAG=${1:-x}; shift
ROOT=$PWD
OUT=$ROOT/gpurun_out/v1_$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python $ROOT/bench.py --steps 20 --warmup 3 \
    --views-per-step 1 --no-cpu-baseline --main-loop-only --no-calibration "$@" > $OUT/kt.log 2>&1
find $OUT/kt -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats_views1.csv
find $OUT/kt -name "*kernel_trace.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_trace_views1.csv
tail -1 $OUT/kt.log > $OUT/bench_line_under_profiler_views1.json
rm -rf $OUT/kt
