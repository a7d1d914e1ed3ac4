#!/bin/bash
# usage (GPU box, repo root): scripts/v1_probe.sh TAG — one view per call (the reference's mode): per-layer times, kernel trace with
# timestamps (gaps between dependent launches), kernel statistics
TAG=${1:-r03}
ROOT=$PWD
OUT=$ROOT/gpurun_out/v1_$TAG
mkdir -p $OUT
python scripts/layer_time.py --views 1 > $OUT/layer_time_f32_views1.txt 2>&1
python scripts/layer_time.py --views 2 > $OUT/layer_time_f32_views2.txt 2>&1
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python $ROOT/bench.py --steps 20 --warmup 3 \
    --views-per-step 1 --no-cpu-baseline --main-loop-only > $OUT/kt.log 2>&1
find $OUT/kt -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats_views1.csv
find $OUT/kt -name "*kernel_trace.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_trace_views1.csv
tail -1 $OUT/kt.log > $OUT/bench_line_under_profiler_views1.json
rm -rf $OUT/kt
cd $ROOT
python bench.py --steps 40 --warmup 5 --views-per-step 1 --no-cpu-baseline --main-loop-only > $OUT/bench_views1.log 2>&1
tail -1 $OUT/bench_views1.log > $OUT/bench_views1.json
ls -la $OUT
