#!/bin/bash
# usage (GPU box, from the repo root): scripts/profile_round.sh TAG      e.g. TAG=r04
# Collects everything profiles/ holds for a round into gpurun_out/profile_TAG/:
#   kernel-trace statistics of the bench loop (16 views per step = the headline batch since r04, and 1 view per step = the reference's
#   calling mode, product path = Winograd convolutions; 16 views per step with direct convolutions and with the two emulated precisions), the PMC passes
#   (separate runs per counter group, never combined with other trace domains) of the product path at both batch sizes,
#   per-layer timing tables (both algorithms), the accuracy tables of the Winograd and emulation kernels, and the bench line
#   itself (roofline.traffic taken from THIS run's PMC files, for the headline and for single_view_mode).
TAG=${1:-r04}
ROOT=$PWD
OUT=$ROOT/gpurun_out/profile_$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
# LITE=1: the product path only (the direct / F(2x2)-only / emulation kernels did not change in the round)
CFGS="f32:16:winograd4 f32:1:winograd4 f32:16:winograd f32:16:direct f16x3:16:direct bf16x6:16:direct"
[ "$LITE" = 1 ] && CFGS="f32:16:winograd4 f32:1:winograd4"
for cfg in $CFGS; do
  prec=${cfg%%:*}; rest=${cfg#*:}; v=${rest%%:*}; algo=${rest##*:}
  name=$prec; [ $prec = f32 ] && [ $algo = direct ] && name=f32_direct; [ $prec = f32 ] && [ $algo = winograd ] && name=f32_f2x2only
  steps=5; [ $v = 1 ] && steps=20
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_${name}_$v -- python $ROOT/bench.py --steps $steps --warmup 2 \
      --views-per-step $v --no-cpu-baseline --main-loop-only --no-calibration --prewarm-seconds 0 --windows 1 --precision $prec --conv-algo $algo > $OUT/kt_${name}_$v.log 2>&1
  find $OUT/kt_${name}_$v -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/${TAG}_bench_kernel_stats_${name}_steps${steps}_views$v.csv
  [ $v = 1 ] && find $OUT/kt_${name}_$v -name "*kernel_trace.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_trace_views1.csv
  grep "^{\"metric\"" $OUT/kt_${name}_$v.log | tail -1 > $OUT/${TAG}_bench_line_under_profiler_${name}_views$v.json
  rm -rf $OUT/kt_${name}_$v
done
cd $ROOT
python scripts/trace_frame.py $OUT/kernel_trace_views1.csv > $OUT/${TAG}_frame_timeline_f32_views1.txt 2>&1
rm -f $OUT/kernel_trace_views1.csv
SQ1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"
SQ2="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS"
for v in 16 1; do
  CONV_ALGO=winograd4 PRECISION=f32 VIEWS=$v scripts/pmc.sh ${TAG}_f32_$v "FETCH_SIZE" "WRITE_SIZE" "$SQ1" "$SQ2" "TCC_HIT_sum TCC_MISS_sum" > $OUT/pmc_f32_$v.log 2>&1
  cp gpurun_out/pmc_${TAG}_f32_$v/merged.json $OUT/${TAG}_pmc_per_kernel_f32_steps2_views$v.json
done
# bench.py takes single_view_mode's roofline.traffic from the newest committed views-1 profile: make this run's the newest
cp $OUT/${TAG}_pmc_per_kernel_f32_steps2_views1.json $ROOT/profiles/
for v in 1 8 16; do
  timeout 300 python scripts/layer_time.py --views $v --winograd4 > $OUT/${TAG}_layer_time_f32_views$v.txt 2>&1
  [ "$LITE" = 1 ] && continue
  timeout 300 python scripts/layer_time.py --views $v --winograd > $OUT/${TAG}_layer_time_f32_f2x2only_views$v.txt 2>&1
  timeout 300 python scripts/layer_time.py --views $v > $OUT/${TAG}_layer_time_f32_direct_views$v.txt 2>&1
done
RNR_WINO_MIN_WGS=1 RNR_WINO2_MIN_WGS=1 timeout 600 python scripts/wino_check.py --views 2 > $OUT/${TAG}_winograd_accuracy.txt 2>&1
RNR_WINO4_MIN_WGS=1 RNR_WINO_MIN_WGS=1 RNR_WINO2_MIN_WGS=1 timeout 600 python scripts/wino_check.py --views 2 --f4x4 > $OUT/${TAG}_winograd_f4x4_accuracy.txt 2>&1
timeout 900 python scripts/wino_frames_720.py winograd4 > $OUT/${TAG}_winograd4_vs_direct_720views.json 2> $OUT/wf4.err
timeout 900 python bench.py --pmc-file $OUT/${TAG}_pmc_per_kernel_f32_steps2_views16.json > $OUT/bench.log 2>&1
grep "^{\"metric\"" $OUT/bench.log | tail -1 > $OUT/${TAG}_bench_final.json
ls -la $OUT
