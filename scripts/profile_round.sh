#!/bin/bash
# usage (GPU box, from the repo root): scripts/profile_round.sh TAG      e.g. TAG=r02
# Collects everything profiles/ holds for a round into gpurun_out/profile_TAG/:
#   kernel-trace statistics of the bench loop for the three conv precisions, the PMC passes (separate runs per counter
#   group, never combined with other trace domains) at the bench batch size, the per-layer accuracy table of the
#   emulation kernels, and the bench line itself (with roofline.traffic taken from THIS run's PMC file).
TAG=${1:-r02}
ROOT=$PWD
OUT=$ROOT/gpurun_out/profile_$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
for prec in f32 f16x3 bf16x6; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_$prec -- python $ROOT/bench.py --steps 5 --warmup 2 \
      --no-cpu-baseline --main-loop-only --precision $prec > $OUT/kt_$prec.log 2>&1
  find $OUT/kt_$prec -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/${TAG}_bench_kernel_stats_${prec}_steps5_views8.csv
  tail -1 $OUT/kt_$prec.log > $OUT/${TAG}_bench_line_under_profiler_${prec}.json
done
cd $ROOT
SQ1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"
SQ2="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS"
for prec in f32 f16x3 bf16x6; do
  PRECISION=$prec VIEWS=8 scripts/pmc.sh ${TAG}_$prec "FETCH_SIZE" "WRITE_SIZE" "$SQ1" "$SQ2" "TCC_HIT_sum TCC_MISS_sum" > $OUT/pmc_$prec.log 2>&1
  cp gpurun_out/pmc_${TAG}_$prec/merged.json $OUT/${TAG}_pmc_per_kernel_${prec}_steps2_views8.json
done
python scripts/emu_layer_table.py > $OUT/${TAG}_emu_layer_table.md 2> $OUT/emu_layer_table.err
python bench.py --pmc-file $OUT/${TAG}_pmc_per_kernel_f32_steps2_views8.json > $OUT/bench.log 2>&1
tail -1 $OUT/bench.log > $OUT/${TAG}_bench_final.json
rm -rf $OUT/kt_f32 $OUT/kt_f16x3 $OUT/kt_bf16x6
ls -la $OUT
