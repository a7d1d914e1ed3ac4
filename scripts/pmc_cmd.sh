#!/bin/bash
# usage: scripts/pmc_cmd.sh TAG "COMMAND ..." "COUNTERS A B" ["COUNTERS ..."]...
# One rocprofv3 --kernel-trace --pmc pass per counter group over an arbitrary command (run from the repo root), merged per
# (each pass under its own timeout: a pass that aborts inside rocprofv3 otherwise sits until gpurun's limit) kernel by scripts/pmc_merge.py into gpurun_out/pmc_TAG/merged.json.  Never combined with other trace domains.
TAG=$1; CMD=$2; shift; shift
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
i=0
for grp in "$@"; do
  (cd $ROOT && timeout ${PMC_TIMEOUT:-180} rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/p$i -- $CMD > $OUT/p$i.log 2>&1)
  i=$((i+1))
done
python $ROOT/scripts/pmc_merge.py $OUT
python - <<PY
import json
m = json.load(open('$OUT/merged.json'))
for k, v in sorted(m.items()):
    if k == '_meta' or 'conv' not in k: continue
    print(k[:90])
    print('   ', {c[:-13]: round(x) for c, x in v.items() if c.endswith('_per_dispatch')})
PY
