#!/bin/bash
# effective shader clock per conv kernel = GRBM_GUI_ACTIVE / 8 XCDs / kernel wall time
for v in "$@"; do
  if [ "$v" != "base" ]; then export RNR_HIP_LIB=$GRAFT_REPO_ROOT/build_abl/librnr_$v.so; else unset RNR_HIP_LIB; fi
  OUT=$GRAFT_REPO_ROOT/gpurun_out/clock_$v; mkdir -p $OUT
  (cd /tmp; export TMPDIR=/tmp; rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --output-format csv -d $OUT -- python $GRAFT_REPO_ROOT/bench.py --views-per-step 8 --steps 3 --warmup 1 --no-cpu-baseline --no-parity > $OUT/log.txt 2>&1)
  python - $OUT $v <<'PY'
import csv,glob,sys,collections
out,v=sys.argv[1:3]
cc=glob.glob(out+'/**/*counter_collection.csv',recursive=True)[0]
kt=glob.glob(out+'/**/*kernel_trace.csv',recursive=True)[0]
dur={r['Dispatch_Id']:(int(r['End_Timestamp'])-int(r['Start_Timestamp']),r['Kernel_Name']) for r in csv.DictReader(open(kt))}
acc=collections.defaultdict(lambda:[0,0])
for r in csv.DictReader(open(cc)):
    if r['Counter_Name']!='GRBM_GUI_ACTIVE' or 'conv_halo' not in r['Kernel_Name']: continue
    d=dur[r['Dispatch_Id']][0]
    if d<200000: continue
    k=r['Kernel_Name'][10:45]
    acc[k][0]+=float(r['Counter_Value'])/8; acc[k][1]+=d
for k,(c,d) in acc.items(): print(v,k,'%.3f GHz'%(c/d), 'total ms %.2f'%(d/1e6))
PY
done
