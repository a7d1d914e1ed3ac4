"""Merge rocprofv3 --pmc passes (one directory per pass under OUT/p*/) into per-kernel figures:
{kernel: {COUNTER_total, COUNTER_per_dispatch, dispatches}} -> OUT/merged.json (the format of profiles/r*_pmc_per_kernel_*.json
that bench.py reads for roofline.traffic)."""
import csv, glob, json, os, sys
from collections import defaultdict
out = sys.argv[1]
acc = defaultdict(lambda: defaultdict(float))
disp = defaultdict(lambda: defaultdict(set))
for f in glob.glob(os.path.join(out, 'p*', '**', '*counter_collection.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        k, c = r['Kernel_Name'], r['Counter_Name']
        acc[k][c] += float(r['Counter_Value'])
        disp[k][c].add(r['Dispatch_Id'])
res = {}
for k, v in acc.items():
    e = {}
    for c, tot in v.items():
        n = max(1, len(disp[k][c]))
        e[c + '_total'] = tot
        e[c + '_per_dispatch'] = tot / n
        e['dispatches'] = n
    res[k] = e
# how the passes were run (scripts/pmc.sh exports these): bench.py refuses to scale a profile to another batch size
res['_meta'] = {'views_per_step': int(os.environ.get('VIEWS', '4')), 'steps': int(os.environ.get('PMC_STEPS', '2')),
                'warmup': int(os.environ.get('PMC_WARMUP', '1')), 'precision': os.environ.get('PRECISION', 'f32'),
                'img_size': int(os.environ.get('IMG', '512')), 'conv_algo': os.environ.get('CONV_ALGO', 'winograd')}
json.dump(res, open(os.path.join(out, 'merged.json'), 'w'), indent=1, sort_keys=True)
print('kernels', len(res))
