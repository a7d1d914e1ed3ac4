"""Merge rocprofv3 --pmc passes (counter_collection.csv) into per-kernel totals: {kernel: {counter: total, dispatches}}."""
import csv, glob, json, os, sys
from collections import defaultdict
out = sys.argv[1]
acc = defaultdict(lambda: defaultdict(float))
disp = defaultdict(set)
for f in glob.glob(os.path.join(out, 'p*', '**', '*counter_collection.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        acc[k][r['Counter_Name']] += float(r['Counter_Value'])
        disp[k].add(r['Dispatch_Id'])
res = {k: dict(v, dispatches=len(disp[k])) for k, v in acc.items()}
json.dump(res, open(os.path.join(out, 'merged.json'), 'w'), indent=1, sort_keys=True)
print('kernels', len(res))
