"""Per-kernel timeline of the LAST frame in a rocprofv3 kernel trace of the one-view-per-call loop (scripts/v1_trace.sh).
Usage: python scripts/trace_frame.py gpurun_out/v1_TAG/kernel_trace_views1.csv"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'frame_prepare' in r['Kernel_Name'] or 'project_vertices' in r['Kernel_Name']]     # first launch of a frame
s, e = idx[-2], idx[-1]
t0 = int(rows[s]['Start_Timestamp'])
tot = {}
for r in rows[s:e]:
    st, en = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    name = r['Kernel_Name'].replace('rnr::', '').replace('void ', '')
    name = name.split('(')[0]
    print('%8.1f %7.1f  %-60s grid %s' % ((st - t0) / 1e3, (en - st) / 1e3, name[:60], r['Grid_Size_X']))
    key = name.split('<')[0]
    tot[key] = tot.get(key, 0) + (en - st) / 1e3
print('frame span %.1f us' % ((int(rows[e]['Start_Timestamp']) - t0) / 1e3))
for k, v in sorted(tot.items(), key=lambda kv: -kv[1]):
    print('  %-40s %8.1f us' % (k, v))
