"""Aggregate rocprofv3 --pmc CSV output (counter_collection.csv of each pass) per kernel name:
mean counter value per dispatch.  Usage: python tools/pmc_summary.py gpurun_out/<tag>"""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
agg = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(root, '**', '*counter_collection.csv'), recursive=True):
    per_dispatch = defaultdict(float)
    names = {}
    with open(f) as fh:
        for r in csv.DictReader(fh):
            key = (r.get('Dispatch_Id'), r['Counter_Name'])
            per_dispatch[key] += float(r['Counter_Value'])
            names[r.get('Dispatch_Id')] = r['Kernel_Name']
    for (d, c), v in per_dispatch.items():
        agg[names[d]][c].append(v)
dur = defaultdict(list)
for f in glob.glob(os.path.join(root, '**', '*kernel_trace.csv'), recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            dur[r['Kernel_Name']].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
for k in sorted(agg):
    d = dur.get(k, [0])
    print('%s\n   dispatches(all passes)=%d  avg_duration_us(profiled)=%.1f' % (k[:110], len(d), sum(d) / max(1, len(d)) / 1e3))
    for c in sorted(agg[k]):
        v = agg[k][c]
        print('   %-32s mean %.6g  (n=%d)' % (c, sum(v) / len(v), len(v)))
