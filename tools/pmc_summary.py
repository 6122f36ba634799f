#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection.csv per kernel (mean per dispatch / per wave)."""
import collections
import csv
import glob
import sys

import os
f = sorted(glob.glob(sys.argv[1] + '/*/*counter_collection.csv'), key=os.path.getmtime)
rows = list(csv.DictReader(open(f[-1])))  # newest run in the directory
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(set)
for r in rows:
    k = r['Kernel_Name'].replace('(anonymous namespace)::', '')[:14] + '|g' + r['Grid_Size']
    agg[k][r['Counter_Name']] += float(r['Counter_Value'])
    cnt[k].add(r['Dispatch_Id'])
names = sorted({r['Counter_Name'] for r in rows})
print("counters:", names)
for k in sorted(agg):
    a, n = agg[k], len(cnt[k])
    line = "%-26s n=%4d " % (k, n)
    for c in names:
        line += " %s=%.0f" % (c.replace('SQ_', ''), a[c] / n)
    print(line)
