#!/usr/bin/env python3
"""Summarise a rocprofv3 kernel_trace.csv: mean/min duration per (kernel, grid size)."""
import collections
import csv
import glob
import sys

import os
f = sorted(glob.glob(sys.argv[1] + '/*/*kernel_trace.csv'), key=os.path.getmtime)[-1]  # newest run
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    name = r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0]
    d[(name, r['Grid_Size_X'] if 'Grid_Size_X' in r else r.get('Grid_Size', '?'))].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
for (k, g), v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    v.sort()
    print("%-28s grid=%-8s n=%6d mean=%8.2f us  median=%8.2f  min=%8.2f" % (k[:28], g, len(v), sum(v) / len(v) / 1e3, v[len(v) // 2] / 1e3, v[0] / 1e3))
