#!/usr/bin/env python3
"""Per-kernel sums of rocprofv3 --pmc counter_collection.csv files. usage: pmc_summary.py <dir with pass*/ subdirs> [name filter]"""
import csv
import glob
import re
import sys
from collections import defaultdict


def short(n):
    n = re.sub(r'^void ', '', n)
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    m = re.match(r'([\w:]+)(<[^(]*>)?', n)
    return m.group(1) + (m.group(2) or '')


acc = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(lambda: defaultdict(int))
for f in glob.glob(sys.argv[1] + '/pass*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = short(r['Kernel_Name'])
        acc[k][r['Counter_Name']] += float(r['Counter_Value'])
        cnt[k][r['Counter_Name']] += 1
flt = sys.argv[2] if len(sys.argv) > 2 else ''
for k in sorted(acc, key=lambda k: -acc[k].get('SQ_WAVE_CYCLES', 0)):
    if flt not in k:
        continue
    print(k)
    for c in sorted(acc[k]):
        print("    %-28s %16.0f   (%d dispatches)" % (c, acc[k][c], cnt[k][c]))
