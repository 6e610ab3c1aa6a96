#!/usr/bin/env python3
"""Per-kernel averages of every counter of one rocprofv3 --pmc pass (counter_collection.csv), first dispatch of each kernel
dropped (cold).  usage: pmc_kernels.py <pass dir> [name filter]"""
import csv
import glob
import re
import sys
from collections import defaultdict

f = glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True)[0]
flt = sys.argv[2] if len(sys.argv) > 2 else ''
disp = {}
for r in csv.DictReader(open(f)):
    n = re.sub(r'\(anonymous namespace\)::|^void ', '', r['Kernel_Name']).split('(')[0]
    if flt and flt not in n:
        continue
    e = disp.setdefault(int(r['Dispatch_Id']), dict(name=n, grid=int(r['Grid_Size']) // max(1, int(r['Workgroup_Size'])),
                                                    us=(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, c={}))
    e['c'][r['Counter_Name']] = float(r['Counter_Value'])
by = defaultdict(list)
for i in sorted(disp):
    by[(disp[i]['name'], disp[i]['grid'])].append(disp[i])
for (n, g), v in by.items():
    v = v[1:] if len(v) > 1 else v
    us = sum(e['us'] for e in v) / len(v)
    names = sorted(v[0]['c'])
    print("%-60s wgs %6d  %8.1f us  " % (n[:60], g, us) + "  ".join("%s %.4g" % (k, sum(e['c'][k] for e in v) / len(v)) for k in names))
