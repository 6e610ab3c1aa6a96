#!/usr/bin/env python3
"""Per-dispatch view of one rocprofv3 --pmc pass (SQ counters): for every dispatch of one training step whose kernel name
contains the filter: grid (workgroups), duration, matrix-pipe utilisation (SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x
duration x clock)), and the split of the resident wave cycles into parked (s_waitcnt / barrier), issue-stalled and issuing.
usage: pmc_dispatch.py <pass dir> [name filter] [clock GHz, default 2.1]"""
import csv
import glob
import re
import sys
from collections import OrderedDict


def short(n):
    n = re.sub(r'^void ', '', n)
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    m = re.match(r'([\w:]+)(<[^(]*>)?', n)
    return m.group(1) + (m.group(2) or '')


f = glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True)[0]
flt = sys.argv[2] if len(sys.argv) > 2 else 'igemm'
ghz = float(sys.argv[3]) if len(sys.argv) > 3 else 2.1
d = OrderedDict()
for r in csv.DictReader(open(f)):
    k = int(r['Dispatch_Id'])
    e = d.setdefault(k, dict(name=short(r['Kernel_Name']), grid=int(r['Grid_Size']) // int(r['Workgroup_Size']),
                             t0=int(r['Start_Timestamp']), t1=int(r['End_Timestamp']), c={}))
    e['c'][r['Counter_Name']] = float(r['Counter_Value'])
ids = sorted(d)
adam = [i for i in ids if 'adam_kernel' in d[i]['name']]
lo, hi = (adam[-2], adam[-1]) if len(adam) >= 2 else (ids[0], ids[-1])
print("%-52s %6s %8s %6s %6s %6s %6s %7s" % ("kernel", "wgs", "us", "mfma%", "park%", "stall%", "issue%", "valu/mf"))
for i in ids:
    if i <= lo or i > hi:
        continue
    e = d[i]
    if flt not in e['name']:
        continue
    c = e['c']
    us = (e['t1'] - e['t0']) / 1e3
    wc = c.get('SQ_WAVE_CYCLES', 0) or 1
    mf = c.get('SQ_INSTS_MFMA', 0) or 1
    util = c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (1024 * us * 1e3 * ghz)
    print("%-52s %6d %8.1f %6.1f %6.1f %6.1f %6.1f %7.2f" % (
        e['name'][:52], e['grid'], us, 100 * util, 100 * c.get('SQ_WAIT_ANY', 0) / wc, 100 * c.get('SQ_WAIT_INST_ANY', 0) / wc,
        100 * c.get('SQ_ACTIVE_INST_ANY', 0) / wc, (c.get('SQ_INSTS_VALU', 0) - mf) / mf))
