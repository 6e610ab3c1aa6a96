#!/usr/bin/env python3
"""Per-dispatch FETCH_SIZE (x2 gfx950 correction, MB) of the last training step in a rocprofv3 --pmc FETCH_SIZE pass.
usage: pmc_fetch.py <pass dir> [name filter]"""
import csv, glob, re, sys
from collections import OrderedDict


def short(n):
    n = re.sub(r'^void ', '', n); n = re.sub(r'\(anonymous namespace\)::', '', n)
    m = re.match(r'([\w:]+)(<[^(]*>)?', n); return m.group(1) + (m.group(2) or '')


d = OrderedDict()
for r in csv.DictReader(open(glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True)[0])):
    d[int(r['Dispatch_Id'])] = (short(r['Kernel_Name']), int(r['Grid_Size']) // int(r['Workgroup_Size']), float(r['Counter_Value']),
                                (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
flt = sys.argv[2] if len(sys.argv) > 2 else 'igemm_pl'
ids = sorted(d); adam = [i for i in ids if 'adam' in d[i][0]]
tot = 0
for i in ids:
    if i <= adam[-2] or i > adam[-1]: continue
    n, g, v, us = d[i]
    mb = v * 1024 * 2 / 1e6
    if flt in n:
        tot += mb
        print("%-58s %5d %8.1f us %8.1f MB %5.2f TB/s" % (n[:58], g, us, mb, mb / us / 1e6 * 1e6))
print("total %.1f MB" % tot)
