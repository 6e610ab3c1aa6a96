#!/usr/bin/env python3
"""Per-dispatch memory-side view of two rocprofv3 --pmc passes of one training step (tools/pmc_run.sh):
  pass A: SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_WAVE_CYCLES
  pass B: TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum GRBM_GUI_ACTIVE
-> effective clock (GRBM_GUI_ACTIVE / 8 XCDs / duration), L2 hit rate and request rate, fabric-side read bytes, LDS issue
stalls and bank conflicts, wave-level VMEM / LDS instruction rates.   usage: pmc_mem.py <pass A dir> <pass B dir> [filter] [min us]"""
import csv
import glob
import re
import sys
from collections import OrderedDict


def load(passdir):
    f = glob.glob(passdir + '/**/*counter_collection.csv', recursive=True)[0]
    d = OrderedDict()
    for r in csv.DictReader(open(f)):
        e = d.setdefault(int(r['Dispatch_Id']), dict(name=r['Kernel_Name'], grid=int(r['Grid_Size']) // int(r['Workgroup_Size']),
                                                     t0=int(r['Start_Timestamp']), t1=int(r['End_Timestamp']), c={}))
        e['c'][r['Counter_Name']] = float(r['Counter_Value'])
    return d


def short(n):
    n = re.sub(r'^void ', '', n)
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    m = re.match(r'([\w:]+)(<[^(]*>)?', n)
    return m.group(1) + (m.group(2) or '')


def last_step(d):
    ids = sorted(d)
    adam = [i for i in ids if 'adam_kernel' in d[i]['name']]
    return [i for i in ids if adam[-2] < i <= adam[-1]]


flt = sys.argv[3] if len(sys.argv) > 3 else 'igemm_pl'
min_us = float(sys.argv[4]) if len(sys.argv) > 4 else 80.0
dA, dB = load(sys.argv[1]), load(sys.argv[2])
print("%-52s %5s %7s | %5s %6s %9s %8s" % ("kernel (pass B)", "wgs", "us", "GHz", "L2hit%", "L2req/us", "EArd MB"))
for i in last_step(dB):
    e = dB[i]
    us = (e['t1'] - e['t0']) / 1e3
    if flt not in e['name'] or us < min_us:
        continue
    c = e['c']
    hit, miss = c.get('TCC_HIT_sum', 0), c.get('TCC_MISS_sum', 0)
    print("%-52s %5d %7.1f | %5.2f %6.1f %9.0f %8.0f" % (short(e['name'])[:52], e['grid'], us, c.get('GRBM_GUI_ACTIVE', 0) / 8 / us / 1e3,
                                                       100 * hit / max(1.0, hit + miss), c.get('TCC_REQ_sum', 0) / us,
                                                       c.get('TCC_EA0_RDREQ_sum', 0) * 64 / 1e6))
print()
print("%-52s %5s %7s | %9s %8s %9s %8s %8s" % ("kernel (pass A)", "wgs", "us", "LDSstall%", "LDSact%", "conf/idx", "LDS/us", "VMEM/us"))
for i in last_step(dA):
    e = dA[i]
    us = (e['t1'] - e['t0']) / 1e3
    if flt not in e['name'] or us < min_us:
        continue
    c = e['c']
    wc = c.get('SQ_WAVE_CYCLES', 1) or 1
    print("%-52s %5d %7.1f | %9.1f %8.1f %9.3f %8.0f %8.0f" % (short(e['name'])[:52], e['grid'], us, 100 * c.get('SQ_WAIT_INST_LDS', 0) / wc,
                                                              100 * c.get('SQ_ACTIVE_INST_LDS', 0) / wc,
                                                              c.get('SQ_LDS_BANK_CONFLICT', 0) / max(1.0, c.get('SQ_LDS_IDX_ACTIVE', 1)),
                                                              c.get('SQ_INSTS_LDS', 0) / us, c.get('SQ_INSTS_VMEM_RD', 0) / us))
