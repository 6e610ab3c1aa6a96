#!/usr/bin/env python3
"""VALU-roofline fraction of the per-pixel loss kernels from one rocprofv3 --pmc pass (counter_collection.csv of
`rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F32 GRBM_GUI_ACTIVE --
python bench_ops.py <filter>`), per kernel name and grid, first dispatch dropped:

  GHz       = GRBM_GUI_ACTIVE / 8 XCDs / duration
  valu/px   = SQ_INSTS_VALU x 64 lanes / pixels,   trans/px = SQ_INSTS_VALU_TRANS_F32 x 64 / pixels   (lane-instructions per pixel)
  pipe occ  = (SQ_INSTS_VALU x 2 + SQ_INSTS_VALU_TRANS_F32 x 6) cycles / (1024 SIMDs x duration x clock):
              the fraction of the chip's VALU issue cycles the kernel's own instructions need — a wave64 VALU instruction
              occupies its SIMD for 2 cycles (packed dual-issue of the two 32-lane halves), a transcendental for 8 (quarter rate) —
              i.e. the kernel's position against the VALU roofline (the same accounting as profiles/r02_ops_valu_counters.txt)

usage: valu_roofline.py <pass dir> <pixels per launch> [name filter]"""
import csv
import glob
import re
import sys
from collections import defaultdict

f = glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True)[0]
npix = float(sys.argv[2])
flt = sys.argv[3] if len(sys.argv) > 3 else ''
disp = {}
for r in csv.DictReader(open(f)):
    n = re.sub(r'\(anonymous namespace\)::|^void ', '', r['Kernel_Name']).split('(')[0]
    if flt and flt not in n:
        continue
    e = disp.setdefault(int(r['Dispatch_Id']), dict(name=n, us=(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, c={}))
    e['c'][r['Counter_Name']] = float(r['Counter_Value'])
by = defaultdict(list)
for i in sorted(disp):        # one row per (kernel, instruction count): a kernel launched with different template / run-time parameters
    e = disp[i]               # (the census at max_distance 1 and 3) executes different instruction counts per pixel
    by["%s [%d valu/px]" % (e['name'], round(e['c'].get('SQ_INSTS_VALU', 0.0) * 64 / npix))].append(e)
print("%-56s %4s %9s %5s %8s %9s %9s" % ("kernel", "n", "us", "GHz", "valu/px", "trans/px", "pipe occ"))
for n, v in by.items():
    v = v[1:] if len(v) > 1 else v
    avg = lambda k: sum(e['c'].get(k, 0.0) for e in v) / len(v)      # noqa: E731
    us = sum(e['us'] for e in v) / len(v)
    ghz = avg('GRBM_GUI_ACTIVE') / 8 / (us * 1e3)
    valu, trans = avg('SQ_INSTS_VALU'), avg('SQ_INSTS_VALU_TRANS_F32')
    occ = (valu * 2 + trans * 6) / (1024 * us * 1e3 * ghz) if ghz > 0 else float('nan')
    print("%-56s %4d %9.1f %5.2f %8.1f %9.1f %8.0f%%" % (n[:56], len(v), us, ghz, valu * 64 / npix, trans * 64 / npix, occ * 100))
