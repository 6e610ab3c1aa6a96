#!/usr/bin/env python3
"""One training step's kernel timeline from a rocprofv3 rocpd sqlite result (the dispatches between the last two Adam
launches): every dispatch over `min_us` with its grid. usage: rocpd_step.py results.db [min_us]"""
import re
import sqlite3
import sys


def short(n):
    n = re.sub(r'^void ', '', n)
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    m = re.match(r'([\w:]+)(<[^(]*>)?', n)
    return m.group(1) + (m.group(2) or '')


db = sqlite3.connect(sys.argv[1])
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 15.0
rows = db.execute("select name, start, end, grid_x, grid_y, grid_z, workgroup_x from kernels order by start").fetchall()
idx = [i for i, r in enumerate(rows) if 'adam_kernel' in r[0]]
a, b = idx[-2], idx[-1]
tot = 0.0
for r in rows[a + 1:b + 1]:
    d = (r[2] - r[1]) / 1e3
    tot += d
    if d >= min_us:
        print("%-62s grid %5d %3d %3d  %8.1f us" % (short(r[0])[:62], r[3] // max(r[6], 1), r[4], r[5], d))
print("kernel time %.1f us, span %.1f us" % (tot, (rows[b][2] - rows[a][2]) / 1e3))
