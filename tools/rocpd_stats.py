#!/usr/bin/env python3
"""Per-kernel statistics from a rocprofv3 (ROCm 7.2) rocpd sqlite result: name, calls, total / average duration.
usage: rocpd_stats.py results.db [steps]   (steps: divide the calls / totals by the number of profiled steps)"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r'^void ', '', name)
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    m = re.match(r'([\w:]+)(<[^(]*>)?', name)
    return (m.group(1) + (m.group(2) or '')) if m else name


def main():
    db = sqlite3.connect(sys.argv[1])
    steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    namecol = 'name' if 'name' in cols else 'kernel_name'
    rows = db.execute("select %s, count(*), sum(end - start), min(end - start), max(end - start) from kernels group by %s "
                      "order by 3 desc" % (namecol, namecol)).fetchall()
    tot = sum(r[2] for r in rows)
    print("%-110s %8s %10s %9s %6s" % ("kernel", "calls/st", "us/step", "avg us", "%"))
    for n, c, t, mn, mx in rows:
        print("%-110s %8.1f %10.1f %9.2f %6.2f" % (short(n)[:110], c / steps, t / steps / 1e3, t / c / 1e3, 100.0 * t / tot))
    print("total kernel time per step: %.1f us" % (tot / steps / 1e3))


if __name__ == '__main__':
    main()
