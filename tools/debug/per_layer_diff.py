#!/usr/bin/env python3
"""Rows of the per-layer tables gpurun_out/pl_<i>_<round>.txt (tools/debug/per_layer_ab.sh) whose median over the rounds differs from the
LAST setting's by more than 2.5 us."""
import glob, os, re, sys
names = sys.argv[1:]
root = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'gpurun_out')
def load(f):
    d = []
    for ln in open(f):
        m = re.match(r'(\w+)\s+(\S+)\s+(\S+)\s+(\d+)\s+([\d.]+)\s+([\d.]+)', ln)
        if m: d.append((m.group(1), m.group(2), m.group(3), float(m.group(5))))
    return d
n = len(names)
T = {i: [load(f) for f in sorted(glob.glob(os.path.join(root, 'pl_%d_*.txt' % i)))] for i in range(1, n + 1)}
med = lambda xs: sorted(xs)[len(xs) // 2]
tot = {i: med([sum(r[3] for r in t) for t in T[i]]) for i in T}
print("totals (median of rounds, us): " + "   ".join("%s %.0f" % (names[i - 1], tot[i]) for i in sorted(T)))
for k in range(len(T[n][0])):
    base = med([t[k][3] for t in T[n]])
    out = []
    for i in range(1, n):
        v = med([t[k][3] for t in T[i]])
        if abs(v - base) > 2.5: out.append("%s %+.1f" % (names[i - 1].replace('UNFLOW_OPT_', ''), v - base))
    if out: print("%-18s %-20s %-20s base %6.1f   %s" % (T[n][0][k][0], T[n][0][k][1], T[n][0][k][2], base, '   '.join(out)))
