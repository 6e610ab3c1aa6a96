#!/usr/bin/env python3
"""For every kernel in a gfx950 .s file: walk the instruction stream and classify each MFMA by what precedes it:
  same-acc clean      : previous instruction is an MFMA with the same destination (accumulator forwarded)
  same-acc interrupted: previous MFMA has the same destination but other instructions lie between (the +43-cycle cliff)
  other-acc           : previous MFMA wrote a different accumulator
usage: mfma_chain.py file.s [kernel filter]"""
import re, sys
txt = open(sys.argv[1]).read().split('\n')
flt = sys.argv[2] if len(sys.argv) > 2 else ''
kern = None
stats = {}
prev_dst = None; gap = 0
for ln in txt:
    m = re.match(r'^(_Z\w+):', ln)
    if m:
        kern = m.group(1); prev_dst = None; gap = 0; stats[kern] = [0, 0, 0, {}]
        continue
    if kern is None: continue
    s = ln.strip()
    if not s or s.startswith(';') or s.startswith('.') or s.endswith(':'): 
        if s.startswith('.LBB') or (s.endswith(':') and not s.startswith(';')): prev_dst = None
        continue
    op = s.split()[0]
    if op.startswith('v_mfma'):
        dst = s.split()[1].rstrip(',')
        if prev_dst == dst:
            if gap == 0: stats[kern][0] += 1
            else:
                stats[kern][1] += 1
                stats[kern][3][gap] = stats[kern][3].get(gap, 0) + 1
        else: stats[kern][2] += 1
        prev_dst = dst; gap = 0
    else:
        gap += 1
for k, (c, i, o, h) in stats.items():
    if c + i + o == 0 or flt not in k: continue
    print("%-90s mfma %5d  same-acc clean %5d  interrupted %5d  other-acc %5d  gaps %s" % (k[:90], c + i + o, c, i, o, dict(sorted(h.items()))))
