#!/usr/bin/env python3
"""Per basic block of a kernel in a hipcc -S listing: instruction counts by kind (MFMA, scratch, SGPR spill lanes, LDS, VMEM,
barriers) — to see at a glance whether spills or waits sit inside a K loop.  usage: isa_blocks.py file.s kernel_substring"""
import re
import sys
src, pat = sys.argv[1], sys.argv[2]
inside = False
blocks, cur = [], None
for ln in open(src):
    s = ln.strip()
    if re.match(r'^[_A-Za-z0-9$.]+:', s) and not s.startswith('.LBB') and not s.startswith(';'):
        name = s.split(':')[0]
        if name.startswith('_Z') or name.startswith('.L') is False:
            inside = pat in name
            if inside:
                cur = [name, {}]
                blocks.append(cur)
        continue
    if not inside:
        continue
    if s.startswith('.LBB'):
        cur = [s.split(':')[0], {}]
        blocks.append(cur)
        continue
    if s.startswith('.') or s.startswith(';') or not s:
        if s.startswith('.Lfunc_end'):
            inside = False
        continue
    op = s.split()[0]
    kind = ('mfma' if op.startswith('v_mfma') else 'scratch' if op.startswith('scratch_') else 'wlane' if op == 'v_writelane_b32' else
            'rlane' if op == 'v_readlane_b32' else 'barrier' if op == 's_barrier' else 'ds' if op.startswith('ds_') else
            'vmem' if op.startswith(('buffer_', 'global_', 'flat_')) else 'waitcnt' if op == 's_waitcnt' else
            'branch' if op.startswith('s_cbranch') or op == 's_branch' else 'valu' if op.startswith('v_') else 'salu' if op.startswith('s_') else 'other')
    cur[1][kind] = cur[1].get(kind, 0) + 1
for name, c in blocks:
    if c:
        print('%-14s %s' % (name[:14], ' '.join('%s=%d' % kv for kv in sorted(c.items()))))
