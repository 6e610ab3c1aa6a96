#!/usr/bin/env python3
"""Scan gfx950 assembly for VGPRs that are touched while a VMEM load into them is still in flight.

VGPR reads have NO interlock with VMEM returns: the only protection is an `s_waitcnt vmcnt(n)` between the load and the first
use of its destination.  The compiler keeps that book for the loads it emits itself; for loads written as INLINE ASM (the
LDS-ring kernels of csrc/correlation_planes.hip issue their fragment loads by hand so that the waits are immediates) it knows
nothing — a register-allocator copy of such a destination (a loop-carried value: `v_mov_b64 v[108:109], v[132:133]` at the end
of the loop body) before the covering wait reads whatever the register held BEFORE the load landed, depending on memory
timing (corr_fwd_ring_kernel: the f0 fragments of the next chunk).

Model: vmcnt counts VMEM instructions (loads, LDS-DMA loads and stores) in issue order and they retire in order;
`s_waitcnt vmcnt(n)` leaves at most the n youngest outstanding.  A site = any instruction that names (reads OR writes) a VGPR
that is the destination of a still-outstanding load.  The scan is linear per function and forgets its state after an
unconditional branch / s_endpgm / s_setpc (the next block's predecessors are elsewhere); a hazard that exists only along a
taken branch whose target precedes the load in the listing is NOT seen (a loop-carried copy at the END of the body is).

usage: isa_load_hazard.py file.s [...]          (exit status 1 if any site is found)
       isa_load_hazard.py --build               (compile every csrc/*.hip to assembly with the library's flags and scan)"""
import os
import re
import sys

VREG = re.compile(r'\bv\[(\d+):(\d+)\]|\bv(\d+)\b')
VMEM = re.compile(r'^(buffer|global|flat|scratch|tbuffer)_(load|store|atomic)')
VMCNT = re.compile(r'vmcnt\((\d+)\)')


def regs_of(text):
    out = set()
    for m in VREG.finditer(text):
        if m.group(3) is not None:
            out.add(int(m.group(3)))
        else:
            out |= set(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


def scan(path):
    kernel, sites = None, []
    pending = []          # outstanding VMEM instructions in issue order: (destination VGPR set, line number, text)
    for i, ln in enumerate(open(path, errors='ignore').read().split('\n')):
        s = ln.strip()
        m = re.match(r'^(_Z\w+|[A-Za-z_]\w*):\s*(;.*)?$', s)
        if m and not s.startswith('.L'):
            kernel, pending = m.group(1), []
            continue
        if not s or s.startswith((';', '.', '//')) or s.endswith(':'):
            continue
        t = s.split(';')[0].strip()
        if not t:
            continue
        op = t.split()[0]
        if op == 's_waitcnt':
            m = VMCNT.search(t)
            if m:
                n = int(m.group(1))
                pending = pending[len(pending) - n:] if n < len(pending) else pending
            elif re.match(r'^s_waitcnt\s+(0x[0-9a-fA-F]+|\d+)\s*$', t):      # raw immediate: vmcnt = bits [3:0] | [15:14] << 4
                v = int(t.split()[1], 0)
                n = (v & 15) | ((v >> 14) & 3) << 4
                pending = pending[len(pending) - n:] if n < len(pending) else pending
            continue
        if op in ('s_branch', 's_endpgm', 's_setpc_b64', 's_swappc_b64'):
            pending = []
            continue
        busy = set()
        for d, _, _ in pending:
            busy |= d
        used = t
        if VMEM.match(op) and '_load' in op and ',' in t:       # a load over an outstanding load's destination retires after it
            used = t.split(',', 1)[1]
        hit = regs_of(used) & busy
        if hit:
            src = [(l, x) for d, l, x in pending if d & hit]
            sites.append((kernel, i + 1, t, sorted(hit), src[0]))
        if VMEM.match(op):
            dest = set()
            if '_load' in op and ' lds' not in t and not t.endswith(' lds'):
                dest = regs_of(t.split(None, 1)[1].split(',')[0])
            elif '_atomic' in op and ' glc' in t:
                dest = regs_of(t.split(None, 1)[1].split(',')[0])
            pending.append((dest, i + 1, t))
    return sites


def build_all():
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here)
    import isa_store_hazard
    return isa_store_hazard.build_all()


def report(files):
    total = 0
    for f in files:
        for kernel, ln, ins, regs, (lln, load) in scan(f):
            total += 1
            print("%s:%d  [%s]\n    %s   <- touches %s while the load of line %d is outstanding:\n    %s"
                  % (os.path.basename(f), ln, kernel, ins, regs, lln, load))
    print("%d in-flight-load hazard site(s) in %d file(s)" % (total, len(files)))
    return total


if __name__ == '__main__':
    files = build_all() if sys.argv[1:] == ['--build'] else sys.argv[1:]
    sys.exit(1 if report(files) else 0)
