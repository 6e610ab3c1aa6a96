#!/usr/bin/env python3
"""Scan gfx950 assembly for the VMEM store-data hazard LLVM does not cover.

The rule the hardware documents (and GCNHazardRecognizer::createsVALUHazard implements): a VMEM store of MORE than 64 bits of
data followed by a VALU write of the VGPRs holding that data needs 1 wait state.  LLVM applies it only when the store has NO
SGPR in its soffset field ("with an SGPR offset the hardware takes an extra cycle").  On MI355X that exemption does not hold:
`buffer_store_dwordx4 v[96:99], v118, s[28:31], s43 offen` followed at once by `v_mov_b32 v96, 0x40000000` stored 0x40000000 as
dword 0 of lanes 12-15 / 28-31 / 44-47 / 60-63 (the last quarter of each 16-lane row: the store reads its data registers four
lanes per row per cycle) in a few hundred to a few thousand stores per launch, timing-dependent (conv_first.hip, round 4:
"unexplained race"; root cause found in round 5).

usage: isa_store_hazard.py file.s [...]         (exit status 1 if any site is found)
       isa_store_hazard.py --build              (compile every csrc/*.hip to assembly with the library's flags and scan)
A site = a buffer store of > 64 data bits with an SGPR soffset whose data VGPRs are written by one of the next WINDOW
instructions (s_nop N counts as N + 1).
Limitation: the scan follows the LISTING (straight-line code after the store, through labels); a store that is the last
instruction before a taken branch is checked against the fall-through instructions, not against the branch target's."""
import glob
import os
import re
import subprocess
import sys
import tempfile

WINDOW = 2          # wait states required between the store and a write of its data registers (documented: 1; margin: 2)
STORE = re.compile(r'^\s*buffer_store_(dwordx3|dwordx4|format_xyzw?|format_d16_xyzw)\s+v\[(\d+):(\d+)\]\s*,\s*([^,]+),\s*s\[\d+:\d+\]\s*,\s*(\S+)')
VREG = re.compile(r'\bv\[(\d+):(\d+)\]|\bv(\d+)\b')


def dest_regs(ins):
    """VGPRs an instruction writes: the first operand of v_*, ds_read*, buffer/global/scratch loads (good enough for a scan)."""
    op = ins.split()[0]
    writes = op.startswith('v_') or op.startswith(('ds_read', 'ds_bpermute', 'ds_permute', 'ds_swizzle', 'buffer_load', 'global_load',
                                                      'scratch_load', 'flat_load', 'v_accvgpr'))
    if not writes or op.startswith(('v_cmp', 'v_cmpx')) and '_e64' not in op and 'vcc' in ins.split(None, 1)[1].split(',')[0]:
        return set()
    rest = ins.split(None, 1)[1] if ' ' in ins.strip() or '\t' in ins else ''
    first = rest.split(',')[0]
    out = set()
    for m in VREG.finditer(first):
        if m.group(3) is not None:
            out.add(int(m.group(3)))
        else:
            out |= set(range(int(m.group(1)), int(m.group(2)) + 1))
    if op.startswith('v_mfma') or op.startswith('v_smfma'):
        return out
    return out


def scan(path):
    lines = open(path, errors='ignore').read().split('\n')
    kernel, sites = None, []
    for i, ln in enumerate(lines):
        s = ln.strip()
        m = re.match(r'^(_Z\w+|[A-Za-z_]\w*):\s*(;.*)?$', s)
        if m and not s.startswith('.L'):
            kernel = m.group(1)
        m = STORE.match(ln)
        if not m or not re.match(r'^s\d+$', m.group(5)):
            continue
        data = set(range(int(m.group(2)), int(m.group(3)) + 1))
        waited, j = 0, i + 1
        while j < len(lines) and waited < WINDOW:
            t = lines[j].strip()
            j += 1
            if not t or t.startswith((';', '.', '//')) or t.endswith(':'):
                continue
            t = t.split(';')[0].strip()
            if t.startswith('s_nop'):
                waited += int(t.split()[1], 0) + 1
                continue
            hit = dest_regs(t) & data
            if hit:
                sites.append((kernel, i + 1, s, j, t, sorted(hit)))
                break
            waited += 1
    return sites


def build_all():
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.dirname(here))
    from unflow_amd import build as B
    out = []
    tmp = tempfile.mkdtemp(prefix='isa_store_hazard_')
    procs = []
    for src in B.sources():
        s = os.path.join(tmp, os.path.basename(src)[:-4] + '.s')
        procs.append((s, subprocess.Popen([B.HIPCC] + [f for f in B.flags_for(src) if f != '-fPIC'] + ['--cuda-device-only', '-S', src, '-o', s])))
    for s, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc -S failed for %s" % s)
        out.append(s)
    return out


if __name__ == '__main__':
    files = build_all() if sys.argv[1:] == ['--build'] else sys.argv[1:]
    total = 0
    for f in files:
        for kernel, ln, store, ln2, writer, regs in scan(f):
            total += 1
            print("%s:%d  [%s]\n    %s\n    line %d: %s   <- writes data register(s) %s" % (os.path.basename(f), ln, kernel, store, ln2, writer, regs))
    print("%d hazard site(s) in %d file(s)" % (total, len(files)))
    sys.exit(1 if total else 0)
