#!/bin/bash
# usage: tools/resusage.sh file.hip [extra hipcc flags]  ->  per kernel: VGPRs, AGPRs, SGPRs, scratch, SGPR spills, occupancy
f=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -fno-vectorize -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -Rpass-analysis=kernel-resource-usage "$@" -c $f -o /tmp/resusage.o 2>&1 | python3 -c "
import sys, re, subprocess
cur, d = None, {}
for l in sys.stdin:
    m = re.search(r'Function Name: (\S+)', l)
    if m:
        cur, d = m.group(1), {}
        continue
    m = re.search(r'remark:\s+([^:]+): (\S+)', l)
    if m and cur:
        d[m.group(1).strip()] = m.group(2)
        if m.group(1).startswith('LDS Size'):
            name = subprocess.run(['c++filt', cur], capture_output=True, text=True).stdout.strip()
            name = re.sub(r'\(anonymous namespace\)::', '', name)[:80]
            print('%-82s V%-4s A%-4s S%-4s scratch %-4s sspill %-4s occ %s' % (name, d.get('VGPRs'), d.get('AGPRs'), d.get('TotalSGPRs'),
                  d.get('ScratchSize [bytes/lane]'), d.get('SGPRs Spill'), d.get('Occupancy [waves/SIMD]')))
    elif 'error' in l:
        print(l.rstrip())
"
