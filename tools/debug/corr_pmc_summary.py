"""Per-kernel averages of a rocprofv3 --pmc pass (counter_collection.csv): usage corr_pmc_summary.py <pass dir> <FETCH|WRITE>.
FETCH_SIZE / WRITE_SIZE come in units of 1024 B; FETCH_SIZE x2 on gfx950 (MI355X_MICROARCH.md, tools/pmc_traffic.py)."""
import csv, glob, re, sys
from collections import defaultdict
acc = defaultdict(list)
for r in csv.DictReader(open(glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True)[0])):
    n = re.sub(r'\(anonymous namespace\)::|^void ', '', r['Kernel_Name']).split('(')[0]
    acc[(n, int(r['Grid_Size']))].append((float(r['Counter_Value']), (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3))
mul = 2048.0 if sys.argv[2].startswith('F') else 1024.0
for (n, g), v in sorted(acc.items()):
    if 'corr' not in n: continue
    v = v[1:] if len(v) > 1 else v      # drop the cold launch
    mb = sum(x[0] for x in v) / len(v) * mul / 1e6
    us = sum(x[1] for x in v) / len(v)
    print("%-28s grid %8d  %s %8.1f MB/launch  %7.1f us (under the profiler)" % (n, g, sys.argv[2], mb, us))
