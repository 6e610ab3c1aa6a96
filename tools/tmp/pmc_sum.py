import csv, glob, sys
from collections import defaultdict
for d in sorted(glob.glob(sys.argv[1] + '/pass*')):
    fs = glob.glob(d + '/**/*counter_collection.csv', recursive=True)
    if not fs: print(d, 'no csv'); continue
    acc = defaultdict(list); dur = []
    for r in csv.DictReader(open(fs[0])):
        if 'corr_fwd_rw' not in r['Kernel_Name']: continue
        acc[r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in acc.items():
        print('%-32s %14.0f  (last of %d dispatches)' % (k, v[-1], len(v)))
