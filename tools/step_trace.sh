#!/bin/bash
# Kernel trace (start / end of every dispatch, both streams) of a few replays of the benchmarked step -> gpurun_out/<tag>_step_trace.csv
# (read by tools/step_timeline.py: per-stream busy time, gaps, overlap of the filter-gradient stream with the main one)
tag=${1:-st}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $out/${tag}_tr -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 3 --no-secondary --no-cpu-baseline --no-alt --no-parity --no-roofline --sustain-seconds 0 > /dev/null 2> $out/${tag}_tr.err )
tr=$(find $out/${tag}_tr -name "*kernel_trace.csv" | head -1)
python - "$tr" > $out/${tag}_step_trace.csv <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'adam_kernel' in r['Kernel_Name']]
sel = rows[idx[-3] + 1: idx[-1] + 1]          # the last two steps
t0 = int(sel[0]['Start_Timestamp'])
print("name,queue,start_us,end_us")
for r in sel:
    nm = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0].replace(',', ';')
    print("%s,%s,%.2f,%.2f" % (nm, r.get('Queue_Id', ''), (int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - t0) / 1e3))
PY
rm -rf $out/${tag}_tr
wc -l $out/${tag}_step_trace.csv
