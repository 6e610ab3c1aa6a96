#!/bin/bash
# Single-stream rocprofv3 record of the benchmarked step (UNFLOW_WGRAD_GROUP=0: filter gradients inline on the main stream, so
# no kernel's duration is inflated by a concurrent one): kernel stats + the per-step launch counts -> tools/roofline_check.py
# reproduces roofline.frac from the tracked CSV.  Output: gpurun_out/<tag>_serial_kernel_stats.csv, <tag>_serial_roofline_check.txt
tag=${1:-ser}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out
( cd /tmp && UNFLOW_WGRAD_GROUP=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_prof -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 6 --no-secondary --no-cpu-baseline --no-alt --no-parity --sustain-seconds 0 > $out/${tag}_serial_bench_under_rocprof.json 2> $out/${tag}_rocprof.err )
find $out/${tag}_prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/${tag}_serial_kernel_stats.csv
( cd /tmp && UNFLOW_WGRAD_GROUP=0 timeout 600 rocprofv3 --kernel-trace --output-format csv -d $out/${tag}_prof1 -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-graph --no-secondary --no-cpu-baseline --no-alt --no-parity --no-roofline --sustain-seconds 0 > /dev/null 2>> $out/${tag}_rocprof.err )
tr=$(find $out/${tag}_prof1 -name "*kernel_trace.csv" | head -1)
python tools/roofline_check.py $out/${tag}_serial_kernel_stats.csv $tr > $out/${tag}_serial_roofline_check.txt 2>&1
rm -rf $out/${tag}_prof $out/${tag}_prof1
cat $out/${tag}_serial_roofline_check.txt | head -60
