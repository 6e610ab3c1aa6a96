set -x
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02n; mkdir -p $O
UNFLOW_GATHER_NSPLIT_MINM=60000 timeout 200 python tools/per_layer_bench.py > $O/per_layer_nsplit.txt 2>$O/err1.txt
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python bench.py --no-cpu-baseline --no-alt --no-parity --sustain-seconds 0 --no-roofline > $O/stats.log 2>&1
rm -f $O/stats/*kernel_trace.csv
