set -x
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02al; mkdir -p $O
timeout 1100 python -m pytest tests -x -q -m gpu 2>&1 | grep -v "^W2026\|^E2026" | tail -8 > $O/pytest.log
timeout 600 python bench.py > $O/bench.log 2>$O/bench.err; tail -1 $O/bench.log > $O/bench_line.json
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python bench.py --no-cpu-baseline --no-alt --no-parity --sustain-seconds 0 > $O/stats.log 2>&1
rm -f $O/stats/*kernel_trace.csv
