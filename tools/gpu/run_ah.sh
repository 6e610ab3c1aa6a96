# final state of the round: full GPU suite, bench line, rocprof kernel stats, per-layer table, PMC traffic, f16 + CSS lines, smoke
set -x
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02ah; mkdir -p $O
timeout 600 python bench.py > $O/bench.log 2>$O/bench.err; tail -1 $O/bench.log > $O/bench_line.json
timeout 1100 python -m pytest tests -x -q -m gpu 2>&1 | grep -v "^W2026\|^E2026" | tail -8 > $O/pytest.log
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python bench.py --no-cpu-baseline --no-alt --no-parity --sustain-seconds 0 > $O/stats.log 2>&1
rm -f $O/stats/*kernel_trace.csv
timeout 200 python tools/per_layer_bench.py > $O/per_layer.txt 2>$O/per_layer.err
timeout 300 python bench.py --dtype f16 --batch 8 --no-alt --no-cpu-baseline 2>&1 | tail -1 > $O/bench_f16_b8.json
timeout 300 python bench.py --flownet CSS --height 768 --width 1024 --batch 2 --no-alt --no-cpu-baseline --no-parity 2>&1 | tail -1 > $O/bench_css.json
UNFLOW_WGRAD_GROUP=0 PMC_TIMEOUT=120 bash tools/pmc_run.sh $O/pmc "FETCH_SIZE" "WRITE_SIZE" > $O/pmc.log 2>&1
find $O -name "*.db" -delete
