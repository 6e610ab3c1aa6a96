export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02aa; mkdir -p $O
timeout 1100 python -m pytest tests -x -q -m gpu 2>&1 | grep -v "^W2026\|^E2026" | tail -8 > $O/pytest.log
timeout 600 python bench.py > $O/bench.log 2>$O/bench.err; tail -1 $O/bench.log > $O/bench_line.json
