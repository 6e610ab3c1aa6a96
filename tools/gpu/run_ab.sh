export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02ab; mkdir -p $O
timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_fullsize_gpu.py -x -q 2>&1 | grep -v "^W2026\|^E2026" | tail -4 > $O/pytest.log
timeout 300 python bench_ops.py 2>/dev/null > $O/ops.jsonl
