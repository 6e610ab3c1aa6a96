cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_fullsize_gpu.py tests/test_engine_gpu.py -m gpu -x -q 2>&1 | tail -8 ) > gpurun_out/r04i_tests.txt
tail -3 gpurun_out/r04i_tests.txt
( timeout 300 python bench_ops.py 2>/dev/null > gpurun_out/r04i_bench_ops.jsonl ); grep -i "warp" gpurun_out/r04i_bench_ops.jsonl | grep -v forward_warp | cut -c1-190
