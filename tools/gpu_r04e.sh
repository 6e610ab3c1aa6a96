cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_fullsize_gpu.py -m gpu -x -q -k "forward_warp or kat_downsample or empty" 2>&1 | tail -15 ) > gpurun_out/r04e_fw_tests.txt
tail -3 gpurun_out/r04e_fw_tests.txt
( timeout 600 python bench_ops.py > gpurun_out/r04e_bench_ops.jsonl 2> gpurun_out/r04e_bench_ops.err )
grep forward_warp gpurun_out/r04e_bench_ops.jsonl
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 ) > gpurun_out/r04e_tests.txt
tail -3 gpurun_out/r04e_tests.txt
