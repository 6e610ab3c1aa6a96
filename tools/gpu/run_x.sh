set -x
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02x; mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_parity_fullsize_gpu.py tests/test_engine_gpu.py tests/test_fullsize_gpu.py -x -q 2>&1 | grep -v "^W2026\|^E2026" | tail -30 > $O/pytest.log
Q="--steps 30 --warmup 8 --no-parity --no-alt --no-cpu-baseline --sustain-seconds 0 --no-roofline"
for i in 1 2; do
timeout 120 python bench.py $Q 2>&1 | tail -1 | cut -c60-130 >> $O/bench_b3.log
UNFLOW_CORR_MATH=fp32 timeout 120 python bench.py $Q 2>&1 | tail -1 | cut -c60-130 >> $O/bench_fp32.log
done
timeout 300 python bench_ops.py 2>/dev/null | grep correlation > $O/ops_corr.jsonl
