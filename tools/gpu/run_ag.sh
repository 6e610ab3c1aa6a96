export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02ag; mkdir -p $O
timeout 900 python tools/debug/wgstream_flake.py 2>&1 | grep -v "^W2026\|^E2026" | tail -12 > $O/flake_tiny_side.log
Q="--steps 30 --warmup 8 --no-parity --no-alt --no-cpu-baseline --sustain-seconds 0 --no-roofline"
for i in 1 2; do
timeout 120 python bench.py $Q 2>&1 | tail -1 | cut -c60-130 >> $O/bench_side.log
UNFLOW_WGRAD_INLINE_TINY=1 timeout 120 python bench.py $Q 2>&1 | tail -1 | cut -c60-130 >> $O/bench_inline.log
done
timeout 600 python -m pytest tests/test_engine_gpu.py tests/test_train_gpu.py -x -q 2>&1 | grep -v "^W2026\|^E2026" | tail -4 > $O/pytest.log
