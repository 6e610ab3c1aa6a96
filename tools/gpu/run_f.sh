export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_train_gpu.py tests/test_engine_gpu.py tests/test_f16_gpu.py -x -q 2>&1 | grep -v "^W2026\|^E2026" | tail -15 > gpurun_out/r02_t13.log
Q="--steps 30 --warmup 8 --no-parity --no-alt --no-cpu-baseline --no-roofline --sustain-seconds 0"
for g in 0 1 0 1 0 1; do
UNFLOW_OVERLAP_ADAM=$g timeout 120 python bench.py $Q 2>&1 | tail -1 | cut -c1-200 >> gpurun_out/r02_ab_adam_$g.log
done
