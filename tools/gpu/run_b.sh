set -x
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
UNFLOW_WGRAD_GROUP=3 timeout 600 python -m pytest tests/test_engine_gpu.py tests/test_parity_fullsize_gpu.py tests/test_train_gpu.py -x -q 2>&1 | tail -8 > gpurun_out/r02_t10.log
Q="--steps 30 --warmup 8 --no-parity --no-alt --no-cpu-baseline --no-roofline --sustain-seconds 0"
for g in 0 1 3 6 100 0 3; do
UNFLOW_WGRAD_GROUP=$g timeout 120 python bench.py $Q 2>&1 | tail -1 | cut -c1-200 >> gpurun_out/r02_ab_wgstream_$g.log
done
