export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02ae; mkdir -p $O
timeout 900 python -m pytest tests/test_planes_gpu.py tests/test_conv_gpu.py tests/test_engine_gpu.py -x -q 2>&1 | grep -v "^W2026\|^E2026" | tail -30 > $O/pytest.log
timeout 200 python tools/per_layer_bench.py > $O/per_layer_tf.txt 2>$O/err1.txt
UNFLOW_WGRAD_TF=0 timeout 200 python tools/per_layer_bench.py > $O/per_layer_notf.txt 2>$O/err2.txt
Q="--steps 30 --warmup 8 --no-parity --no-alt --no-cpu-baseline --sustain-seconds 0 --no-roofline"
for i in 1 2; do
timeout 120 python bench.py $Q 2>&1 | tail -1 | cut -c60-130 >> $O/bench_tf.log
UNFLOW_WGRAD_TF=0 timeout 120 python bench.py $Q 2>&1 | tail -1 | cut -c60-130 >> $O/bench_notf.log
done
