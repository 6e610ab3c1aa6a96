set -x
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02q; mkdir -p $O
timeout 600 python -m pytest tests/test_planes_gpu.py tests/test_conv_gpu.py tests/test_engine_gpu.py -x -q 2>&1 | grep -v "^W2026\|^E2026" | tail -30 > $O/pytest.log
timeout 200 python tools/per_layer_bench.py > $O/per_layer_2d.txt 2>$O/err1.txt
UNFLOW_GATHER_TILE2D=0 timeout 200 python tools/per_layer_bench.py > $O/per_layer_1d.txt 2>$O/err2.txt
Q="--steps 30 --warmup 8 --no-parity --no-alt --no-cpu-baseline --sustain-seconds 0"
for i in 1 2; do
timeout 120 python bench.py $Q 2>&1 | tail -1 >> $O/bench_2d.log
UNFLOW_GATHER_TILE2D=0 timeout 120 python bench.py $Q 2>&1 | tail -1 >> $O/bench_1d.log
done
