# XCD-aware work order: correctness, per-layer table, quick bench A/B (UNFLOW_XCD_SWIZZLE=0 = old linear order), FETCH pass
set -x
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02i; mkdir -p $O
timeout 600 python -m pytest tests/test_planes_gpu.py tests/test_conv_gpu.py tests/test_engine_gpu.py -x -q 2>&1 | grep -v "^W2026\|^E2026" | tail -8 > $O/pytest.log
timeout 200 python tools/per_layer_bench.py > $O/per_layer.txt 2>$O/per_layer.err
UNFLOW_XCD_SWIZZLE=0 timeout 200 python tools/per_layer_bench.py > $O/per_layer_noxcd.txt 2>$O/per_layer_noxcd.err
Q="--steps 30 --warmup 8 --no-parity --no-alt --no-cpu-baseline --sustain-seconds 0"
for i in 1 2; do
timeout 120 python bench.py $Q 2>&1 | tail -1 >> $O/bench_xcd.log
UNFLOW_XCD_SWIZZLE=0 timeout 120 python bench.py $Q 2>&1 | tail -1 >> $O/bench_noxcd.log
done
export UNFLOW_WGRAD_GROUP=0
PMC_TIMEOUT=150 bash tools/pmc_run.sh $O/pmc "FETCH_SIZE" > $O/pmc.log 2>&1
find $O -name "*.db" -delete
