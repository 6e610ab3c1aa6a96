# work orders 0/1/2 forced for every gather/halo launch: per-layer tables; default policy: tests + FETCH pass
set -x
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02j; mkdir -p $O
timeout 200 python tools/per_layer_bench.py > $O/per_layer_default.txt 2>$O/err_default.txt
for o in 0 1 2; do
UNFLOW_XCD_ORDER=$o UNFLOW_XCD_ORDER_HALO=$o timeout 200 python tools/per_layer_bench.py > $O/per_layer_o$o.txt 2>$O/err_o$o.txt
done
timeout 600 python -m pytest tests/test_planes_gpu.py tests/test_conv_gpu.py -x -q 2>&1 | grep -v "^W2026\|^E2026" | tail -4 > $O/pytest.log
UNFLOW_XCD_ORDER=2 UNFLOW_XCD_ORDER_HALO=1 timeout 600 python -m pytest tests/test_planes_gpu.py tests/test_conv_gpu.py -x -q 2>&1 | grep -v "^W2026\|^E2026" | tail -4 >> $O/pytest.log
export UNFLOW_WGRAD_GROUP=0
PMC_TIMEOUT=150 bash tools/pmc_run.sh $O/pmc "FETCH_SIZE" > $O/pmc.log 2>&1
find $O -name "*.db" -delete
