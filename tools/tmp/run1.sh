cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_parity_fullsize_gpu.py tests/test_ops_gpu.py tests/test_planes_gpu.py -q -m gpu -x -k "correlation" 2>&1 | tail -5
for i in 1 2; do
timeout 120 python tools/debug/corr_bwd_time.py 2>&1 | tail -1
UNFLOW_LIB_PATH=$GRAFT_REPO_ROOT/tools/tmp/trace/libunflow_old.so timeout 120 python tools/debug/corr_bwd_time.py 2>&1 | tail -1
done
