cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_planes_gpu.py -q -m gpu -x -k "correlation" 2>&1 | tail -5
timeout 120 python tools/tmp/corr_time.py 2>&1 | tail -8
