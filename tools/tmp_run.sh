cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_planes_gpu.py -q -m gpu -x -k "first_layer or rgb4 or conv1" 2>&1 | tail -15
timeout 300 python tools/debug/conv1_time.py 2>&1 | tail -8
