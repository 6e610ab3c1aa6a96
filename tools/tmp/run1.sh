cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_train_gpu.py -q -m gpu -x 2>&1 | tail -5
UNFLOW_OPT_CORR_RW=0 timeout 900 python -m pytest tests/test_train_gpu.py -q -m gpu -x -k two_ranks 2>&1 | tail -3
