#!/bin/bash
# correlation backward: parity, timing, HBM counters
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r02aq; mkdir -p $O; rm -f $O/corr_hbm_counters.txt
timeout 300 python -m pytest tests/test_planes_gpu.py tests/test_ops_gpu.py -m gpu -x -q -k "correlation" 2>&1 | tail -2
UNFLOW_CORR_BWD_ROT=1 timeout 300 python -m pytest tests/test_planes_gpu.py -m gpu -x -q -k "correlation_planes_bwd" 2>&1 | tail -1
timeout 60 python tools/debug/corr_bwd_time.py 2>/dev/null
UNFLOW_CORR_BWD_ROT=1 timeout 60 python tools/debug/corr_bwd_time.py 2>/dev/null
UNFLOW_CORR_BWD_ROT=0 timeout 60 python tools/debug/corr_bwd_time.py 2>/dev/null
