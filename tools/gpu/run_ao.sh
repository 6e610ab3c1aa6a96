#!/bin/bash
# correlation kernels: parity tests + timings
mkdir -p gpurun_out
timeout 400 python -m pytest tests/test_planes_gpu.py tests/test_ops_gpu.py -m gpu -x -q -k "correlation" > gpurun_out/ao_tests.txt 2>&1
tail -5 gpurun_out/ao_tests.txt
timeout 60 python tools/debug/corr_fwd_time.py 2>/dev/null
UNFLOW_CORR_WB=0 timeout 60 python tools/debug/corr_fwd_time.py 2>/dev/null
