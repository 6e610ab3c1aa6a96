#!/bin/bash
# correlation backward: parity, timing, HBM counters
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r02aq; mkdir -p $O; rm -f $O/corr_hbm_counters.txt
timeout 300 python -m pytest tests/test_planes_gpu.py tests/test_ops_gpu.py -m gpu -x -q -k "correlation" 2>&1 | tail -2
timeout 60 python tools/debug/corr_bwd_time.py 2>/dev/null
for c in FETCH_SIZE; do
  timeout -s KILL 90 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/$c -o p -- python tools/debug/corr_ops_once.py > $O/$c.log 2>&1
  python tools/debug/corr_pmc_summary.py $O/$c $c | tee -a $O/corr_hbm_counters.txt
done
rm -rf $O/FETCH_SIZE $O/WRITE_SIZE
