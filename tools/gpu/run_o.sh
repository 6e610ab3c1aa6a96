export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02o; mkdir -p $O
rocprofv3 -L > $O/counters.txt 2>&1
