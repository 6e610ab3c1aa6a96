cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
bash tools/ab_bench.sh - "UNFLOW_WGRAD_GROUP=0" "UNFLOW_WGRAD_GROUP=3" "UNFLOW_WGRAD_GROUP=10" - "UNFLOW_OPT_STREAMK_GROUPS=16" "UNFLOW_OPT_STREAMK_GROUPS=4" "UNFLOW_OPT_STREAMK=0" > gpurun_out/r04m_ab.txt 2>&1
cat gpurun_out/r04m_ab.txt
