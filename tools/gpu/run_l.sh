# LDS-DMA filter-gradient kernel: ablations (UNFLOW_DBG 1: no loads, 8: no vmcnt wait) and SQ counters
set -x
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02l; mkdir -p $O
for d in 1 8; do
UNFLOW_DBG=$d timeout 200 python tools/per_layer_bench.py > $O/per_layer_dbg$d.txt 2>$O/err_dbg$d.txt
done
export UNFLOW_WGRAD_GROUP=0
PMC_TIMEOUT=150 bash tools/pmc_run.sh $O/pmc "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA" > $O/pmc.log 2>&1
find $O -name "*.db" -delete
