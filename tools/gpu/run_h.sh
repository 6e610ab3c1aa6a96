# per-layer table (serial launches) with the UNFLOW_DBG ablations, PMC passes (traffic + SQ counters)
set -x
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02h; mkdir -p $O
for d in 0 1 2 3 4; do
UNFLOW_DBG=$d timeout 200 python tools/per_layer_bench.py > $O/per_layer_dbg$d.txt 2>$O/per_layer_dbg$d.err
done
export UNFLOW_WGRAD_GROUP=0
PMC_TIMEOUT=150 bash tools/pmc_run.sh $O/pmc "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU" > $O/pmc.log 2>&1
find $O -name "*.db" -delete
find $O -name "*kernel_trace.csv" -size +20M -delete
ls -R $O | head -50
