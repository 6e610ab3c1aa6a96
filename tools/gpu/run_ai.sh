export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02ai; mkdir -p $O
timeout -s KILL 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F32 SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/pmc -o p -- python bench_ops.py > $O/ops_pmc.log 2>&1
find $O -name "*.db" -delete
ls $O/pmc
