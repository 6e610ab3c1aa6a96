#!/bin/bash
# Fresh SQ counters of the census / second-order kernels at 16 x 768 x 1024 -> their VALU-roofline fraction (tools/valu_roofline.py)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=$GRAFT_REPO_ROOT/gpurun_out/${1:-valu}
timeout -s KILL 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F32 GRBM_GUI_ACTIVE --output-format csv -d ${out}_pmc -o p -- python bench_ops.py "ternary" > ${out}_rows.jsonl 2> ${out}_pmc.err
python tools/valu_roofline.py ${out}_pmc 12582912 > ${out}_valu_roofline.txt 2>&1
grep -i "ternary\|second_order\|kernel " ${out}_valu_roofline.txt
rm -rf ${out}_pmc
