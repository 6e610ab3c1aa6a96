# round-2 baseline measurement at HEAD: bench line, rocprof kernel stats, one-step eager trace, PMC traffic + SQ counters, full GPU suite
set -x
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02g; mkdir -p $O
timeout 600 python bench.py > $O/bench.log 2>$O/bench.err; tail -1 $O/bench.log > $O/bench_line.json
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python bench.py --no-cpu-baseline --no-alt --no-parity --sustain-seconds 0 > $O/stats.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o t -- python bench.py --no-graph --steps 3 --warmup 2 --no-cpu-baseline --no-alt --no-parity --no-roofline --sustain-seconds 0 > $O/trace.log 2>&1
PMC_TIMEOUT=150 bash tools/pmc_run.sh $O/pmc "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU" > $O/pmc.log 2>&1
find $O -name "*.db" -delete
timeout 1100 python -m pytest tests -x -q -m gpu 2>&1 | grep -v "^W2026\|^E2026" | tail -15 > $O/pytest.log
ls -R $O | head -50
