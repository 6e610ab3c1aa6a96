export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02am; mkdir -p $O
timeout 900 python -m pytest tests/test_planes_gpu.py tests/test_ops_gpu.py tests/test_engine_gpu.py tests/test_parity_fullsize_gpu.py -x -q -k "correlation or step or stack or css or flownetc" 2>&1 | grep -v "^W2026\|^E2026" | tail -12 > $O/pytest.log
Q="--steps 30 --warmup 8 --no-alt --no-cpu-baseline --sustain-seconds 0 --no-roofline"
for i in 1 2; do
timeout 120 python bench.py $Q 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['parity']['rel'], d['parity']['final_flow_epe_fw_px'])" >> $O/bench_pl.log
UNFLOW_CORR_BWD_PLANES=0 timeout 120 python bench.py $Q 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'])" >> $O/bench_b3.log
done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python bench.py --no-cpu-baseline --no-alt --no-parity --sustain-seconds 0 --no-roofline --steps 10 > $O/stats.log 2>&1
grep -i "corr" $O/stats/s_kernel_stats.csv | cut -c1-160 > $O/corr_stats.txt
