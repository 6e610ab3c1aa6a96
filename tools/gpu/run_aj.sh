export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02aj; mkdir -p $O
Q="--steps 30 --warmup 8 --no-alt --no-cpu-baseline --sustain-seconds 0 --no-roofline"
for i in 1 2; do
timeout 120 python bench.py $Q 2>&1 | tail -1 | cut -c60-130 >> $O/bench_base.log
UNFLOW_HEAD_STREAM=1 timeout 120 python bench.py $Q 2>&1 | tail -1 > $O/tmp.json; cut -c60-130 $O/tmp.json >> $O/bench_head.log; python -c "
import json; d=json.load(open('$O/tmp.json')); print(d['parity'])" >> $O/parity_head.log
done
UNFLOW_HEAD_STREAM=1 timeout 900 python -m pytest tests/test_engine_gpu.py tests/test_parity_fullsize_gpu.py -x -q 2>&1 | grep -v "^W2026\|^E2026" | tail -4 > $O/pytest_head.log
