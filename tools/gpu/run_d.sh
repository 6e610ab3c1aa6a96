export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | grep -v "^W2026\|^E2026" | tail -15 > gpurun_out/r02_t12.log
Q="--steps 30 --warmup 8 --no-parity --no-alt --no-cpu-baseline --no-roofline --sustain-seconds 0"
for g in 0 6 4 8 0 6; do
UNFLOW_WGRAD_GROUP=$g timeout 120 python bench.py $Q 2>&1 | tail -1 | cut -c1-200 >> gpurun_out/r02_ab2_wgstream_$g.log
done
