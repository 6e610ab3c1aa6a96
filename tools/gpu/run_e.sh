export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
Q="--steps 30 --warmup 8 --no-parity --no-alt --no-cpu-baseline --no-roofline --sustain-seconds 0"
for g in 0 2 3 4 5 7 10 0 2 3 4 5 7 10; do
UNFLOW_WGRAD_GROUP=$g timeout 120 python bench.py $Q 2>&1 | tail -1 | cut -c1-200 >> gpurun_out/r02_ab3_wgstream_$g.log
done
