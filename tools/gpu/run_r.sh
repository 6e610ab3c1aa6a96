set -x
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02r; mkdir -p $O
Q="--steps 30 --warmup 8 --no-parity --no-alt --no-cpu-baseline --sustain-seconds 0 --no-roofline"
for i in 1 2; do
for d in 0 64 128 192; do
UNFLOW_DBG=$d timeout 120 python bench.py $Q 2>&1 | tail -1 | cut -c1-140 >> $O/bench_dbg$d.log
done
done
