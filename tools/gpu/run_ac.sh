export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02ac; mkdir -p $O
Q="--steps 30 --warmup 8 --no-parity --no-alt --no-cpu-baseline --sustain-seconds 0 --no-roofline"
for i in 1 2 3; do
timeout 120 python bench.py $Q 2>&1 | tail -1 | cut -c60-130 >> $O/bench_base.log
UNFLOW_LIB_PATH=$GRAFT_REPO_ROOT/scratch/nt/libunflow_hip_nt.so timeout 120 python bench.py $Q 2>&1 | tail -1 | cut -c60-130 >> $O/bench_nt.log
done
