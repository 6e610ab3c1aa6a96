export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02z; mkdir -p $O
UNFLOW_LIB_PATH=$GRAFT_REPO_ROOT/scratch/noslp/libunflow_hip_noslp.so timeout 900 python tools/debug/wgstream_flake.py 2>&1 | grep -v "^W2026\|^E2026" | tail -12 > $O/flake_noslp.log
Q="--steps 30 --warmup 8 --no-parity --no-alt --no-cpu-baseline --sustain-seconds 0 --no-roofline"
for i in 1 2; do
timeout 120 python bench.py $Q 2>&1 | tail -1 | cut -c60-130 >> $O/bench_slp.log
UNFLOW_LIB_PATH=$GRAFT_REPO_ROOT/scratch/noslp/libunflow_hip_noslp.so timeout 120 python bench.py $Q 2>&1 | tail -1 | cut -c60-130 >> $O/bench_noslp.log
done
