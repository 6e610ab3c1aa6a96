cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
F="8x192x256x64|8x96x128x128|8x48x64x388|8x24x32x772|8x48x64x256|8x24x32x512"
for ms in 16 1 2 3 16 1 2; do
( UNFLOW_OPT_HALO_MAX_SPLIT=$ms timeout 300 python tools/per_layer_bench.py --filter "$F" --reps 20 > gpurun_out/r04j_pl_ms${ms}_$RANDOM.txt 2>&1 )
done
ls gpurun_out/r04j_pl_*
