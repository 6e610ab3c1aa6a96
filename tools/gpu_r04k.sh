cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
( timeout 300 python tools/per_layer_bench.py --dtype f16 --batch 8 > gpurun_out/r04k_per_layer_f16_b8.txt 2>&1 )
tail -1 gpurun_out/r04k_per_layer_f16_b8.txt
