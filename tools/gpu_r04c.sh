cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_planes_gpu.py -m gpu -x -q -k "streamk" 2>&1 | tail -30 ) > gpurun_out/r04c_sk_tests.txt
tail -3 gpurun_out/r04c_sk_tests.txt
F="8x48x64x476|8x24x32x512            8x24x32x512|8x48x64x256            8x24x32x512|8x96x128x128           8x48x64x256|8x24x32x772            8x48x64x128"
( UNFLOW_OPT_STREAMK=0 timeout 300 python tools/per_layer_bench.py --filter "$F" > gpurun_out/r04c_pl_sk0.txt 2>&1 )
( UNFLOW_OPT_STREAMK=2 UNFLOW_OPT_STREAMK_GROUPS=1 timeout 300 python tools/per_layer_bench.py --filter "$F" > gpurun_out/r04c_pl_sk2_g1.txt 2>&1 )
( UNFLOW_OPT_STREAMK=2 UNFLOW_OPT_STREAMK_GROUPS=8 timeout 300 python tools/per_layer_bench.py --filter "$F" > gpurun_out/r04c_pl_sk2_g8.txt 2>&1 )
( UNFLOW_OPT_STREAMK=2 UNFLOW_OPT_STREAMK_GROUPS=16 timeout 300 python tools/per_layer_bench.py --filter "$F" > gpurun_out/r04c_pl_sk2_g16.txt 2>&1 )
for f in sk0 sk2_g1 sk2_g8 sk2_g16; do echo $f; grep -v "^pass" gpurun_out/r04c_pl_$f.txt | awk '{printf "%s %s %s %s %s | ", $1,$2,$3,$5,$6} END {print ""}'; done
( UNFLOW_OPT_STREAMK=1 timeout 600 python bench.py --no-secondary --no-cpu-baseline --no-alt > gpurun_out/r04c_bench_sk1.json 2> gpurun_out/r04c_bench_sk1.err )
cut -c1-200 gpurun_out/r04c_bench_sk1.json
