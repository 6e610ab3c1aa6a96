cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_planes_gpu.py -m gpu -x -q -k "streamk" 2>&1 | tail -30 ) > gpurun_out/r04b_sk_tests.txt
tail -5 gpurun_out/r04b_sk_tests.txt
for m in 0 1 2; do
  ( UNFLOW_OPT_STREAMK=$m timeout 300 python tools/per_layer_bench.py > gpurun_out/r04b_per_layer_sk$m.txt 2>&1 )
  tail -1 gpurun_out/r04b_per_layer_sk$m.txt
done
( UNFLOW_OPT_STREAMK=1 timeout 600 python bench.py --no-secondary --no-cpu-baseline --no-alt > gpurun_out/r04b_bench_sk1.json 2> gpurun_out/r04b_bench_sk1.err )
cut -c1-400 gpurun_out/r04b_bench_sk1.json
