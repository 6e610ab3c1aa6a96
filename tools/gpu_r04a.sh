cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/r04a_tests.txt
( timeout 900 python bench.py > gpurun_out/r04a_bench.json 2> gpurun_out/r04a_bench.err ) 
( timeout 300 python tools/per_layer_bench.py > gpurun_out/r04a_per_layer.txt 2>&1 )
tail -3 gpurun_out/r04a_tests.txt; cut -c1-600 gpurun_out/r04a_bench.json; tail -2 gpurun_out/r04a_per_layer.txt
