cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_planes_gpu.py -m gpu -x -q -k "streamk" 2>&1 | tail -30 ) > gpurun_out/r04h_tests.txt
tail -4 gpurun_out/r04h_tests.txt
F="8x192x256x64|8x96x128x128|8x48x64x388|8x24x32x772|8x48x64x256"
( UNFLOW_OPT_STREAMK4=0 timeout 300 python tools/per_layer_bench.py --filter "$F" > gpurun_out/r04h_pl_a.txt 2>&1 )
( UNFLOW_OPT_STREAMK4=2 timeout 300 python tools/per_layer_bench.py --filter "$F" > gpurun_out/r04h_pl_b.txt 2>&1 )
( UNFLOW_OPT_STREAMK4=0 timeout 300 python tools/per_layer_bench.py --filter "$F" > gpurun_out/r04h_pl_c.txt 2>&1 )
( UNFLOW_OPT_STREAMK4=2 timeout 300 python tools/per_layer_bench.py --filter "$F" > gpurun_out/r04h_pl_d.txt 2>&1 )
for s in 0 2 0 2; do
( UNFLOW_OPT_STREAMK4=$s timeout 600 python bench.py --no-secondary --no-cpu-baseline --no-alt --no-parity > gpurun_out/r04h_bench_$s.json 2> gpurun_out/r04h_bench.err )
python3 -c "
import json;d=json.loads(open('gpurun_out/r04h_bench_$s.json').read().strip().splitlines()[-1]);print('streamk4=$s',d['value'],d['sustained_value'],d['roofline']['frac'],d['roofline']['ms_per_step_in_kernel_class'])"
done
