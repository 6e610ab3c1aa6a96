cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_planes_gpu.py tests/test_ops_gpu.py -m gpu -x -q -k "streamk or forward_warp" 2>&1 | tail -30 ) > gpurun_out/r04g_tests.txt
tail -4 gpurun_out/r04g_tests.txt
( timeout 300 python bench_ops.py 2>/dev/null | grep forward_warp > gpurun_out/r04g_fw.jsonl ); cut -c1-170 gpurun_out/r04g_fw.jsonl
F="8x384x512x4|8x192x256x64|8x96x128x128|8x48x64x388"
( UNFLOW_OPT_STREAMK=0 timeout 300 python tools/per_layer_bench.py --filter "$F" > gpurun_out/r04g_pl_sk0.txt 2>&1 )
( UNFLOW_OPT_STREAMK=1 timeout 300 python tools/per_layer_bench.py --filter "$F" > gpurun_out/r04g_pl_sk1.txt 2>&1 )
( UNFLOW_OPT_STREAMK=2 timeout 300 python tools/per_layer_bench.py --filter "$F" > gpurun_out/r04g_pl_sk2.txt 2>&1 )
for s in 0 1 0 1; do
( UNFLOW_OPT_STREAMK=$s timeout 600 python bench.py --no-secondary --no-cpu-baseline --no-alt --no-parity > gpurun_out/r04g_bench_sk$s.json 2> gpurun_out/r04g_bench.err )
python3 -c "
import json;d=json.loads(open('gpurun_out/r04g_bench_sk$s.json').read().strip().splitlines()[-1]);print('streamk=$s',d['value'],d['sustained_value'],d['roofline']['frac'],d['roofline']['ms_per_step_in_kernel_class'])"
done
