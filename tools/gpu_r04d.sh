cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
F="8x48x64x476|8x24x32x512|8x48x64x256|8x96x128x128|8x24x32x772"
( UNFLOW_OPT_STREAMK=0 timeout 300 python tools/per_layer_bench.py --filter "$F" > gpurun_out/r04d_pl_sk0.txt 2>&1 )
for g in 1 8 16 32 96; do
( UNFLOW_OPT_STREAMK=2 UNFLOW_OPT_STREAMK_GROUPS=$g timeout 300 python tools/per_layer_bench.py --filter "$F" > gpurun_out/r04d_pl_sk2_g$g.txt 2>&1 )
done
for s in 0 1 0 1; do
( UNFLOW_OPT_STREAMK=$s timeout 600 python bench.py --no-secondary --no-cpu-baseline --no-alt --no-parity > gpurun_out/r04d_bench_sk$s.json 2> gpurun_out/r04d_bench.err )
python3 -c "
import json;d=json.loads(open('gpurun_out/r04d_bench_sk$s.json').read().strip().splitlines()[-1]);print('streamk=$s',d['value'],d['sustained_value'],d['roofline']['frac'],d['roofline']['ms_per_step_in_kernel_class'])"
done
