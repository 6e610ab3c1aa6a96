set -x
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 400 python -m pytest tests/test_planes_gpu.py tests/test_parity_fullsize_gpu.py -x -q 2>&1 | tail -5 > gpurun_out/r02_t9.log
Q="--steps 30 --warmup 8 --no-parity --no-alt --no-cpu-baseline --sustain-seconds 0"
for i in 1 2; do
UNFLOW_RGB4=0 timeout 120 python bench.py $Q 2>&1 | tail -1 >> gpurun_out/r02_ab_rgb4_off.log
timeout 120 python bench.py $Q 2>&1 | tail -1 >> gpurun_out/r02_ab_rgb4_on.log
UNFLOW_GATHER_MIN_KT=4 UNFLOW_GATHER_MAX_SPLIT=32 timeout 120 python bench.py $Q 2>&1 | tail -1 >> gpurun_out/r02_ab_split32.log
UNFLOW_GATHER_MIN_KT=4 timeout 120 python bench.py $Q 2>&1 | tail -1 >> gpurun_out/r02_ab_minkt4.log
done
timeout 300 python bench.py --dtype f16 --batch 8 --no-alt --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/r02_bench_f16_b8.log
timeout 300 python bench.py --flownet CSS --height 768 --width 1024 --batch 2 --no-alt --no-cpu-baseline --no-parity 2>&1 | tail -1 > gpurun_out/r02_bench_css.log
