cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_train_gpu.py tests/test_ops_gpu.py tests/test_fullsize_gpu.py -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/r04f_tests.txt
tail -3 gpurun_out/r04f_tests.txt
( timeout 600 python bench_ops.py > gpurun_out/r04f_bench_ops.jsonl 2> gpurun_out/r04f_bench_ops.err )
grep forward_warp gpurun_out/r04f_bench_ops.jsonl | cut -c1-200
( RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29517 UNFLOW_FORCE_REDUCER=1 timeout 600 python bench.py --gpus 1 --no-secondary --no-cpu-baseline --no-alt --no-parity --no-roofline > gpurun_out/r04f_comm_forced_world1.json 2> gpurun_out/r04f_comm.err )
python3 -c "
import json;d=json.loads(open('gpurun_out/r04f_comm_forced_world1.json').read().strip().splitlines()[-1]);print(d['value']);print(json.dumps(d.get('comm'),indent=1)[:2500])"
tail -5 gpurun_out/r04f_comm.err
