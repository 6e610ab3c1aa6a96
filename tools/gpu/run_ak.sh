export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02ak; mkdir -p $O
Q="--steps 30 --warmup 8 --no-parity --no-alt --no-cpu-baseline --sustain-seconds 0 --no-roofline"
run() { echo "$1: $(env $1 timeout 120 python bench.py $Q 2>&1 | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())[\"value\"])")" >> $O/sweep.log; }
for rep in 1 2; do
run "X=0"
run "UNFLOW_WGRAD_GROUP=2"
run "UNFLOW_WGRAD_GROUP=3"
run "UNFLOW_WGRAD_GROUP=6"
run "UNFLOW_WGRAD_GROUP=8"
run "UNFLOW_WGRAD_MIN_KT=4"
run "UNFLOW_WGRAD_MIN_KT=16"
run "UNFLOW_GATHER_MIN_KT=4"
run "UNFLOW_GATHER_MIN_KT=16"
run "UNFLOW_GATHER_MAX_SPLIT=8"
run "UNFLOW_GATHER_MAX_SPLIT=32"
done
