export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02an; mkdir -p $O
Q="--steps 30 --warmup 8 --no-parity --no-alt --no-cpu-baseline --sustain-seconds 0 --no-roofline"
run() { echo "$1: $(env $1 timeout 120 python bench.py $Q 2>&1 | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())[\"value\"])")" >> $O/sweep.log; }
for rep in 1 2; do
run "UNFLOW_WGRAD_GROUP=6"
run "UNFLOW_WGRAD_GROUP=5"
run "UNFLOW_WGRAD_GROUP=7"
run "UNFLOW_WGRAD_GROUP=10"
run "UNFLOW_WGRAD_GROUP=12"
run "UNFLOW_WGRAD_GROUP=16"
run "UNFLOW_PLANE_PAD=16"
run "UNFLOW_PLANE_PAD=64"
done
