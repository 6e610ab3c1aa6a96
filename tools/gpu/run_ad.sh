export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02ad; mkdir -p $O
for d in 0 256 512 1; do
UNFLOW_DBG=$d timeout 200 python tools/per_layer_bench.py > $O/per_layer_dbg$d.txt 2>$O/err$d.txt
done
