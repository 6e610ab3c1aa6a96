export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02y; mkdir -p $O
timeout 500 python tools/debug/wgstream_flake.py 2>&1 | grep -v "^W2026\|^E2026" | tail -60 > $O/flake.log
UNFLOW_WGRAD_DMA=0 timeout 300 python tools/debug/wgstream_flake.py 2>&1 | grep -v "^W2026\|^E2026" | tail -30 > $O/flake_nodma.log
