export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 300 python tools/debug/wgstream_diff.py 2 2>&1 | grep -v "^W2026\|^E2026" | tail -30 > gpurun_out/r02_wgdiff.log
