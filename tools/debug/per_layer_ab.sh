#!/bin/bash
# per-layer table under several settings, interleaved ROUNDS times (env ROUNDS, default 2): tools/debug/per_layer_ab.sh "ENV=.." "ENV=.." ...
# -> gpurun_out/pl_<i>_<round>.txt ; tools/debug/per_layer_diff.py prints the rows that differ from the LAST setting
cd $GRAFT_REPO_ROOT
rm -f gpurun_out/pl_*_*.txt
for r in $(seq ${ROUNDS:-2}); do i=0; for cfg in "$@"; do i=$((i+1)); [ "$cfg" = "-" ] && cfg=""; env $cfg python tools/per_layer_bench.py $PL_ARGS > gpurun_out/pl_${i}_$r.txt 2>/dev/null; done; done
python tools/debug/per_layer_diff.py "$@"
