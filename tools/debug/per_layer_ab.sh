#!/bin/bash
# per-layer table under two settings, interleaved twice: tools/debug/per_layer_ab.sh "ENV=.." "ENV=.."  -> gpurun_out/pl_<i>_<round>.txt
cd $GRAFT_REPO_ROOT
for r in 1 2; do i=0; for cfg in "$@"; do i=$((i+1)); [ "$cfg" = "-" ] && cfg=""; env $cfg python tools/per_layer_bench.py > gpurun_out/pl_${i}_$r.txt 2>/dev/null; tail -1 gpurun_out/pl_${i}_$r.txt; done; done
