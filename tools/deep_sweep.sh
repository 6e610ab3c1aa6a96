#!/bin/bash
cd $GRAFT_REPO_ROOT
F="8x12x16x512|8x6x8x1024|8x24x32x512 8x12x16x512|8x12x16x1028 8x24x32x256"
for cfg in "-" "UNFLOW_OPT_GATHER_MAX_SPLIT=4" "UNFLOW_OPT_GATHER_MAX_SPLIT=8" "UNFLOW_OPT_GATHER_MAX_SPLIT=32 UNFLOW_OPT_GATHER_MIN_KT=4" "UNFLOW_OPT_GATHER_MAX_SPLIT=1" "UNFLOW_OPT_FUSED_SPLITK=16" "UNFLOW_OPT_FUSED_SPLITK=4"; do
  [ "$cfg" = "-" ] && cfg=""
  echo "== ${cfg:-defaults}"
  env $cfg python tools/per_layer_bench.py --filter "$F" 2>&1 | grep -v "^pass\|filter"
done
