#!/bin/bash
# A/B of library options (csrc/options.h) on ONE GPU box: every argument is a space-separated list of ENV=VALUE settings
# (use UNFLOW_OPT_<OPTION>=<int>; "-" = defaults); each runs `bench.py --steps 30 --warmup 8` without the CPU / oracle /
# secondary legs and prints value, ms/step and the conv-family class time.
#   tools/ab_bench.sh - "UNFLOW_OPT_WGRAD_KGROUPS=0" "UNFLOW_OPT_NTAIL_SKIP=0" > gpurun_out/ab.txt
cd "$(dirname "$0")/.."
for cfg in "$@"; do
  [ "$cfg" = "-" ] && cfg=""
  line=$(env $cfg python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-alt --no-parity --no-secondary --sustain-seconds 0 2>/dev/null | grep '^{"metric"' | tail -1)
  echo "$line" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}
print('%-60s value %8.2f  ms/step %.4f  class_ms %s  frac %s' % ('${cfg:-defaults}', d['value'], d['ms_per_step'], r.get('ms_per_step_in_kernel_class'), r.get('frac')))"
done
