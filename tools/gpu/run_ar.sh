#!/bin/bash
# filter-gradient block count A/B (partial-sum traffic vs parallelism)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for v in 0 200 150 100 0; do
  echo -n "UNFLOW_WGRAD_SLOTS_PCT=$v: "
  UNFLOW_WGRAD_SLOTS_PCT=$v timeout 120 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-alt --no-parity --no-roofline --sustain-seconds 0 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['value'])"
done | tee gpurun_out/ar_slots.txt
