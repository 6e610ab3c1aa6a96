#!/bin/bash
# PMC passes for the bench step (eager launches: counters need --no-graph).  usage: tools/pmc_run.sh <outdir> "<counters pass 1>" ["<pass 2>" ...]
# (gpurun refuses --pmc combined with sys/hip traces: counters only, with --kernel-trace).  Every pass runs under `timeout`:
# rocprofv3 aborts and then hangs in its finaliser for some counter combinations (TA_* together with TCP_* did).
out=$1; shift
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p $out
i=0
for c in "$@"; do
  i=$((i+1))
  timeout -s KILL ${PMC_TIMEOUT:-90} rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/pass$i -o p -- python bench.py --steps 2 --warmup 1 --no-graph --no-parity --no-roofline --no-alt --no-cpu-baseline --sustain-seconds 0 > $out/pass$i.log 2>&1
  echo "pass $i ($c): rc $?"
done
