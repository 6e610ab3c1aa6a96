#!/bin/bash
# Same-box A/B of two builds of the library on bench_ops.py rows: tools/ab_ops.sh <row filter> <lib A> <lib B> [rounds]
cd $GRAFT_REPO_ROOT
f=$1; a=$2; b=$3; n=${4:-2}
for i in $(seq $n); do
  for l in $a $b; do
    echo "== $l"
    UNFLOW_LIB_PATH=$GRAFT_REPO_ROOT/$l python bench_ops.py "$f" 2>/dev/null | cut -c1-200
  done
done
