#!/bin/bash
# Builds the library with EXTRA compiler flags into tools/debug/lib_<name>.so (objects in /tmp), for same-box A/Bs through
# UNFLOW_LIB_PATH.  usage: tools/build_variant.sh <name> <extra flags...>      e.g.  build_variant.sh ilp -mllvm -amdgpu-sched-strategy=max-ilp
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
out=/tmp/unflow_variant_$name; mkdir -p $out
FLAGS="--offload-arch=gfx950 -O3 -fno-slp-vectorize -fno-vectorize -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden"
pids=()
for src in $root/unflow_amd/csrc/*.hip; do
  obj=$out/$(basename ${src%.hip}).o
  /opt/rocm/bin/hipcc $FLAGS "$@" -Rpass-analysis=kernel-resource-usage -c $src -o $obj 2> $out/$(basename ${src%.hip}).log &
  pids+=($!)
done
rc=0; for p in "${pids[@]}"; do wait $p || rc=1; done
[ $rc = 0 ] || { echo "compile failed"; grep -l "error" $out/*.log; exit 1; }
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $root/tools/debug/lib_$name.so $out/*.o -ldl || exit 1
echo "kernels with scratch:"; grep -h "ScratchSize" $out/*.log | grep -v ": 0 " | wc -l
echo $root/tools/debug/lib_$name.so
