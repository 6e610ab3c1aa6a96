set -x
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02u; mkdir -p $O
Q="--dtype f16 --batch 8 --steps 30 --warmup 8 --no-parity --no-alt --no-cpu-baseline --sustain-seconds 0 --no-roofline"
for i in 1 2; do
timeout 120 python bench.py $Q 2>&1 | tail -1 | cut -c60-130 >> $O/f16_default.log
UNFLOW_XCD_SWIZZLE=0 timeout 120 python bench.py $Q 2>&1 | tail -1 | cut -c60-130 >> $O/f16_noxcd.log
UNFLOW_GATHER_TILE2D=0 timeout 120 python bench.py $Q 2>&1 | tail -1 | cut -c60-130 >> $O/f16_1d.log
UNFLOW_XCD_ORDER=1 timeout 120 python bench.py $Q 2>&1 | tail -1 | cut -c60-130 >> $O/f16_o1.log
UNFLOW_XCD_ORDER=0 timeout 120 python bench.py $Q 2>&1 | tail -1 | cut -c60-130 >> $O/f16_o0.log
done
timeout 200 python tools/per_layer_bench.py --dtype f16 --batch 8 > $O/per_layer_f16.txt 2>$O/err.txt
UNFLOW_XCD_SWIZZLE=0 timeout 200 python tools/per_layer_bench.py --dtype f16 --batch 8 > $O/per_layer_f16_noxcd.txt 2>$O/err2.txt
