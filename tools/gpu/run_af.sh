export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02af; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o t -- python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-alt --no-parity --no-roofline --sustain-seconds 0 > $O/trace.log 2>&1
