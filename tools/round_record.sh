# Round record (run at the end of a round; copy what is to be judged from gpurun_out/rec_* into profiles/rNN_*): full GPU suite, the default bench line, per-layer table, op-level table, rocprofv3 kernel stats, PMC traffic,
# the forced one-rank comm record and the two-ranks-on-one-GPU gloo run.  Everything lands under gpurun_out/rec_*.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
( timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -8 ) > gpurun_out/rec_gpu_tests.txt
tail -2 gpurun_out/rec_gpu_tests.txt
( timeout 900 python bench.py > gpurun_out/rec_bench_line.json 2> gpurun_out/rec_bench.err )
cut -c1-300 gpurun_out/rec_bench_line.json
( timeout 300 python tools/per_layer_bench.py > gpurun_out/rec_per_layer.txt 2>&1 ); tail -1 gpurun_out/rec_per_layer.txt
( timeout 300 python tools/per_layer_bench.py --dtype f16 --batch 8 > gpurun_out/rec_per_layer_f16_b8.txt 2>&1 ); tail -1 gpurun_out/rec_per_layer_f16_b8.txt
( timeout 400 python bench_ops.py > gpurun_out/rec_bench_ops.jsonl 2>/dev/null ); wc -l gpurun_out/rec_bench_ops.jsonl
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/rec_prof -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 6 --no-secondary --no-cpu-baseline --no-alt --no-parity > $GRAFT_REPO_ROOT/gpurun_out/rec_bench_under_rocprof.json 2> $GRAFT_REPO_ROOT/gpurun_out/rec_rocprof.err )
find gpurun_out/rec_prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/rec_bench_kernel_stats.csv
head -5 gpurun_out/rec_bench_kernel_stats.csv | cut -c1-160
PMC_TIMEOUT=150 bash tools/pmc_run.sh gpurun_out/rec_pmc "FETCH_SIZE" "WRITE_SIZE"
python tools/pmc_traffic.py gpurun_out/rec_pmc/pass1 gpurun_out/rec_pmc/pass2 > gpurun_out/rec_pmc_traffic.json 2> gpurun_out/rec_pmc_traffic.err; head -c 300 gpurun_out/rec_pmc_traffic.json; grep -A4 conv_family gpurun_out/rec_pmc_traffic.json
( RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29519 UNFLOW_FORCE_REDUCER=1 timeout 600 python bench.py --gpus 1 --no-secondary --no-cpu-baseline --no-alt --no-parity --no-roofline > gpurun_out/rec_comm_record_forced_world1.json 2> gpurun_out/rec_comm.err )
# two ranks on the ONE GPU of the box, started by bench.py itself (no launcher around it: the command shape the driver uses);
# RCCL refuses two ranks on one device: gloo transport, same host code / streams / buckets
( UNFLOW_DIST_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 10 --warmup 3 --no-secondary --no-cpu-baseline --no-alt --no-parity --no-roofline > gpurun_out/rec_2ranks.out 2> gpurun_out/rec_2ranks.err )
grep '^{"metric"' gpurun_out/rec_2ranks.out > gpurun_out/rec_bench_2ranks_gloo_one_gpu.json; tail -c 700 gpurun_out/rec_bench_2ranks_gloo_one_gpu.json
rm -rf gpurun_out/rec_prof gpurun_out/rec_pmc/pass*/ 2>/dev/null; ls gpurun_out | head -30
# round 6: the single-stream rocprof record (roofline_check reproduces the class time from the tracked CSV), the step timeline, the census VALU roofline
bash tools/serial_profile.sh rec > /dev/null 2>&1; tail -3 gpurun_out/rec_serial_roofline_check.txt
bash tools/step_trace.sh rec > /dev/null 2>&1; python tools/step_timeline.py gpurun_out/rec_step_trace.csv > gpurun_out/rec_step_timeline.txt 2>&1; head -2 gpurun_out/rec_step_timeline.txt
bash tools/census_valu.sh rec_census > /dev/null 2>&1
