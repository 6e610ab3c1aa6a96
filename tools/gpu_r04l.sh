cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 120 ./tools/microbench/pk_mul_hazard > gpurun_out/r04l_pk_mul_hazard.txt 2>&1
cat gpurun_out/r04l_pk_mul_hazard.txt
