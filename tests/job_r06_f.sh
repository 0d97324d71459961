#!/bin/bash
cd $GRAFT_REPO_ROOT
B="--steps 4 --warmup 1 --no-cpu-baseline --no-coupled --no-configs --no-kernel-profile --no-fluidised --no-parity"
for args in "--bed fluidised" "--bed fluidised --particles 100000" ""; do
  echo "== $args"
  SF_LIB_PATH=$GRAFT_REPO_ROOT/sedifoam_amd/libsedifoam_amd_bph.so python bench.py $B $args 2>&1 >/dev/null | grep "k_build_neigh"
done > gpurun_out/r06_build_phase.txt 2>&1
(python -m pytest tests/test_dem_gpu.py tests/test_edge_cases_gpu.py tests/test_full_size_gpu.py tests/test_cloud_gpu.py tests/test_fuzz_gpu.py -x -q 2>&1 | tail -15) > gpurun_out/r06_suite_6.log
tests/ab_env.sh "--bed fluidised --no-fluidised --no-parity" SF_GHOST_FREE=1 > gpurun_out/r06_loose_quick.txt 2>&1
tests/ab_env.sh "--bed fluidised --particles 100000 --no-fluidised --no-parity" SF_GHOST_FREE=1 >> gpurun_out/r06_loose_quick.txt 2>&1
cat gpurun_out/r06_build_phase.txt gpurun_out/r06_loose_quick.txt; tail -3 gpurun_out/r06_suite_6.log
