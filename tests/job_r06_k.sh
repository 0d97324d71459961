#!/bin/bash
# list build with NL lanes per atom + linear row walk (5 waves per SIMD at one lane): parity / bit-identity tests, A/B, traces;
# the one-XCD barrier micro-benchmark
cd $GRAFT_REPO_ROOT
(python -m pytest tests/test_dem_gpu.py tests/test_edge_cases_gpu.py tests/test_full_size_gpu.py tests/test_fuzz_gpu.py -x -q 2>&1 | tail -15) > gpurun_out/r06_suite_k.log
{
for rep in 1 2; do
tests/ab_env.sh "--bed fluidised --no-fluidised --no-parity" "SF_BUILD_LPA=1" "SF_BUILD_LPA=2"
tests/ab_env.sh "--bed fluidised --particles 100000 --no-fluidised --no-parity" "SF_BUILD_LPA=1" "SF_BUILD_LPA=2" "SF_BUILD_LPA=4"
tests/ab_env.sh "--bed fluidised --particles 300000 --no-fluidised --no-parity" "SF_BUILD_LPA=1" "SF_BUILD_LPA=2" "SF_BUILD_LPA=4"
done
} > gpurun_out/r06_build_lpa_ab.txt 2>&1
SF_TRACE_MIN_NS=12000 tests/trace_rebuild.sh r06_c3k "--bed fluidised --particles 100000 --no-fluidised --no-parity" > gpurun_out/r06_trace_c3k.txt 2>&1
tests/trace_rebuild.sh r06_l1mk "--bed fluidised --no-fluidised --no-parity" > gpurun_out/r06_trace_l1mk.txt 2>&1
rm -rf gpurun_out/kt_r06_c3k gpurun_out/kt_r06_l1mk
(cd /tmp && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o xcd_barrier $GRAFT_REPO_ROOT/tests/micro/xcd_barrier.hip && timeout 120 ./xcd_barrier 2000) > gpurun_out/r06_xcd_barrier.txt 2>&1
tail -3 gpurun_out/r06_suite_k.log; cat gpurun_out/r06_build_lpa_ab.txt gpurun_out/r06_trace_c3k.txt gpurun_out/r06_trace_l1mk.txt gpurun_out/r06_xcd_barrier.txt
