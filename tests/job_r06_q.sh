#!/bin/bash
# single-precision shadow walk: parity tests, A/B, traces; barrier probe with buffer_inv sc1
cd $GRAFT_REPO_ROOT
(python -m pytest tests/test_dem_gpu.py tests/test_edge_cases_gpu.py tests/test_full_size_gpu.py tests/test_fuzz_gpu.py -x -q 2>&1 | tail -15) > gpurun_out/r06_suite_q.log
{
for rep in 1 2; do
tests/ab_env.sh "--bed fluidised --no-fluidised --no-parity" "SF_BUILD_SHADOW=0" "SF_BUILD_SHADOW=1"
tests/ab_env.sh "--bed fluidised --particles 100000 --no-fluidised --no-parity" "SF_BUILD_SHADOW=0" "SF_BUILD_SHADOW=1"
done
tests/ab_env.sh "--no-fluidised --no-parity" "SF_BUILD_SHADOW=0" "SF_BUILD_SHADOW=1"
} > gpurun_out/r06_shadow_ab.txt 2>&1
SF_TRACE_MIN_NS=12000 tests/trace_rebuild.sh r06_c3q "--bed fluidised --particles 100000 --no-fluidised --no-parity" > gpurun_out/r06_trace_c3q.txt 2>&1
tests/trace_rebuild.sh r06_l1mq "--bed fluidised --no-fluidised --no-parity" > gpurun_out/r06_trace_l1mq.txt 2>&1
rm -rf gpurun_out/kt_r06_c3q gpurun_out/kt_r06_l1mq
(cd /tmp && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o xcd_barrier $GRAFT_REPO_ROOT/tests/micro/xcd_barrier.hip && timeout 120 ./xcd_barrier 2000) > gpurun_out/r06_xcd_barrier.txt 2>&1
tail -3 gpurun_out/r06_suite_q.log; cat gpurun_out/r06_shadow_ab.txt gpurun_out/r06_trace_c3q.txt gpurun_out/r06_trace_l1mq.txt gpurun_out/r06_xcd_barrier.txt
