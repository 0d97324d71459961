#!/bin/bash
# host waits for the flag copy, not for the stream: parity tests, whole-run A/B, trace
cd $GRAFT_REPO_ROOT
(python -m pytest tests/test_dem_gpu.py tests/test_edge_cases_gpu.py tests/test_cloud_gpu.py tests/test_c_abi.py -x -q 2>&1 | tail -5) > gpurun_out/r06_suite_u.log
{
for rep in 1 2 3; do
tests/ab_env.sh "--bed fluidised --particles 100000 --no-fluidised --no-parity" "SF_FLAG_SPIN=0" "SF_FLAG_SPIN=1"
tests/ab_env.sh "--bed fluidised --no-fluidised --no-parity" "SF_FLAG_SPIN=0" "SF_FLAG_SPIN=1"
tests/ab_env.sh "--particles 10000 --no-fluidised --no-parity" "SF_FLAG_SPIN=0" "SF_FLAG_SPIN=1"
done
} > gpurun_out/r06_flag_spin_ab.txt 2>&1
SF_TRACE_MIN_NS=12000 tests/trace_rebuild.sh r06_c3u "--bed fluidised --particles 100000 --no-fluidised --no-parity" > gpurun_out/r06_trace_c3u.txt 2>&1
rm -rf gpurun_out/kt_r06_c3u
tail -3 gpurun_out/r06_suite_u.log; cat gpurun_out/r06_flag_spin_ab.txt gpurun_out/r06_trace_c3u.txt
