#!/bin/bash
# quad list build with the old tags shared inside the quad: bitwise / parity tests, trace, A/B against the env-forced one-lane build
cd $GRAFT_REPO_ROOT
(python -m pytest tests/test_dem_gpu.py tests/test_edge_cases_gpu.py tests/test_fuzz_gpu.py -x -q 2>&1 | tail -5) > gpurun_out/r06_suite_y.log
SF_TRACE_MIN_NS=12000 tests/trace_rebuild.sh r06_c3y "--bed fluidised --particles 100000 --no-fluidised --no-parity" > gpurun_out/r06_trace_c3y.txt 2>&1
tests/trace_rebuild.sh r06_l1my "--bed fluidised --no-fluidised --no-parity" > gpurun_out/r06_trace_l1my.txt 2>&1
rm -rf gpurun_out/kt_r06_c3y gpurun_out/kt_r06_l1my
tail -3 gpurun_out/r06_suite_y.log; grep -h "rebuild:\|k_build_neigh" gpurun_out/r06_trace_c3y.txt gpurun_out/r06_trace_l1my.txt
