#!/bin/bash
# k_build_neigh_quad: bit-identity + parity tests, traces, whole-run A/B
cd $GRAFT_REPO_ROOT
(python -m pytest tests/test_dem_gpu.py tests/test_edge_cases_gpu.py tests/test_full_size_gpu.py tests/test_fuzz_gpu.py -x -q 2>&1 | tail -12) > gpurun_out/r06_suite_w.log
for q in 4 8; do
SF_BUILD_QUAD=$q SF_TRACE_MIN_NS=12000 tests/trace_rebuild.sh r06_c3w$q "--bed fluidised --particles 100000 --no-fluidised --no-parity" > gpurun_out/r06_trace_c3w$q.txt 2>&1
SF_BUILD_QUAD=$q tests/trace_rebuild.sh r06_l1mw$q "--bed fluidised --no-fluidised --no-parity" > gpurun_out/r06_trace_l1mw$q.txt 2>&1
SF_BUILD_QUAD=$q tests/trace_rebuild.sh r06_p1mw$q "--no-fluidised --no-parity" > gpurun_out/r06_trace_p1mw$q.txt 2>&1
done
rm -rf gpurun_out/kt_r06_c3w* gpurun_out/kt_r06_l1mw* gpurun_out/kt_r06_p1mw*
{
for rep in 1 2; do
tests/ab_env.sh "--bed fluidised --particles 100000 --no-fluidised --no-parity" "SF_BUILD_QUAD=0" "SF_BUILD_QUAD=4" "SF_BUILD_QUAD=8"
tests/ab_env.sh "--bed fluidised --no-fluidised --no-parity" "SF_BUILD_QUAD=0" "SF_BUILD_QUAD=4" "SF_BUILD_QUAD=8"
done
tests/ab_env.sh "--no-fluidised --no-parity" "SF_BUILD_QUAD=0" "SF_BUILD_QUAD=4" "SF_BUILD_QUAD=8"
} > gpurun_out/r06_quad_ab.txt 2>&1
tail -5 gpurun_out/r06_suite_w.log; grep -h "rebuild:\|k_build_neigh" gpurun_out/r06_trace_*w0.txt gpurun_out/r06_trace_*w1.txt; cat gpurun_out/r06_quad_ab.txt
