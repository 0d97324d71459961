#!/bin/bash
# F_MAXNEIGH atomic only by waves that raise it: whole-run A/B of the lanes per atom, traces
cd $GRAFT_REPO_ROOT
{
for rep in 1 2; do
tests/ab_env.sh "--bed fluidised --no-fluidised --no-parity" "SF_BUILD_LPA=1" "SF_BUILD_LPA=2"
tests/ab_env.sh "--bed fluidised --particles 100000 --no-fluidised --no-parity" "SF_BUILD_LPA=1" "SF_BUILD_LPA=2" "SF_BUILD_LPA=4"
tests/ab_env.sh "--bed fluidised --particles 300000 --no-fluidised --no-parity" "SF_BUILD_LPA=1" "SF_BUILD_LPA=2" "SF_BUILD_LPA=4"
done
} > gpurun_out/r06_build_lpa_ab3.txt 2>&1
for lpa in 1 4; do
SF_BUILD_LPA=$lpa SF_TRACE_MIN_NS=12000 tests/trace_rebuild.sh r06_c3o$lpa "--bed fluidised --particles 100000 --no-fluidised --no-parity" > gpurun_out/r06_trace_c3o$lpa.txt 2>&1
done
tests/trace_rebuild.sh r06_l1mo "--bed fluidised --no-fluidised --no-parity" > gpurun_out/r06_trace_l1mo.txt 2>&1
rm -rf gpurun_out/kt_r06_c3o1 gpurun_out/kt_r06_c3o4 gpurun_out/kt_r06_l1mo
cat gpurun_out/r06_build_lpa_ab3.txt gpurun_out/r06_trace_c3o1.txt gpurun_out/r06_trace_c3o4.txt gpurun_out/r06_trace_l1mo.txt
