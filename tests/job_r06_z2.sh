#!/bin/bash
# two against four lanes per atom in the list build: bitwise test, traces, whole run
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_dem_gpu.py -x -q -k "four_lanes or variants_agree" 2>&1 | tail -2
for q in 2 4; do
SF_BUILD_QUAD=$q SF_TRACE_MIN_NS=12000 tests/trace_rebuild.sh r06_c3z$q "--bed fluidised --particles 100000 --no-fluidised --no-parity" > gpurun_out/r06_trace_c3z$q.txt 2>&1
SF_BUILD_QUAD=$q tests/trace_rebuild.sh r06_l1mz$q "--bed fluidised --no-fluidised --no-parity" > gpurun_out/r06_trace_l1mz$q.txt 2>&1
SF_BUILD_QUAD=$q tests/trace_rebuild.sh r06_p1mz$q "--no-fluidised --no-parity" > gpurun_out/r06_trace_p1mz$q.txt 2>&1
echo "SF_BUILD_QUAD=$q: $(grep -h 'k_build_neigh' gpurun_out/r06_trace_c3z$q.txt gpurun_out/r06_trace_l1mz$q.txt gpurun_out/r06_trace_p1mz$q.txt | awk '{print $4}' | tr '\n' ' ')"
done
rm -rf gpurun_out/kt_r06_*z*
for rep in 1 2; do
tests/ab_env.sh "--bed fluidised --particles 100000 --no-fluidised --no-parity" "SF_BUILD_QUAD=2" "SF_BUILD_QUAD=4"
tests/ab_env.sh "--bed fluidised --no-fluidised --no-parity" "SF_BUILD_QUAD=2" "SF_BUILD_QUAD=4"
done
