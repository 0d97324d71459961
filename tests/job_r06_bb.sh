#!/bin/bash
# flag words published by a kernel with release ordering: the whole suite, then the halo file three more times, A/B of the wait
cd $GRAFT_REPO_ROOT
(python -m pytest tests -m gpu -x -q 2>&1 | tail -4) > gpurun_out/r06_suite_bb.log
for k in 1 2 3; do python -m pytest tests/test_halo_gpu.py -x -q 2>&1 | tail -1; done > gpurun_out/r06_halo_bb.log 2>&1
{
tests/ab_env.sh "--bed fluidised --particles 100000 --no-fluidised --no-parity" "SF_FLAG_SPIN=0" "SF_FLAG_SPIN=1" "SF_FLAG_SPIN=0" "SF_FLAG_SPIN=1"
tests/ab_env.sh "--particles 10000 --no-fluidised --no-parity" "SF_FLAG_SPIN=0" "SF_FLAG_SPIN=1" "SF_FLAG_SPIN=0" "SF_FLAG_SPIN=1"
} > gpurun_out/r06_flag_publish_ab.txt 2>&1
tail -2 gpurun_out/r06_suite_bb.log; cat gpurun_out/r06_halo_bb.log gpurun_out/r06_flag_publish_ab.txt
