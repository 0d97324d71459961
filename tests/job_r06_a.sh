#!/bin/bash
# round 6, job a: the GPU suite from test_dem_gpu on + rebuild traces of the two loose beds
cd $GRAFT_REPO_ROOT
(python -m pytest tests -m gpu -x -q 2>&1 | tail -40) > gpurun_out/r06_suite_3.log
tests/trace_rebuild.sh r06_c3 "--bed fluidised --particles 100000 --no-fluidised --no-parity" > gpurun_out/r06_trace_c3.txt 2>&1
tests/trace_rebuild.sh r06_l1m "--bed fluidised --no-fluidised --no-parity" > gpurun_out/r06_trace_l1m.txt 2>&1
rm -rf gpurun_out/kt_r06_c3 gpurun_out/kt_r06_l1m
tail -3 gpurun_out/r06_suite_3.log
