#!/bin/bash
cd $GRAFT_REPO_ROOT
(python -m pytest tests -m gpu -x -q 2>&1 | tail -40) > gpurun_out/r06_suite_5.log
SF_TRACE_MIN_NS=12000 tests/trace_rebuild.sh r06_c3b "--bed fluidised --particles 100000 --no-fluidised --no-parity" > gpurun_out/r06_trace_c3b.txt 2>&1
tests/trace_rebuild.sh r06_l1mb "--bed fluidised --no-fluidised --no-parity" > gpurun_out/r06_trace_l1mb.txt 2>&1
rm -rf gpurun_out/kt_r06_c3b gpurun_out/kt_r06_l1mb
tail -3 gpurun_out/r06_suite_5.log
