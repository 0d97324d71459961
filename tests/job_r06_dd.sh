#!/bin/bash
cd $GRAFT_REPO_ROOT
SF_TRACE_MIN_NS=12000 tests/trace_rebuild.sh r06_c3dd "--bed fluidised --particles 100000 --no-fluidised --no-parity" > gpurun_out/r06_trace_c3dd.txt 2>&1
tests/trace_rebuild.sh r06_l1mdd "--bed fluidised --no-fluidised --no-parity" > gpurun_out/r06_trace_l1mdd.txt 2>&1
rm -rf gpurun_out/kt_r06_*dd
cut -c1-100 gpurun_out/r06_trace_c3dd.txt gpurun_out/r06_trace_l1mdd.txt
