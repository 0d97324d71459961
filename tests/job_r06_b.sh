#!/bin/bash
cd $GRAFT_REPO_ROOT
SF_TRACE_MIN_NS=12000 tests/trace_rebuild.sh r06_c3 "--bed fluidised --particles 100000 --no-fluidised --no-parity" > gpurun_out/r06_trace_c3.txt 2>&1
rm -rf gpurun_out/kt_r06_c3
(SF_LPA=1 SF_PERSIST=1 python -m pytest tests/test_dem_gpu.py -x -q 2>&1 | tail -5) > gpurun_out/r06_persist_parity.log
for rep in 1 2; do
  tests/ab_env.sh "--no-fluidised --no-parity" SF_PERSIST=0 SF_PERSIST=1 "SF_PERSIST=1 SF_PERSIST_WAVES=448" "SF_PERSIST=1 SF_PERSIST_WAVES=512"
done > gpurun_out/r06_persist_ab.txt 2>&1
cat gpurun_out/r06_persist_parity.log gpurun_out/r06_persist_ab.txt
