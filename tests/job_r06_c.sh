#!/bin/bash
cd $GRAFT_REPO_ROOT
(SF_LPA=1 SF_PERSIST=1 python -m pytest tests/test_dem_gpu.py -x -q 2>&1 | tail -5) > gpurun_out/r06_persist_parity.log
for rep in 1 2; do
  tests/ab_env.sh "--no-fluidised --no-parity" SF_PERSIST=0 SF_PERSIST=1
done > gpurun_out/r06_persist_ab2.txt 2>&1
cat gpurun_out/r06_persist_parity.log gpurun_out/r06_persist_ab2.txt
