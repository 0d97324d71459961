#!/bin/bash
# list build: integer image codes, unclamped record loads, shift-free test for waves away from periodic faces: parity, traces
cd $GRAFT_REPO_ROOT
(python -m pytest tests/test_dem_gpu.py tests/test_edge_cases_gpu.py tests/test_full_size_gpu.py tests/test_fuzz_gpu.py -x -q 2>&1 | tail -5) > gpurun_out/r06_suite_v.log
SF_TRACE_MIN_NS=12000 tests/trace_rebuild.sh r06_c3v "--bed fluidised --particles 100000 --no-fluidised --no-parity" > gpurun_out/r06_trace_c3v.txt 2>&1
tests/trace_rebuild.sh r06_l1mv "--bed fluidised --no-fluidised --no-parity" > gpurun_out/r06_trace_l1mv.txt 2>&1
tests/trace_rebuild.sh r06_p1mv "--no-fluidised --no-parity" > gpurun_out/r06_trace_p1mv.txt 2>&1
rm -rf gpurun_out/kt_r06_c3v gpurun_out/kt_r06_l1mv gpurun_out/kt_r06_p1mv
tail -3 gpurun_out/r06_suite_v.log; grep -h "rebuild:\|k_build_neigh" gpurun_out/r06_trace_c3v.txt gpurun_out/r06_trace_l1mv.txt gpurun_out/r06_trace_p1mv.txt
