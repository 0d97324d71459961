#!/bin/bash
# histogram atomics by runs of equal keys: parity tests, traces
cd $GRAFT_REPO_ROOT
(python -m pytest tests/test_dem_gpu.py tests/test_edge_cases_gpu.py tests/test_full_size_gpu.py tests/test_fuzz_gpu.py -x -q 2>&1 | tail -4) > gpurun_out/r06_suite_aa.log
tests/trace_rebuild.sh r06_l1maa "--bed fluidised --no-fluidised --no-parity" > gpurun_out/r06_trace_l1maa.txt 2>&1
tests/trace_rebuild.sh r06_p1maa "--no-fluidised --no-parity" > gpurun_out/r06_trace_p1maa.txt 2>&1
rm -rf gpurun_out/kt_r06_*aa
tail -2 gpurun_out/r06_suite_aa.log; cut -c1-90 gpurun_out/r06_trace_l1maa.txt; grep -h "rebuild:\|k_pbc_keys\|k_key_place" gpurun_out/r06_trace_p1maa.txt | cut -c1-90
