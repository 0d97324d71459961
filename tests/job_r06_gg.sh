#!/bin/bash
# one-launch chained scan of mid-sized cell histograms: parity tests, trace, A/B
cd $GRAFT_REPO_ROOT
(timeout 900 python -m pytest tests/test_dem_gpu.py tests/test_edge_cases_gpu.py tests/test_full_size_gpu.py tests/test_fuzz_gpu.py tests/test_cloud_gpu.py -x -q 2>&1 | tail -3) > gpurun_out/r06_suite_gg.log
SF_TRACE_MIN_NS=12000 tests/trace_rebuild.sh r06_c3gg "--bed fluidised --particles 100000 --no-fluidised --no-parity" > gpurun_out/r06_trace_c3gg.txt 2>&1
rm -rf gpurun_out/kt_r06_c3gg
{
for rep in 1 2 3; do
tests/ab_env.sh "--bed fluidised --particles 100000 --no-fluidised --no-parity" "SF_SCAN_CHAINED=0" "SF_SCAN_CHAINED=1"
done
tests/ab_env.sh "--bed fluidised --particles 300000 --no-fluidised --no-parity" "SF_SCAN_CHAINED=0" "SF_SCAN_CHAINED=1" "SF_SCAN_CHAINED=0" "SF_SCAN_CHAINED=1"
} > gpurun_out/r06_scan_chained_ab.txt 2>&1
tail -2 gpurun_out/r06_suite_gg.log; cut -c1-100 gpurun_out/r06_trace_c3gg.txt; cat gpurun_out/r06_scan_chained_ab.txt
