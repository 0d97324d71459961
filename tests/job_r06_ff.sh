#!/bin/bash
# rank + permutation in one launch for small beds: parity tests, A/B
cd $GRAFT_REPO_ROOT
(python -m pytest tests/test_dem_gpu.py tests/test_edge_cases_gpu.py tests/test_full_size_gpu.py tests/test_fuzz_gpu.py tests/test_cloud_gpu.py -x -q 2>&1 | tail -3) > gpurun_out/r06_suite_ff.log
{
for rep in 1 2 3; do
tests/ab_env.sh "--bed fluidised --particles 100000 --no-fluidised --no-parity" "SF_RANK_PERMUTE=0" "SF_RANK_PERMUTE=1"
done
tests/ab_env.sh "--bed fluidised --no-fluidised --no-parity" "SF_RANK_PERMUTE=0" "SF_RANK_PERMUTE=1" "SF_RANK_PERMUTE=0" "SF_RANK_PERMUTE=1"
tests/ab_env.sh "--particles 10000 --no-fluidised --no-parity" "SF_RANK_PERMUTE=0" "SF_RANK_PERMUTE=1" "SF_RANK_PERMUTE=0" "SF_RANK_PERMUTE=1"
} > gpurun_out/r06_rank_permute_ab.txt 2>&1
tail -2 gpurun_out/r06_suite_ff.log; cat gpurun_out/r06_rank_permute_ab.txt
