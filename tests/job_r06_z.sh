#!/bin/bash
# packed 1 M bed: one lane per atom against two / four lanes on consecutive records (list build by trace; whole run)
cd $GRAFT_REPO_ROOT
for q in 0 2 4; do
SF_BUILD_QUAD=$q tests/trace_rebuild.sh r06_p1mz$q "--no-fluidised --no-parity" > gpurun_out/r06_trace_p1mz$q.txt 2>&1
echo "packed SF_BUILD_QUAD=$q: $(grep -h 'rebuild:\|k_build_neigh' gpurun_out/r06_trace_p1mz$q.txt | tr '\n' ' ' | cut -c1-150)"
done
SF_BUILD_QUAD=2 tests/trace_rebuild.sh r06_l1mz2 "--bed fluidised --no-fluidised --no-parity" > gpurun_out/r06_trace_l1mz2.txt 2>&1
echo "loose SF_BUILD_QUAD=2: $(grep -h 'rebuild:\|k_build_neigh' gpurun_out/r06_trace_l1mz2.txt | tr '\n' ' ' | cut -c1-150)"
rm -rf gpurun_out/kt_r06_p1mz* gpurun_out/kt_r06_l1mz*
SF_BUILD_QUAD=2 python -m pytest tests/test_dem_gpu.py -x -q -k "four_lanes or variants_agree or periodic_images or closed_box" 2>&1 | tail -2
