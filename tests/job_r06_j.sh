#!/bin/bash
# list build at four waves per SIMD (row / cell paths split, old tags loaded behind the walk, parking rows sized by the last
# list), flag resets folded into k_pbc_keys / k_back_slots, one-block scan: parity tests, whole-run A/B, rebuild traces, C5_wide
cd $GRAFT_REPO_ROOT
(python -m pytest tests/test_dem_gpu.py tests/test_edge_cases_gpu.py tests/test_full_size_gpu.py tests/test_cloud_gpu.py tests/test_fuzz_gpu.py -x -q 2>&1 | tail -15) > gpurun_out/r06_suite_j.log
{
for rep in 1 2; do
tests/ab_env.sh "--bed fluidised --no-fluidised --no-parity" "SF_HIST_IN_PLACE=0 SF_BUILD_LDS=0" "SF_HIST_IN_PLACE=1"
tests/ab_env.sh "--bed fluidised --particles 100000 --no-fluidised --no-parity" "SF_HIST_IN_PLACE=0 SF_BUILD_LDS=0" "SF_HIST_IN_PLACE=1"
done
} > gpurun_out/r06_build_lds_ab2.txt 2>&1
SF_TRACE_MIN_NS=12000 tests/trace_rebuild.sh r06_c3j "--bed fluidised --particles 100000 --no-fluidised --no-parity" > gpurun_out/r06_trace_c3j.txt 2>&1
tests/trace_rebuild.sh r06_l1mj "--bed fluidised --no-fluidised --no-parity" > gpurun_out/r06_trace_l1mj.txt 2>&1
rm -rf gpurun_out/kt_r06_c3j gpurun_out/kt_r06_l1mj
python bench.py --gpus 1 --no-cpu-baseline > gpurun_out/r06_bench_j.json 2> gpurun_out/r06_bench_j.err
tail -3 gpurun_out/r06_suite_j.log; cat gpurun_out/r06_build_lds_ab2.txt gpurun_out/r06_trace_c3j.txt gpurun_out/r06_trace_l1mj.txt
python - <<'P'
import json
d = json.loads(open("gpurun_out/r06_bench_j.json").readline())
print("headline", d["value"], d["roofline"]["frac"], d["roofline"]["mean_kernel_us"])
f = d["fluidised_bed"]; print("fluidised", f["roofline_frac"], f["roofline_frac_whole_run"], f["neighbor_rebuild_ms"])
for k, v in d["configs"].items():
    print(k, {q: v.get(q) for q in ("value", "mean_kernel_us", "roofline_frac", "roofline_frac_whole_run", "error", "neighbor_rebuilds_in_run", "longest_row", "k_half")})
P
