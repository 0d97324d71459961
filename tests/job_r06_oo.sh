#!/bin/bash
cd $GRAFT_REPO_ROOT
(python -m pytest tests/test_dem_gpu.py tests/test_full_size_gpu.py tests/test_cloud_gpu.py -x -q 2>&1 | tail -2)
for rep in 1 2; do
tests/ab_env.sh "--bed fluidised --particles 100000 --no-fluidised --no-parity" "SF_LPA=1" "SF_X=0"
tests/ab_env.sh "--bed fluidised --particles 300000 --no-fluidised --no-parity" "SF_LPA=1" "SF_X=0"
done
python bench.py --gpus 1 > gpurun_out/r06_bench_oo.json 2>/dev/null
python - <<'P'
import json
b = json.loads(open("gpurun_out/r06_bench_oo.json").readline())
f = b["fluidised_bed"]; c = b["configs"]; r = b["roofline"]
print("%.3e" % b["value"], round(r["frac"], 4), r["mean_kernel_us"], "fluid", round(f["roofline_frac_whole_run"], 4), "C3", round(c["C3"]["roofline_frac_whole_run"], 4), round(c["C3"]["mean_kernel_us"], 1),
      "C5", round(c["C5"]["roofline_frac"], 4), "C2", "%.3e" % c["C2"]["value"], "C5w", round(c["C5_wide"]["roofline_frac"], 4))
P
