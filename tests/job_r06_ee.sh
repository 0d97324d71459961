#!/bin/bash
# the driver's bench command with the longer side configurations (twice on this box: spread)
cd $GRAFT_REPO_ROOT
for k in 1 2; do SECONDS=0; python bench.py --gpus 1 > gpurun_out/r06_bench_ee$k.json 2> gpurun_out/r06_bench_ee$k.err; echo "wall $SECONDS s"; done
python - <<'P'
import json
for k in (1, 2):
    b = json.loads(open("gpurun_out/r06_bench_ee%d.json" % k).readline())
    f = b["fluidised_bed"]; c = b["configs"]
    print(k, "%.3e" % b["value"], round(b["roofline"]["frac"], 4), "fluid", round(f["roofline_frac_whole_run"], 4), f["neighbor_rebuilds_in_run"], f["steps"],
          "C3", round(c["C3"]["roofline_frac_whole_run"], 4), c["C3"]["steps"], "C5", round(c["C5"]["roofline_frac"], 4), "C2", "%.3e" % c["C2"]["value"], "C5w", round(c["C5_wide"]["roofline_frac"], 4))
P
