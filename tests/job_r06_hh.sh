#!/bin/bash
# the round's last library: whole GPU suite, smoke, the driver's bench command
cd $GRAFT_REPO_ROOT
(python -m pytest tests -m gpu -x -q 2>&1 | tail -3) > gpurun_out/r06_suite_hh.log
(python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2) > gpurun_out/r06_smoke_hh.log
python bench.py --gpus 1 > gpurun_out/r06_bench_hh.json 2> gpurun_out/r06_bench_hh.err
tail -n 2 gpurun_out/r06_suite_hh.log; cat gpurun_out/r06_smoke_hh.log
python - <<'P'
import json
b = json.loads(open("gpurun_out/r06_bench_hh.json").readline())
f = b["fluidised_bed"]; c = b["configs"]; r = b["roofline"]
print("%.3e" % b["value"], round(r["frac"], 4), r["mean_kernel_us"], r.get("traffic"), "fluid", round(f["roofline_frac_whole_run"], 4), "C3", round(c["C3"]["roofline_frac_whole_run"], 4),
      "C5", round(c["C5"]["roofline_frac"], 4), "C2", "%.3e" % c["C2"]["value"], "C5w", round(c["C5_wide"]["roofline_frac"], 4))
P
