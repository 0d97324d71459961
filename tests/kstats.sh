#!/bin/bash
# per-kernel averages of a short DEM-only bench under rocprofv3 (GPU box): tests/kstats.sh TAG [bench args]
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/ks_$tag -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-coupled --no-configs "$@" > $GRAFT_REPO_ROOT/gpurun_out/ks_$tag.log 2>&1
python - <<PY
import csv
rows=list(csv.DictReader(open("$GRAFT_REPO_ROOT/gpurun_out/ks_$tag/p_kernel_stats.csv")))
for r in rows[:22]:
    print("%-72s calls %5s avg %9.1f us total %9.1f us" % (r["Name"][:72], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e3))
PY
