#!/bin/bash
# development helper (GPU box): PMC passes over the list-build kernel of a loose-bed run. usage: tests/pmc_build.sh "CTR1 CTR2" "CTR3" ...
cd /tmp && export TMPDIR=/tmp
i=0
for set in "$@"; do
  i=$((i+1))
  timeout 240 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmcb_$i -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-fluidised --no-coupled --no-configs --no-parity --bed fluidised > $GRAFT_REPO_ROOT/gpurun_out/pmcb_$i.log 2>&1
done
python - <<PY
import csv, collections, glob
for d in sorted(glob.glob("$GRAFT_REPO_ROOT/gpurun_out/pmcb_*/p_counter_collection.csv")):
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(d)):
        if "k_build_neigh" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in agg.items(): print("%-28s mean %.5g  (n=%d)"%(k,sum(v[1:])/max(1,len(v)-1),len(v)))
PY
