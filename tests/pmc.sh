#!/bin/bash
# rocprofv3 PMC passes over the default bench (GPU box). usage: [PMC_BENCH_ARGS="--jitter .."] tests/pmc.sh TAG "CTR1 CTR2" "CTR3" ...
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
i=0
for set in "$@"; do
  i=$((i+1))
  timeout 240 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_${tag}_$i -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-fluidised --no-coupled --no-configs $PMC_BENCH_ARGS > $GRAFT_REPO_ROOT/gpurun_out/pmc_${tag}_$i.log 2>&1
done
python - <<PY
import csv, collections, glob
for d in sorted(glob.glob("$GRAFT_REPO_ROOT/gpurun_out/pmc_${tag}_*/p_counter_collection.csv")):
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(d)):
        if "k_substep" in r["Kernel_Name"] and int(r["End_Timestamp"])-int(r["Start_Timestamp"])>50000:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in agg.items(): print("%-28s mean %.5g  (n=%d)"%(k,sum(v)/len(v),len(v)))
PY
