#!/bin/bash
# development helper (GPU box): kernel time + FETCH_SIZE/WRITE_SIZE + L2 hit/miss of the sub-step kernel for a list of env settings
# usage: [AB_FETCH_SETS="TCC_HIT_sum,TCC_MISS_sum FETCH_SIZE WRITE_SIZE"] tests/ab_fetch.sh "ENV=.." "ENV2=.." ...
# (one --pmc pass per set; FETCH_SIZE and WRITE_SIZE must be passes of their own: together the run hangs)
root=$GRAFT_REPO_ROOT
i=0
for v in "$@"; do
  i=$((i+1))
  echo "== $v"
  ( cd $root; env $v python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-coupled --no-configs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('   kernel_us %.1f frac %.3f value %.3e'%(d['roofline']['mean_kernel_us'],d['roofline']['frac'],d['value']))" )
  for set in ${AB_FETCH_SETS:-TCC_HIT_sum,TCC_MISS_sum}; do
    ( cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/abf; env $v timeout 100 rocprofv3 --pmc ${set//,/ } --kernel-trace --output-format csv -d /tmp/abf -o p -- python $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-coupled --no-configs > /tmp/abf.log 2>&1
      python - <<'PY'
import csv, collections
agg=collections.defaultdict(list)
for r in csv.DictReader(open("/tmp/abf/p_counter_collection.csv")):
    if "k_substep" in r["Kernel_Name"] and int(r["End_Timestamp"])-int(r["Start_Timestamp"])>50000:
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("   "+"  ".join("%s %.4g"%(k,sum(v)/len(v)) for k,v in agg.items()))
PY
    )
  done
done
