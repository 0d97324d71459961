#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of rocprofv3 against known byte counts (GPU box): the row-stream micro-benchmark under the
# two PMC passes; prints reported / true bytes per kernel.  usage: tests/calibrate_traffic.sh TAG
tag=${1:-cal}
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE TCC_MISS_sum; do
  timeout 240 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/cal_${tag}_$c -o p -- $GRAFT_REPO_ROOT/tests/micro/stream_bench > $GRAFT_REPO_ROOT/gpurun_out/cal_${tag}_$c.log 2>&1
done
python - <<PY
import csv, collections, re, json
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE", "TCC_MISS_sum"):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open("$GRAFT_REPO_ROOT/gpurun_out/cal_${tag}_%s/p_counter_collection.csv" % c)):
        if r["Counter_Name"] == c:
            agg[r["Kernel_Name"]].append(float(r["Counter_Value"]) * (64.0 if c == "TCC_MISS_sum" else 1024.0))
    for k, v in agg.items():
        m = re.search(r"k_rows<(\d+), (\d+), (true|false), (true|false)>", k)
        if not m:
            continue
        W, rows, write, nt = int(m.group(1)), int(m.group(2)), m.group(3) == "true", m.group(4) == "true"
        grid_threads = None
        out.setdefault((W, rows, write, nt), {})[c] = sum(v) / len(v)
n = 1 << 20
res = []
for (W, rows, write, nt), d in sorted(out.items()):
    # the 12-row "x4 wide" / 24-row "x2 wide" cases stream W*n elements per row
    elems = n * (W if (rows, W) in ((12, 4), (24, 2)) else 1)
    true_rd = rows * elems * 8.0
    line = {"load_bytes_per_lane": 8 * W, "rows": rows, "write": write, "nontemporal": nt,
            "fetch_reported_over_true": d.get("FETCH_SIZE", 0.0) / true_rd,
            "tcc_miss_x64B_over_true_read_plus_write": d.get("TCC_MISS_sum", 0.0) / (true_rd * (2 if write else 1))}
    if write:
        line["write_reported_over_true"] = d.get("WRITE_SIZE", 0.0) / true_rd
    res.append(line)
    print(line)
json.dump(res, open("$GRAFT_REPO_ROOT/gpurun_out/cal_${tag}.json", "w"), indent=1)
PY
