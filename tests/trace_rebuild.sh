#!/bin/bash
# development helper (GPU box): the kernels of one neighbour rebuild (between two executed sub-step kernels) with their
# durations and the gaps before them.  usage: tests/trace_rebuild.sh TAG "bench args"
tag=$1; args=$2
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $root/gpurun_out/kt_$tag -o p -- \
  python $root/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-coupled --no-configs --no-kernel-profile $args > $root/gpurun_out/kt_$tag.log 2>&1
cd $root
python - "$tag" <<'P'
import csv, glob, sys
tag = sys.argv[1]
rows = []
for f in glob.glob("gpurun_out/kt_%s/*kernel_trace.csv" % tag):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
for f in glob.glob("gpurun_out/kt_%s/*memory_copy_trace.csv" % tag):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "MEMCPY " + r.get("Direction", "") + " " + r.get("Bytes", "")))
rows.sort()
import os
thr = int(os.environ.get("SF_TRACE_MIN_NS", "50000"))   # (an executed sub-step kernel: longer than an early exit)
big = [k for k, r in enumerate(rows) if "k_substep" in r[2] and r[1] - r[0] > thr]
# rebuilds: pairs of consecutive executed sub-steps with a k_build_neigh in between; take the last one
spans = [(a, b) for a, b in zip(big[:-1], big[1:]) if any("k_build_neigh" in rows[k][2] for k in range(a, b))]
a, b = spans[-1]
t0 = rows[a][1]
print("rebuild: %.1f us from the end of the last sub-step to the start of the next" % ((rows[b][0] - t0) / 1e3))
prev = t0
tot = 0
for r in rows[a + 1:b]:
    print("  gap %7.1f  run %7.1f  %s" % ((r[0] - prev) / 1e3, (r[1] - r[0]) / 1e3, r[2][:100]))
    tot += r[1] - r[0]
    prev = max(prev, r[1])
print("  gap %7.1f  (to the next sub-step)   kernels+copies busy %.1f us" % ((rows[b][0] - prev) / 1e3, tot / 1e3))
P
