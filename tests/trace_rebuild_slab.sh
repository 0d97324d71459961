#!/bin/bash
# development helper (GPU box): timeline of one DECOMPOSED rebuild (self-communication over RCCL)
tag=$1; n=${2:-125000}
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
SF_HALO_SELF_COMM=1 timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $root/gpurun_out/kt_$tag -o p -- \
  python $root/bench.py --slab-driver --particles $n --steps 8 --warmup 1 --no-cpu-baseline --no-coupled --no-configs --no-kernel-profile > $root/gpurun_out/kt_$tag.log 2>&1
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
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "MEMCPY " + r.get("Direction", "")))
rows.sort()
thr = 5000 if "125" in tag or True else 50000
big = [k for k, r in enumerate(rows) if "k_substep" in r[2] and r[1] - r[0] > 15000]
spans = [(a, b) for a, b in zip(big[:-1], big[1:]) if any("k_build_neigh" in rows[k][2] for k in range(a, b))]
a, b = spans[-1]
t0 = rows[a][1]
print("rebuild: %.1f us from the end of the last executed sub-step to the start of the next" % ((rows[b][0] - t0) / 1e3))
prev = t0; tot = 0; gaps = 0
for r in rows[a + 1:b]:
    g = (r[0] - prev) / 1e3
    if g > 8: print("  gap %7.1f" % g)
    gaps += max(g, 0)
    print("           run %7.1f  %s" % ((r[1] - r[0]) / 1e3, r[2][:90]))
    tot += r[1] - r[0]
    prev = max(prev, r[1])
print("kernels+copies busy %.1f us, gaps %.1f us" % (tot / 1e3, gaps))
P
