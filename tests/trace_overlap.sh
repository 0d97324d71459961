#!/bin/bash
# development helper (GPU box): kernel timeline of two sub-steps of the overlapped halo loop (self-communication)
tag=$1; n=${2:-125000}
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
SF_HALO_SELF_COMM=1 SF_HALO_OVERLAP=1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $root/gpurun_out/kt_$tag -o p -- \
  python $root/bench.py --slab-driver --particles $n --steps 4 --warmup 1 --no-cpu-baseline --no-coupled --no-configs --no-kernel-profile > $root/gpurun_out/kt_$tag.log 2>&1
cd $root
python - "$tag" <<'P'
import csv, glob, sys
tag = sys.argv[1]
f = glob.glob("gpurun_out/kt_%s/*kernel_trace.csv" % tag)[0]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")) for r in csv.DictReader(open(f))))
idx = [k for k, r in enumerate(rows) if "k_substep" in r[2]]
mid = idx[len(idx) // 2]
t0 = rows[mid][0]
for r in rows[mid:mid + 14]:
    print("%9.2f us  +%7.2f us  q%s  %s" % ((r[0] - t0) / 1e3, (r[1] - r[0]) / 1e3, r[3], r[2][:80]))
P
