#!/bin/bash
# development helper (GPU box): kernel trace of the halo loop with one rank exchanging with its own periodic images
# over RCCL (SF_HALO_SELF_COMM=1): what the forward exchange costs per sub-step next to k_substep.
# usage: [BENCH_EXTRA="--decomposition bricks"] [SF_HALO_DIRECT=1] tests/trace_selfcomm.sh TAG PARTICLES
#   (--decomposition bricks: ONE brick whose periodic dimensions are external -- the brick driver's exchange, over RCCL or,
#    with SF_HALO_DIRECT=1, by direct ghost writes into its own receive areas)
tag=$1; n=${2:-125000}
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
SF_HALO_SELF_COMM=1 SF_HALO_DIRECT_TIMEOUT=10 timeout -k 20 300 rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/kt_$tag -o p -- \
  python $root/bench.py --slab-driver --particles $n --steps 4 --warmup 1 --no-cpu-baseline --no-coupled --no-configs --no-kernel-profile $BENCH_EXTRA > $root/gpurun_out/kt_$tag.log 2>&1
cd $root
tail -1 gpurun_out/kt_$tag.log | cut -c1-300
f=$(ls gpurun_out/kt_$tag/*/*kernel_stats.csv 2>/dev/null | head -1); [ -z "$f" ] && f=$(ls gpurun_out/kt_$tag/*kernel_stats.csv | head -1)
head -12 $f | cut -c1-200
python - "$tag" <<'P'
import csv, glob, sys
tag = sys.argv[1]
f = (glob.glob("gpurun_out/kt_%s/*/*kernel_trace.csv" % tag) + glob.glob("gpurun_out/kt_%s/*kernel_trace.csv" % tag))[0]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))))
# the steady loop: find consecutive k_substep launches and print the timeline of one period in the middle
idx = [k for k, r in enumerate(rows) if r[2].startswith("void sf::k_substep") and r[1] - r[0] > 5000]
mid = idx[len(idx) // 2]
nxt = [k for k in idx if k > mid][0]
t0 = rows[mid][0]
for r in rows[mid:nxt + 1]:
    print("%9.2f us  +%7.2f us  %s" % ((r[0] - t0) / 1e3, (r[1] - r[0]) / 1e3, r[2][:90]))
per = [rows[b][0] - rows[a][0] for a, b in zip(idx[:-1], idx[1:]) if b - a < 8]
ker = [rows[a][1] - rows[a][0] for a in idx]
per.sort(); ker.sort()
print("median period %.2f us, median k_substep %.2f us, exchange+gaps %.2f us" % (per[len(per) // 2] / 1e3, ker[len(ker) // 2] / 1e3, (per[len(per) // 2] - ker[len(ker) // 2]) / 1e3))
P
