#!/bin/bash
# GPU box: kernel trace of bench.py --gpus 4 --one-gpu (2x1x2 bricks, ranks sharing the GPU) with direct ghost writes:
# what stands between two sub-step kernels of a rank.  usage: tests/trace_direct.sh PARTICLES_TOTAL
n=${1:-504000}
root=$GRAFT_REPO_ROOT
/opt/rocm/bin/hipcc -shared -fPIC -O2 $root/tests/c_abi/standin_rccl.cpp -o /tmp/libstandin_rccl.so || exit 1
cd /tmp && export TMPDIR=/tmp
rm -rf $root/gpurun_out/kt_direct
SF_RCCL_LIB=/tmp/libstandin_rccl.so SF_HALO_DIRECT=1 timeout 900 rocprofv3 --kernel-trace --output-format csv -d $root/gpurun_out/kt_direct -o p -- \
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29744 \
  $root/bench.py --gpus 4 --one-gpu --particles $n --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-profile --no-parity --scaling strong > $root/gpurun_out/kt_direct.log 2>&1
python - <<PY
import csv, glob, collections, statistics
files = sorted(glob.glob("$root/gpurun_out/kt_direct/**/*kernel_trace.csv", recursive=True))
print("trace files:", len(files))
for f in files[:1]:
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    dur = collections.defaultdict(list)
    for r in rows:
        dur[r["Kernel_Name"].split("(")[0][:70]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1]))[:8]:
        print("%-72s n %5d  min %8.1f  median %8.1f  mean %8.1f us" % (k, len(v), min(v), statistics.median(v), sum(v) / len(v)))
    # period of the sub-step kernel and what lies between two of them
    sub = [r for r in rows if "k_substep" in r["Kernel_Name"] and int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) > 10000]
    gaps = []
    for a, b in zip(sub[:-1], sub[1:]):
        gaps.append((int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3)
    if gaps:
        print("between two sub-step kernels of this rank: min %.1f  p10 %.1f  median %.1f us  (n %d)" % (
            min(gaps), sorted(gaps)[len(gaps) // 10], statistics.median(gaps), len(gaps)))
PY
tail -2 $root/gpurun_out/kt_direct.log | cut -c1-300
