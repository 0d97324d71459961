#!/bin/bash
# development helper (GPU box): rebuild-path kernels of differently built libraries on the same box.
# usage: tests/ab_rebuild.sh "bench args" name1[:ENV=VAL,ENV2=VAL2] name2 ...   ("default" = the shipped build)
args=$1; shift
root=$GRAFT_REPO_ROOT
for spec in "$@"; do
  v=${spec%%:*}; envs=""; [ "$spec" != "$v" ] && envs=$(echo ${spec#*:} | tr ',' ' ')
  p=""; [ "$v" != "default" ] && p=$root/sedifoam_amd/libsedifoam_amd_$v.so
  export SF_LIB_PATH=$p
  cd /tmp && export TMPDIR=/tmp
  rm -rf $root/gpurun_out/abr_$v
  timeout 300 env $envs rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/abr_$v -o p -- \
    python $root/bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-coupled --no-configs --no-fluidised --no-parity $args > $root/gpurun_out/abr_$v.log 2>&1
  cd $root
  python - "$v" "$spec" <<'P'
import csv, sys, json
v = sys.argv[1]
rows = list(csv.DictReader(open("gpurun_out/abr_%s/p_kernel_stats.csv" % v)))
want = ("k_build_neigh", "k_partner_tags", "k_permute_all", "k_gather4", "k_gather_rows", "k_copy_rows", "k_back_slots", "k_substep")
out = []
for r in rows:
    for w in want:
        if w in r["Name"]:
            out.append("%s %sx%.1f" % (w, r["Calls"], float(r["AverageNs"]) / 1e3))
line = [l for l in open("gpurun_out/abr_%s.log" % v) if l.startswith("{")]
val = json.loads(line[-1])["value"] if line else float("nan")
print("%-22s value %.3e | %s" % (sys.argv[2], val, " | ".join(out)))
P
done
