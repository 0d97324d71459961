#!/bin/bash
# GPU box: the C5 variant (500 k polydisperse grains, gran/hertzFix/history + lubricate/poly + fix cohesive):
# kernel time per style mix on one box, kernel trace + stats of the full mix, PMC passes.  usage: tests/profile_c5.sh TAG [N]
tag=${1:-r05_c5}; n=${2:-500000}
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $out
for rep in 1 2; do
  for m in hertz cohesive lub all; do python tests/micro/poly_bench.py $n $m 2>/dev/null | tail -1; done
done | tee $out/${tag}_modes.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_$tag -o p -- python $GRAFT_REPO_ROOT/tests/micro/poly_bench.py $n all > $out/kt_$tag.log 2>&1
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VMEM_RD SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU" "TA_TA_BUSY_sum TA_BUSY_avr" "TD_TD_BUSY_sum TD_TC_STALL_sum" \
           "TCP_GATE_EN1_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS" \
           "SQ_BUSY_CU_CYCLES SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR" "GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES"; do
  i=$((i+1))
  timeout 240 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/pmc_${tag}_$i -o p -- python $GRAFT_REPO_ROOT/tests/micro/poly_bench.py $n all 3 > $out/pmc_${tag}_$i.log 2>&1
done
cd $GRAFT_REPO_ROOT
python - <<PY
import csv, collections, glob
rows = list(csv.DictReader(open("$out/kt_$tag/p_kernel_stats.csv")))
for r in rows[:14]:
    print("%-72s calls %5s avg %9.1f us total %9.1f us" % (r["Name"][:72], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e3))
for d in sorted(glob.glob("$out/pmc_${tag}_*/p_counter_collection.csv")):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(d)):
        if "k_substep" in r["Kernel_Name"] and int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) > 50000:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        print("%-32s mean %.6g  (n=%d)" % (k, sum(v) / len(v), len(v)))
PY
