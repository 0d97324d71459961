#!/bin/bash
# round 6 verification: whole GPU suite on the split sources, then the profile round (calibration, trace + stats, PMC passes,
# bench line) and three counter passes over k_build_neigh of the loose 1 M bed
cd $GRAFT_REPO_ROOT
(python -m pytest tests -m gpu -x -q 2>&1 | tail -8) > gpurun_out/r06_suite_s.log
tests/profile_round.sh r06 > gpurun_out/r06_profile_round.log 2>&1
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVES" "TCP_GATE_EN1_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TA_TA_BUSY_sum TD_TD_BUSY_sum GRBM_GUI_ACTIVE FETCH_SIZE WRITE_SIZE"; do
  i=$((i+1))
  timeout 240 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmcb_r06_$i -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-fluidised --no-coupled --no-configs --no-parity --bed fluidised > $GRAFT_REPO_ROOT/gpurun_out/pmcb_r06_$i.log 2>&1
done
cd $GRAFT_REPO_ROOT
python - > gpurun_out/r06_build_pmc.txt <<'PY'
import csv, collections, glob
for d in sorted(glob.glob("gpurun_out/pmcb_r06_*/p_counter_collection.csv")):
    agg = collections.defaultdict(list); dur = []
    for r in csv.DictReader(open(d)):
        if "k_build_neigh" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
            dur.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for k, v in agg.items():
        v = v[1:] if len(v) > 1 else v   # (the first build of a run is the packed-order one)
        print("%-32s mean %.6g  (n=%d)" % (k, sum(v) / len(v), len(v)))
    if dur: print("   kernel us mean %.1f" % (sum(dur[1:]) / max(1, len(dur) - 1) / 1e3 / max(1, len(agg))))
PY
rm -rf gpurun_out/pmcb_r06_*
tail -3 gpurun_out/r06_suite_s.log; tail -25 gpurun_out/r06_profile_round.log; cat gpurun_out/r06_build_pmc.txt; du -sh gpurun_out
