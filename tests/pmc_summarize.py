"""Fold the rocprofv3 outputs of tests/pmc.sh (+ one --kernel-trace --stats run) into profiles/<tag>_pmc_summary.json.

usage: python tests/pmc_summarize.py TAG "description of the code state"
reads  gpurun_out/pmc_TAG_*/p_counter_collection.csv and gpurun_out/kt_TAG/p_kernel_trace.csv
"""
import collections
import csv
import glob
import json
import sys

tag, desc = sys.argv[1], sys.argv[2]
N = 1000188
CUS, CLK = 256, 2.4e9  # cycles are reported per CU; TCP_GATE_EN1 gives the active cycles directly

ctr = collections.defaultdict(list)
for path in sorted(glob.glob(f"gpurun_out/pmc_{tag}_*/p_counter_collection.csv")):
    for r in csv.DictReader(open(path)):
        if "k_substep" in r["Kernel_Name"] and int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) > 50000:
            ctr[r["Counter_Name"]].append(float(r["Counter_Value"]))
c = {k: {"mean": sum(v) / len(v), "launches": len(v)} for k, v in ctr.items()}
m = lambda k: c[k]["mean"]

durs = []
for path in glob.glob(f"gpurun_out/kt_{tag}/*kernel_trace.csv"):
    for r in csv.DictReader(open(path)):
        if "k_substep" in r["Kernel_Name"]:
            durs.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
ex = [d for d in durs if d > 50.0]

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sedifoam_amd.build import kernel_source_hash  # noqa: E402

out = {"kernel_source_sha256_16": kernel_source_hash(),
       "workload": "bench.py default (1,000,188-particle Hertz bed), kernel k_substep<2,false,false>, executed launches "
                   "only (duration > 50 us); code = " + desc,
       "particles": N, "counters": c}
if durs:
    out["kernel_trace"] = {"launches": len(durs), "executed": len(ex), "avg_us_all": sum(durs) / len(durs),
                           "avg_us_executed": sum(ex) / len(ex)}
rd, wr = m("FETCH_SIZE") * 1024.0, m("WRITE_SIZE") * 1024.0
# Calibration on known byte counts (tests/calibrate_traffic.sh -> gpurun_out/cal_<tag>.json, the read / read+write row
# streams of tests/micro/stream_bench.hip): FETCH_SIZE reports 0.500 of the bytes read, for 8, 16 and 32 bytes per lane
# alike (128-byte fabric requests tallied at 64), WRITE_SIZE 1.000, and TCC_MISS_sum counts 128-byte lines for reads
# AND writes (TCC_MISS_sum x 128 B = bytes read + written).  For k_substep the two agree: 2 x FETCH_SIZE + WRITE_SIZE =
# TCC_MISS_sum x 128 B within 1 % -- so the gathers' misses are 128-byte fetches tallied at 64 as well, and
#   read_calibrated = 2 x FETCH_SIZE.
cal = {}
try:
    rows = json.load(open(f"gpurun_out/cal_{tag}.json"))
    st = [r for r in rows if not r["write"]]
    cal = {"fetch_reported_over_true_streams": sum(r["fetch_reported_over_true"] for r in st) / len(st),
           "write_reported_over_true": sum(r["write_reported_over_true"] for r in rows if r["write"] and r["load_bytes_per_lane"] < 32)
                                       / max(1, len([r for r in rows if r["write"] and r["load_bytes_per_lane"] < 32])),
           "cases": rows}
except (OSError, KeyError, ZeroDivisionError):
    pass
rd_cal = 2.0 * rd
out["hbm_bytes_per_launch"] = {
    "read_raw": rd, "write": wr, "total_raw": rd + wr,
    "read_calibrated": rd_cal, "total_calibrated": rd_cal + wr,
    "tcc_miss_x_128B": m("TCC_MISS_sum") * 128.0,
    "algorithmic": 594.35 * N,
    "calibration": cal,
    "note": "FETCH_SIZE/WRITE_SIZE are KiB (separate --pmc passes). On known byte counts FETCH_SIZE reports exactly 1/2 "
            "of the bytes read (every load width), WRITE_SIZE all of the bytes written, TCC_MISS_sum 128-byte lines "
            "read + written: read_calibrated = 2 x read_raw, cross-checked by tcc_miss_x_128B ~ total_calibrated. "
            "bench.py reports total_calibrated as `traffic`."}
out["l2_hit_rate"] = m("TCC_HIT_sum") / (m("TCC_HIT_sum") + m("TCC_MISS_sum"))
wc = m("SQ_WAVE_CYCLES")
out["wave_cycle_split"] = {"waiting_any": m("SQ_WAIT_ANY") / wc, "issue_stall": m("SQ_WAIT_INST_ANY") / wc,
                           "issuing": m("SQ_ACTIVE_INST_ANY") / wc}
out["vector_l1_model"] = {
    "l2_read_requests_per_particle": m("TCP_TCC_READ_REQ_sum") / N,
    "l2_write_requests_per_particle": m("TCP_TCC_WRITE_REQ_sum") / N,
    "mean_l2_read_latency_cycles": m("TCP_TCC_READ_REQ_LATENCY_sum") / m("TCP_TCC_READ_REQ_sum"),
    "mean_outstanding_read_requests_per_cu": m("TCP_TCC_READ_REQ_LATENCY_sum") / m("TCP_GATE_EN1_sum"),
    "tcp_pending_stall_fraction": m("TCP_PENDING_STALL_CYCLES_sum") / m("TCP_GATE_EN1_sum"),
    "tcp_total_read_per_particle": m("TCP_TOTAL_READ_sum") / N,
    "vmem_read_instructions_per_wave": m("SQ_INSTS_VMEM_RD") / m("SQ_WAVES"),
    "valu_instructions_per_wave": m("SQ_INSTS_VALU") / m("SQ_WAVES"),
    "note": "requests in flight per CU = sum(latency)/active cycles: ~64, the vector L1's outstanding-miss capacity; "
            "the kernel runs at (64 requests x 256 CUs) / mean latency. See DESIGN.md section 5."}
json.dump(out, open(f"profiles/{tag}_pmc_summary.json", "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != "counters"}, indent=1))
