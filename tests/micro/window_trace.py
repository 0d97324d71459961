"""development helper: kernels between two consecutive k_substep launches in the middle of a rocprofv3 kernel trace
usage: python tests/micro/window_trace.py <p_kernel_trace.csv> [n_substeps=3]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
nsub = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
subs = [i for i, r in enumerate(rows) if "k_substep" in r["Kernel_Name"] and int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) > 20000]
a = subs[len(subs) // 2]
b = subs[len(subs) // 2 + nsub]
t0 = int(rows[a]["Start_Timestamp"]); prev = None
for r in rows[a:b + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("+%8.1f us  gap %6.1f  dur %6.1f  %s" % ((s - t0) / 1e3, (s - prev) / 1e3 if prev else 0.0, (e - s) / 1e3, r["Kernel_Name"][:70]))
    prev = e
