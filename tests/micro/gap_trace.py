"""development helper: start-to-start period and duration of k_substep launches from a rocprofv3 kernel trace
usage: python tests/micro/gap_trace.py gpurun_out/<dir>/p_kernel_trace.csv"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
sub = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if "k_substep" in r["Kernel_Name"]]
per = [(b[0] - a[0]) / 1e3 for a, b in zip(sub, sub[1:])]
dur = [(e - s) / 1e3 for s, e in sub]
per_s = sorted(per)
print("launches %d  dur median %.1f us  period median %.1f us  p10 %.1f  p90 %.1f" % (
    len(sub), sorted(dur)[len(dur) // 2], per_s[len(per) // 2], per_s[len(per) // 10], per_s[9 * len(per) // 10]))
# what runs between two consecutive sub-steps (median case): list kernels in a window of 6 launches mid-trace
mid = sub[len(sub) // 2][0]
win = [r for r in rows if mid <= int(r["Start_Timestamp"]) <= mid + 4 * per_s[len(per) // 2] * 1e3]
prev = None
for r in win:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("  +%8.1f us  gap %6.1f  dur %6.1f  %s" % ((s - mid) / 1e3, (s - prev) / 1e3 if prev else 0.0, (e - s) / 1e3, r["Kernel_Name"][:60]))
    prev = e
