"""development helper: GPU time between the last full sub-step kernel before a rebuild and the first one after it,
from a rocprofv3 --kernel-trace csv.  usage: python tests/micro/rebuild_window.py <p_kernel_trace.csv>"""
import collections
import csv
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
idx = [i for i, r in enumerate(rows) if "k_build_neigh" in r[2]]
for i in idx[1:]:
    a = i
    while not ("k_substep" in rows[a][2] and rows[a][1] - rows[a][0] > 50000):
        a -= 1
    b = i
    while b < len(rows) and not ("k_substep" in rows[b][2] and rows[b][1] - rows[b][0] > 50000):
        b += 1
    if b >= len(rows):
        continue
    t0 = rows[a][1]
    agg = collections.OrderedDict()
    for s, e, k in rows[a + 1:b]:
        key = k[:48]
        agg.setdefault(key, [0, 0])
        agg[key][0] += 1
        agg[key][1] += e - s
    busy = sum(v[1] for v in agg.values())
    print("rebuild window %.1f us, busy %.1f us, %d kernels" % ((rows[b][0] - t0) / 1e3, busy / 1e3, b - a - 1))
    for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:8]:
        print("    %-48s x%-3d %8.1f us" % (k, c, t / 1e3))
