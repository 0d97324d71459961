"""Development helper: the fill / drain timeline of ONE k_substep launch from the per-workgroup stamps of a
-DSF_EXP_STAMP=1 variant library (SF_STAMP_FILE).  Prints, per XCD and for the chip: first start, last end, resident
workgroups over time, the lifetime of a workgroup by dispatch round, and what the ramp and the tail cost against a
launch that kept the plateau occupancy from its first to its last microsecond.
usage: python tests/micro/stamp_timeline.py stamps.bin [bin_us]"""
import sys

import numpy as np

raw = np.fromfile(sys.argv[1], dtype=np.uint64)
grid = int(raw[0])
a = raw[1:1 + 4 * grid].reshape(-1, 4)
phase = raw[1 + 4 * grid:1 + 36 * grid].reshape(-1, 32)
binw = float(sys.argv[2]) if len(sys.argv) > 2 else 2.0
ok = a[:, 3] > a[:, 2] + 200   # (workgroups that did work: more than 2 us)
print("workgroups %d, recorded %d" % (len(a), ok.sum()))
a = a[ok]
phase = phase[ok]
xcc = (a[:, 0] & 0xF).astype(int)
hw = a[:, 1].astype(np.int64)
cu = (hw >> 8) & 0xF
se = (hw >> 13) & 0x7
t0 = (a[:, 2] - a[:, 2].min()).astype(np.float64) / 100.0   # us (100 MHz)
t1 = (a[:, 3] - a[:, 2].min()).astype(np.float64) / 100.0
life = t1 - t0
T = t1.max()
print("launch span %.1f us (first start -> last end); workgroup lifetime mean %.1f  p10 %.1f  p50 %.1f  p90 %.1f  max %.1f us"
      % (T, life.mean(), *np.percentile(life, [10, 50, 90]), life.max()))
nb = int(np.ceil(T / binw))
edges = np.arange(nb + 1) * binw
res = np.zeros(nb)
for k in range(nb):   # resident workgroups at the centre of each bin
    c = edges[k] + 0.5 * binw
    res[k] = np.count_nonzero((t0 <= c) & (t1 > c))
plateau = np.percentile(res, 75)
print("resident workgroups: plateau (p75 of bins) %.0f" % plateau)
print("  t[us]  resident   (per XCD)")
for k in range(nb):
    c = edges[k] + 0.5 * binw
    if k < 8 or k >= nb - 24 or k % 10 == 0:
        per = [np.count_nonzero((t0 <= c) & (t1 > c) & (xcc == x)) for x in range(8)]
        print("  %6.1f  %6d   %s" % (c, res[k], " ".join("%4d" % p for p in per)))
# work-equivalent loss: integral of (plateau - resident)+ over the launch / plateau
loss = np.sum(np.clip(plateau - res, 0, None)) * binw / plateau
head = np.sum(np.clip(plateau - res[: nb // 2], 0, None)) * binw / plateau
print("occupancy deficit against the plateau: %.1f us of the %.1f us launch (first half %.1f, second half %.1f)"
      % (loss, T, head, loss - head))
print("per XCD: workgroups, first start, last end, mean lifetime")
for x in range(8):
    m = xcc == x
    if m.any():
        print("  xcc %d: %6d  %7.1f  %7.1f  %6.1f" % (x, m.sum(), t0[m].min(), t1[m].max(), life[m].mean()))
# lifetime by start time (dispatch rounds)
order = np.argsort(t0)
q = np.array_split(order, 10)
print("lifetime by dispatch decile (start time): " + " ".join("%.1f" % life[i].mean() for i in q))
print("start time by dispatch decile:            " + " ".join("%.1f" % t0[i].mean() for i in q))

# where a wave's life goes (SF_EXP_PHASE): marks 0 entry, 1 loop start, 2 + s slot s, 27 loop end, 28 fixes done, 29 before
# the record stores, 30 end
if phase.any():
    base = a[:, 2].astype(np.float64)
    ph = np.where(phase > 0, (phase.astype(np.float64) - base[:, None]) / 100.0, np.nan)
    mid = (t0 > 60) & (t0 < 140)          # steady state
    lastw = t0 > np.percentile(t0, 92)    # the drain
    for name, m in (("steady-state waves", mid), ("last-dispatched 8 %", lastw)):
        p = ph[m]
        print("%s (%d): mean clock since wave start [us]" % (name, m.sum()))
        print("   entry %.2f  loop start %.2f  loop end %.2f  fixes %.2f  before stores %.2f  end %.2f   (lifetime %.2f)"
              % (np.nanmean(p[:, 0]), np.nanmean(p[:, 1]), np.nanmean(p[:, 27]), np.nanmean(p[:, 28]), np.nanmean(p[:, 29]),
                 np.nanmean(p[:, 30]), life[m].mean()))
        d = np.diff(np.concatenate([p[:, 2:26], p[:, 27:28]], axis=1), axis=1)
        print("   slot durations: " + " ".join("%.2f" % v for v in np.nanmean(d[:, :13], axis=0)))
