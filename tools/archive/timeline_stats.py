#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV of the multi-stream bench: fraction of wall time with >= 1 kernel resident,
time-weighted mean number of concurrent kernels, and per-kernel share of the busy union (attributed by equal split)."""
import csv, glob, sys, collections
paths = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
ev = []
rows = []
for p in paths:
    for r in csv.DictReader(open(p)):
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        rows.append((s, e, r["Kernel_Name"].split("(")[0].replace("void ", "").replace("gl355::", "")))
rows.sort()
# the steady multi-stream part = the 0.6-second window holding the most kernel launches
import bisect
starts = [r[0] for r in rows]
W = int(0.6e9)
best, lo = -1, rows[0][0]
for i in range(0, len(rows), 50):
    j = bisect.bisect_left(starts, starts[i] + W)
    if j - i > best:
        best, lo = j - i, starts[i]
hi = lo + W
for s, e, n in rows:
    if e <= lo or s >= hi:
        continue
    ev.append((max(s, lo), 1, n)); ev.append((min(e, hi), -1, n))
ev.sort(key=lambda x: (x[0], x[1]))
active = collections.Counter()
cur = 0; last = lo; busy = 0.0; conc = 0.0; share = collections.Counter(); hist = collections.Counter()
for t, d, n in ev:
    dt = t - last
    if dt > 0:
        if cur > 0:
            busy += dt; conc += dt * cur
            for k, c in active.items():
                if c: share[k] += dt * c / cur
        hist[min(cur, 12)] += dt
    last = t
    cur += d; active[n] += d
wall = hi - lo
print("window %.1f ms: GPU has >=1 kernel resident %.1f %% of the time; mean concurrency while busy %.2f" % (wall / 1e6, 100 * busy / wall, conc / max(busy, 1)))
print("time at concurrency k: " + ", ".join("%d: %.1f%%" % (k, 100 * v / wall) for k, v in sorted(hist.items())))
for k, v in share.most_common(10):
    print("  %-34s %5.1f %% of busy time (equal-split attribution)" % (k[:34], 100 * v / busy))
