#!/usr/bin/env python3
"""GPU occupancy of a multi-slot bench run from a rocprofv3 kernel trace: fraction of wall time with >= 1 kernel in flight,
average number of kernels in flight, and per-kernel share of (kernel-duration) time.  Not product code.
usage: trace_overlap.py <kernel_trace.csv> [skip_fraction]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("kosk::", "")) for r in rows)
t_lo, t_hi = ev[0][0], max(e[1] for e in ev)
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
w0 = t_lo + (t_hi - t_lo) * skip      # steady-state window: drop the set-up part of the run
w1 = t_hi - (t_hi - t_lo) * 0.05
pts = []
dur = collections.Counter()
for s, e, n in ev:
    s, e = max(s, w0), min(e, w1)
    if e > s:
        pts.append((s, 1)); pts.append((e, -1)); dur[n] += e - s
pts.sort()
busy = 0; depth = 0; last = w0; area = 0
for t, d in pts:
    if depth > 0: busy += t - last
    area += depth * (t - last)
    depth += d; last = t
W = w1 - w0
print("window %.1f ms: >=1 kernel in flight %.1f %% of the time, mean kernels in flight %.2f" % (W / 1e6, 100.0 * busy / W, area / W))
tot = sum(dur.values())
for n, v in dur.most_common(16):
    print("  %-40s %5.1f %% of kernel time, %.2f in flight on average" % (n[:40], 100.0 * v / tot, v / W))
