#!/usr/bin/env python3
"""GPU occupancy of a multi-slot bench run from a rocprofv3 kernel trace (not product code).

usage: tools/trace_overlap.py <kernel_trace.csv>
Prints, over the steady-state middle half of the trace: wall time, union of kernel intervals
(GPU non-idle), sum of kernel durations (avg concurrency = sum / union) and per-kernel totals.
"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
t0, t1 = ev[0][0], ev[-1][1]
lo, hi = t0 + (t1 - t0) // 4, t0 + 3 * (t1 - t0) // 4
ev = [e for e in ev if e[0] >= lo and e[1] <= hi]
union, cur_s, cur_e, tot = 0, None, None, 0
per = collections.Counter()
for s, e, n in ev:
    tot += e - s
    per[n.split("(")[0].replace("void ", "").replace("kosk::", "")[:44]] += e - s
    if cur_e is None or s > cur_e:
        if cur_e is not None: union += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
union += cur_e - cur_s
wall = hi - lo
steps = sum(1 for e in ev if "k_prover_pre" in e[2])
print("window %.1f ms, %d prove steps -> %.1f us/step wall" % (wall / 1e6, steps, wall / 1e3 / steps))
print("GPU non-idle %.1f %% of wall; sum of kernel durations %.0f us/step; avg concurrency while busy %.2f" % (100.0 * union / wall, tot / 1e3 / steps, tot / union))
for n, d in per.most_common(int(sys.argv[2]) if len(sys.argv) > 2 else 14):
    print("  %-44s %7.1f us/step" % (n, d / 1e3 / steps))
