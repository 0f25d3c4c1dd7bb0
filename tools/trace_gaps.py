#!/usr/bin/env python3
"""Where ONE stream's time goes between its kernels (rocprofv3 kernel trace of a one-cohort run, e.g. tools/gpu_busy.sh with
BUSY_ARGS="--slots 3 --combine 3"): for every kernel, the idle time on the stream BEFORE it (previous kernel's end -> its start),
averaged over the steady-state steps and listed in launch order of one step.  A 'step' starts at the first k_prover_pre after a
k_check_opened; the idle time between two steps (the caller's turn-around) is reported on its own.  Not product code.   usage: trace_gaps.py <kernel_trace.csv>"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("kosk::", "")) for r in rows)
steps, cur, closing = [], [], False
for s, e, n in ev:
    if closing and n.startswith("k_prover_pre"):
        steps.append(cur); cur = []; closing = False
    cur.append((s, e, n))
    if n.startswith("k_check_opened"): closing = True
steps = [st for st in steps[2:] if len(st) > 20]
if not steps: sys.exit("no steps found")
L = collections.Counter(len(st) for st in steps).most_common(1)[0][0]
steps = [st for st in steps if len(st) == L]
print("%d steps of %d launches; mean step %.0f us wall, %.0f us in kernels" % (len(steps), L, sum(st[-1][1] - st[0][0] for st in steps) / len(steps) / 1e3,
      sum(sum(e - s for s, e, _ in st) for st in steps) / len(steps) / 1e3))
tot_gap = 0
for i in range(L):
    name = steps[0][i][2]
    dur = sum(st[i][1] - st[i][0] for st in steps) / len(steps) / 1e3
    gap = sum((st[i][0] - st[i - 1][1]) if i else 0 for st in steps) / len(steps) / 1e3
    tot_gap += gap
    print("%3d %-36s gap before %7.1f us   kernel %7.1f us%s" % (i, name[:36], gap, dur, "   <<" if gap > 30 else ""))
print("sum of gaps inside a step: %.0f us; between a step's last kernel and the next step's first: %.0f us" % (tot_gap,
      sum(b[0][0] - a[-1][1] for a, b in zip(steps, steps[1:]) if b[0][0] - a[-1][1] < 5e6) / max(1, len(steps) - 1) / 1e3))
