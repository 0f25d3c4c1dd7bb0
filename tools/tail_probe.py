#!/usr/bin/env python3
"""Where do the rare long steps of a long run come from?  bench.py's default arrangement (12 callers, cohorts of four, 46 Kyber-768 proofs per
call) for N steps per caller; every caller keeps (start, keygen-call end, verify-call end) of every step and the library's phase clocks after
each call (only the handle that LED the merged run has new clocks).  Steps above 1.6 x the median are printed with the phases of the run
they were in, beside the median of each phase.  Not product code.   usage: tail_probe.py [steps per caller] [slots] [combine]"""
import os, sys, time, threading
sys.path.insert(0, ".")
import torch
import bench
from mpcith_kyber_kosk_amd import api
N = int(sys.argv[1]) if len(sys.argv) > 1 else 150
S = int(sys.argv[2]) if len(sys.argv) > 2 else 12
CMB = int(sys.argv[3]) if len(sys.argv) > 3 else 4
PH = ["host_pre", "gpu_commit", "fs_alpha", "gpu_relation", "fs_open", "gpu_assemble", "d2h", "p1_issue", "p2_issue", "p3_issue",
      "v1_issue", "v1_wait", "v_fs_alpha", "v2_issue", "v2_wait", "v_fs_open"]
slots = [bench.Slot(api, torch, 3, 46, 0, si * 4 * 46, 4, combine=CMB) for si in range(S)]
for sl in slots:
    sl.step(torch, 0)
rec = [[] for _ in range(S)]
bar = threading.Barrier(S)

def worker(si):
    sl = slots[si]
    c = sl.c
    kg, vf, h, pk, sk, ptrs, ones = sl._fast
    bar.wait()
    for i in range(N + 20):
        t0 = time.perf_counter()
        assert kg(h, 46, ptrs[i % 4], sl.stride, pk, sk) == 0
        t1 = time.perf_counter()
        p1 = c.phase_seconds()
        t1b = time.perf_counter()
        assert vf(h, 46, None, sl._ok) == 0
        t2 = time.perf_counter()
        p2 = c.phase_seconds()
        rec[si].append((t0, t1, t1b, t2, p1, p2))
import gc
gc_stats0 = gc.get_stats()
if os.environ.get("TAIL_GC", "1") == "0":  # is the process-wide stall the interpreter's cyclic collector (a full collection holds the lock every caller needs)?
    gc.collect(); gc.disable()
elif os.environ.get("TAIL_GC") == "freeze":
    gc.collect(); gc.freeze()
ths = [threading.Thread(target=worker, args=(si,)) for si in range(S)]
t_s = time.perf_counter()
for t in ths: t.start()
for t in ths: t.join()
t_e = time.perf_counter()
print("%d callers x %d steps, cohorts of %d: %.0f proofs/s overall" % (S, N + 20, CMB, S * (N + 20) * 46 / (t_e - t_s)))
print("run started at CLOCK_MONOTONIC %.3f ms (the t= below are relative to it)" % (t_s * 1e3))
print("TAIL_GC=%s; collections during the run per generation: %s" % (os.environ.get("TAIL_GC", "1"), [b["collections"] - a["collections"] for a, b in zip(gc_stats0, gc.get_stats())]))
lat = sorted((r[3] - r[0]) for si in range(S) for r in rec[si][20:])
med = lat[len(lat) // 2]
print("latency ms: median %.2f p90 %.2f p99 %.2f max %.2f" % (med * 1e3, lat[int(len(lat) * .9)] * 1e3, lat[int(len(lat) * .99)] * 1e3, lat[-1] * 1e3))
def med_of(vals):
    v = sorted(vals); return v[len(v) // 2]
# median of each phase over the calls whose clocks changed (= led a run)
led_kg, led_vf = [], []
for si in range(S):
    prev = None
    for r in rec[si][20:]:
        if prev is not None:
            if r[4][:10] != prev[5][:10]: led_kg.append(r[4])
            if r[5][10:] != r[4][10:]: led_vf.append(r[5])
        prev = r
mp = [med_of([p[i] for p in (led_kg if i < 10 else led_vf)]) * 1e3 if (led_kg and led_vf) else 0 for i in range(16)]
print("median phases (ms) of the runs' leaders: " + " ".join("%s %.3f" % (PH[i], mp[i]) for i in range(16)))
print("median keygen call %.2f verify call %.2f gap between them %.3f ms" % (med_of([r[1] - r[0] for si in range(S) for r in rec[si][20:]]) * 1e3,
      med_of([r[3] - r[2] for si in range(S) for r in rec[si][20:]]) * 1e3, med_of([r[2] - r[1] for si in range(S) for r in rec[si][20:]]) * 1e3))
out = []
for si in range(S):
    for i, r in enumerate(rec[si]):
        if i >= 20 and r[3] - r[0] > 1.6 * med:
            out.append((r[0], si, i, r))
out.sort()
print("%d steps above 1.6 x median:" % len(out))
for t0, si, i, r in out[:80]:
    prev = rec[si][i - 1]
    ch_kg = r[4][:10] != prev[5][:10]
    ch_vf = r[5][10:] != r[4][10:]
    line = "t=%8.2f ms slot %2d (cohort %d) step %3d: keygen call %.2f verify call %.2f between %.3f since previous step's end %.3f" % (
        (t0 - t_s) * 1e3, si, si // CMB, i, (r[1] - r[0]) * 1e3, (r[3] - r[2]) * 1e3, (r[2] - r[1]) * 1e3, (r[0] - prev[3]) * 1e3)
    if ch_kg:
        line += " | led keygen: " + " ".join("%s %.2f" % (PH[k], r[4][k] * 1e3) for k in range(10) if r[4][k] * 1e3 > 1.5 * mp[k] + 0.05)
    if ch_vf:
        line += " | led verify: " + " ".join("%s %.2f" % (PH[k], r[5][k] * 1e3) for k in range(10, 16) if r[5][k] * 1e3 > 1.5 * mp[k] + 0.05)
    print(line)
for sl in slots:
    sl.c.close()
