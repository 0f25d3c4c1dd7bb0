#!/usr/bin/env python3
"""Per-call wall time of prove_resident / verify_resident with S pipeline slots (not product code)."""
import sys, threading, time, hashlib
sys.path.insert(0, ".")
from mpcith_kyber_kosk_amd import api
S = int(sys.argv[1]) if len(sys.argv) > 1 else 6
N = int(sys.argv[2]) if len(sys.argv) > 2 else 60
B, k = 46, 3
slots = [api.Kosk(kyber_k=k, max_batch=B, device=0) for _ in range(S)]
for si, c in enumerate(slots):
    t = [hashlib.shake_256(b"kosk-tape-v1:%d" % (si * B + i)).digest(c.tape_bytes) for i in range(B)]
    c.stage_prover_inputs(t); c.prove_resident(B); c.verify_resident(B)
tp, tv = [[] for _ in range(S)], [[] for _ in range(S)]
NAMES = ["host_pre", "gpu_commit", "fs_alpha", "gpu_relation", "fs_open", "gpu_assemble", "d2h", "p1_issue", "p2_issue", "p3_issue",
         "v1_issue", "v1_wait", "v_fs_alpha", "v2_issue", "v2_wait", "v_fs_open"]
acc = [[0.0] * 16 for _ in range(S)]; cnt = [0] * S
def work(si):
    c = slots[si]
    for it in range(N):
        a = time.perf_counter(); c.prove_resident(B); b = time.perf_counter(); c.verify_resident(B); d = time.perf_counter()
        if it >= N // 4:
            tp[si].append(b - a); tv[si].append(d - b)
            ph = c.phase_seconds(); cnt[si] += 1
            for i_ in range(16): acc[si][i_] += ph[i_]
for rep in range(2):
    for l in tp + tv: l.clear()
    acc = [[0.0] * 16 for _ in range(S)]; cnt = [0] * S
    th = [threading.Thread(target=work, args=(si,)) for si in range(S)]
    t0 = time.perf_counter(); [t.start() for t in th]; [t.join() for t in th]; dt = time.perf_counter() - t0
allp = sum(tp, []); allv = sum(tv, [])
print("slots %d: %.3f ms/step wall; prove %.3f ms, verify %.3f ms per call (mean); sum %.3f ms = %.2f x slots*step" % (
    S, dt / (S * N) * 1e3, sum(allp) / len(allp) * 1e3, sum(allv) / len(allv) * 1e3,
    (sum(allp) / len(allp) + sum(allv) / len(allv)) * 1e3, (sum(allp) / len(allp) + sum(allv) / len(allv)) / (dt / N)))
print("  mean phases (us):", {NAMES[i_]: round(sum(acc[si][i_] for si in range(S)) / sum(cnt) * 1e6) for i_ in range(1, 16) if i_ != 6})
