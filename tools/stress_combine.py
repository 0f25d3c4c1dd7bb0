#!/usr/bin/env python3
"""Soak of the combined path (kosk_options::combine = $STRESS_COMBINE, default 6 = bench.py's default since round 5): S caller threads, one handle each, N x (kosk_verifiable_keygen_resident on device tapes +
kosk_verify_resident_pk).  Every verify bit is checked; every 50th step the proofs and keys of the step are compared with the ones an
uncombined handle produced for the same tapes.  Not product code.     python tools/stress_combine.py 18 3000 [check interval]"""
import os, sys, threading, hashlib, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
CMB = int(os.environ.get("STRESS_COMBINE", "6"))
FS = 1 if os.environ.get("STRESS_FS", "host") == "device" else 0  # STRESS_FS=device: the cohorts hash on the GPU
from mpcith_kyber_kosk_amd import api
S = int(sys.argv[1]) if len(sys.argv) > 1 else 18
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
EVERY = int(sys.argv[3]) if len(sys.argv) > 3 else 50  # byte comparison every EVERY-th step (0: never; it stalls the caller for ~60 ms)
B, k = 46, 3
slots = [api.Kosk(kyber_k=k, max_batch=B, device=0, combine=CMB, fs_mode=FS) for _ in range(S)]
plain = api.Kosk(kyber_k=k, max_batch=B, device=0)
stride = (plain.tape_bytes + 63) // 64 * 64
banks, want = [], []
for si in range(S):
    tp = [hashlib.shake_256(b"kosk-tape-v1:%d" % (7000 + si * B + i)).digest(plain.tape_bytes) for i in range(B)]
    host = np.zeros((B, stride), np.uint8)
    for b, t in enumerate(tp):
        host[b, :len(t)] = np.frombuffer(t, np.uint8)
    banks.append(torch.from_numpy(host).to("cuda"))
    plain.verifiable_keygen_resident(tp)
    want.append((plain.keys(B), hashlib.sha3_256(b"".join(plain.fetch_proofs(B))).hexdigest()))
torch.cuda.synchronize()
bad, lock = [], threading.Lock()
def work(si):
    c = slots[si]
    try:
        for it in range(N):
            c.verifiable_keygen_resident(banks[si].data_ptr(), n=B, tape_stride=stride)
            ok = c.verify_resident_pk(B)
            if not all(ok):
                with lock: bad.append((si, it, "rejected %d, masks %s" % (ok.count(False), [hex(m) for m in c.fail_masks(B) if m][:4])))
            if EVERY and it % EVERY == EVERY - 1:
                if c.keys(B) != want[si][0] or hashlib.sha3_256(b"".join(c.fetch_proofs(B))).hexdigest() != want[si][1]:
                    with lock: bad.append((si, it, "keys or proof bytes differ from the uncombined handle's"))
    except Exception as e:  # noqa: BLE001
        with lock: bad.append((si, -1, repr(e)))
t0 = time.time()
th = [threading.Thread(target=work, args=(si,)) for si in range(S)]
[t.start() for t in th]; [t.join() for t in th]
dt = time.time() - t0
calls = sum(c.combine_stats()[0] for c in slots); members = sum(c.combine_stats()[1] for c in slots)
print("combined soak: %d callers x %d steps = %d proofs in %.0f s (%.0f proofs/s), mean callers per run %.2f: %d problems" % (S, N, S * N * B, dt, S * N * B / dt, members / max(1, calls), len(bad)))
for e in bad[:12]:
    print("  caller %d step %d: %s" % e)
for c in slots + [plain]:
    c.close()
sys.exit(1 if bad else 0)
