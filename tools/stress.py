#!/usr/bin/env python3
"""Multi-slot stress: prove+verify loops on S slots, report every rejected honest proof with its fail mask,
and re-verify the same resident proofs to tell a bad proof from a bad verification.  Not product code."""
import sys, threading, hashlib, collections
sys.path.insert(0, ".")
from mpcith_kyber_kosk_amd import api
S = int(sys.argv[1]) if len(sys.argv) > 1 else 6
N = int(sys.argv[2]) if len(sys.argv) > 2 else 400
B, k = 46, 3
slots = [api.Kosk(kyber_k=k, max_batch=B, device=0) for _ in range(S)]
for si, c in enumerate(slots):
    t = [hashlib.shake_256(b"kosk-tape-v1:%d" % (si * B + i)).digest(c.tape_bytes) for i in range(B)]
    c.stage_prover_inputs(t); c.prove_resident(B); assert all(c.verify_resident(B))
ref = [hashlib.sha3_256(b"".join(c.fetch_proofs(B))).hexdigest() for c in slots]
lock = threading.Lock(); events = []
def work(si):
    c = slots[si]
    for it in range(N):
        c.prove_resident(B)
        ok = c.verify_resident(B)
        if not all(ok):
            m1 = c.fail_masks(B)
            ok2 = c.verify_resident(B); m2 = c.fail_masks(B)
            d = hashlib.sha3_256(b"".join(c.fetch_proofs(B))).hexdigest()
            with lock:
                events.append((si, it, ok.count(False), collections.Counter(hex(x) for x in m1 if x), ok2.count(False),
                               collections.Counter(hex(x) for x in m2 if x), d == ref[si]))
th = [threading.Thread(target=work, args=(si,)) for si in range(S)]
[t.start() for t in th]; [t.join() for t in th]
print("slots %d x %d steps: %d failing verifications" % (S, N, len(events)))
for e in events[:12]:
    print("  slot %d step %d: %d rejected, masks %s | re-verify: %d rejected, masks %s | proof bytes as reference: %s" % e)
