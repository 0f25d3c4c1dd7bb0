"""Does the FIRST call on a fresh handle see its constant tables?  (round-2 SIGABRT / round-3 rejected-honest-proofs investigation)

A fresh KOSK_STREAMS=3 handle is created and used at once, over and over, while a background thread keeps the host-to-device
copy path busy with large transfers.  Round 2's library uploaded its tables with plain hipMemcpy / hipMemset (legacy null stream),
which is not ordered against the contexts' non-blocking streams: the first kernels could run before the tables had landed.
Run from the root of the tree to be probed:   python3 tools/init_race_probe.py [iterations]
Prints one line per iteration that went wrong; exit code 1 if any did (a GPU fault aborts the process instead)."""
import hashlib
import os
import sys
import threading

sys.path.insert(0, os.getcwd())
import torch
from mpcith_kyber_kosk_amd import api

k, n = 3, 7
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 60
tapes = [hashlib.shake_256(("kosk-tape-v1:%d" % (40 + b)).encode()).digest(api.tape_bytes(k)) for b in range(n)]
ref_h = api.Kosk(kyber_k=k, max_batch=n)
ref = ref_h.verifiable_keygen(tapes)
assert ref_h.verify(ref[2], ref[0]) == [True] * n

stop = False
def traffic():
    host = torch.empty(512 << 20, dtype=torch.uint8).pin_memory()
    dev = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")
    st = torch.cuda.Stream()
    while not stop:
        with torch.cuda.stream(st):
            dev.copy_(host, non_blocking=True)
        st.synchronize()
th = threading.Thread(target=traffic, daemon=True)
th.start()
bad = 0
for it in range(iters):
    os.environ["KOSK_STREAMS"] = "3"
    h = api.Kosk(kyber_k=k, max_batch=6)
    del os.environ["KOSK_STREAMS"]
    got = h.verifiable_keygen(tapes)          # first call on the fresh handle: prover tables
    ok = h.verify(ref[2], ref[0])             # first verify on it: verifier tables, zeroed opened matrix
    if got != ref or ok != [True] * n:
        bad += 1
        print("iteration %d: proofs equal %s, verify bits %s" % (it, got == ref, ok), flush=True)
    h.close()
stop = True
th.join()
print("init_race_probe: %d of %d fresh handles misbehaved on their first calls" % (bad, iters))
sys.exit(1 if bad else 0)
