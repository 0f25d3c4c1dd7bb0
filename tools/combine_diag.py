"""Diagnostic: which calls of barrier-aligned caller threads end up merged (kosk_options::combine = 3).  Prints, per thread and call, the size
of the run that served it and the time since the previous combiner call of that thread."""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpcith_kyber_kosk_amd import api
from tests import oracle_lib as oracle
k, per, threads, rounds = 2, 3, 6, 4
hs = [api.Kosk(kyber_k=k, max_batch=per, combine=3, combine_wait_us=200000, combine_idle_us=100000) for _ in range(threads)]
tapes = [[oracle.tape_bytes_for(k, 100 + t * per + b) for b in range(per)] for t in range(threads)]
barrier = threading.Barrier(threads)
log = [[] for _ in range(threads)]
def worker(t):
    h = hs[t]
    h.verifiable_keygen_resident(tapes[t])
    barrier.wait()
    last = time.perf_counter()
    for r in range(rounds):
        for kind in "KV":
            barrier.wait()
            c0 = h.combine_stats()
            t0 = time.perf_counter()
            if kind == "K":
                h.verifiable_keygen_resident(tapes[t])
            else:
                assert h.verify_resident_pk(per) == [True] * per
            t1 = time.perf_counter()
            c1 = h.combine_stats()
            log[t].append((kind, c1[1] - c0[1], round((t0 - last) * 1e3, 2), round((t1 - t0) * 1e3, 2)))
            last = t1
            h.fetch_proofs(per)
ths = [threading.Thread(target=worker, args=(t,)) for t in range(threads)]
[x.start() for x in ths]; [x.join() for x in ths]
for t in range(threads):
    print(t, log[t])
for h in hs:
    h.close()
