#!/usr/bin/env python3
"""Latency of ONE proof (keygen + prove + verify, resident) on an otherwise idle GPU, with and without hipGraph replay of the
pipeline segments.  Not product code.  usage: latency_one.py [kyber_k]"""
import os, sys, time, hashlib
sys.path.insert(0, ".")
k = int(sys.argv[1]) if len(sys.argv) > 1 else 3
for graphs in ("0", "1"):
    os.environ["KOSK_GRAPHS"] = graphs
    from mpcith_kyber_kosk_amd import api
    c = api.Kosk(kyber_k=k, max_batch=1)
    tapes = [[hashlib.shake_256(("kosk-tape-v1:%d" % (1000 + i)).encode()).digest(c.tape_bytes)] for i in range(40)]
    ts, tp, tv = [], [], []
    for i in range(40):
        t0 = time.perf_counter()
        c.verifiable_keygen_resident(tapes[i])
        t1 = time.perf_counter()
        assert c.verify_resident_pk(1) == [True]
        t2 = time.perf_counter()
        ts.append(t2 - t0); tp.append(t1 - t0); tv.append(t2 - t1)
    med = lambda v: sorted(v[8:])[len(v[8:]) // 2] * 1e3
    print("KOSK_GRAPHS=%s  K=%d  one proof: keygen+prove %.3f ms, verify %.3f ms, both %.3f ms (median of 32)" % (graphs, k, med(tp), med(tv), med(ts)))
    c.close()
