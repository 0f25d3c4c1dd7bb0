#!/usr/bin/env python3
"""PCIe-inclusive rate of the two host-buffer calls from ONE caller thread, by number of sub-contexts (kosk_options::streams) and host
threads per sub-context. Not product code.  usage: stream_bench.py [streams ...]"""
import ctypes as C, hashlib, os, sys, time
sys.path.insert(0, ".")
k, B = 3, 46
for S in [int(x) for x in sys.argv[1:]] or [1, 3, 4, 6]:
    from mpcith_kyber_kosk_amd import api
    c = api.Kosk(kyber_k=k, max_batch=S * B, streams=S, host_threads=int(os.environ.get("STREAM_BENCH_THREADS", "0")))
    n = 2 * S * B
    tapes = [hashlib.shake_256(("kosk-tape-v1:%d" % b).encode()).digest(c.tape_bytes) for b in range(B)]
    blob = C.create_string_buffer(b"".join(tapes) * (2 * S), c.tape_bytes * n)
    pk = C.create_string_buffer(c.pk_bytes * n); sk = C.create_string_buffer(c.sk_bytes * n)
    pi = C.create_string_buffer(c.proof_bytes * n); ok = C.create_string_buffer(n)
    lib, h = api.lib, c.handle
    def prove(): assert lib.kosk_verifiable_keygen_batch(h, n, blob, c.tape_bytes, pk, sk, pi) == 0
    def verify(): assert lib.kosk_verify_batch(h, n, pi, pk, ok) == 0 and ok.raw == b"\x01" * n
    prove(); verify()
    t0 = time.perf_counter()
    for _ in range(3): prove()
    tp = (time.perf_counter() - t0) / 3
    t0 = time.perf_counter()
    for _ in range(3): verify()
    tv = (time.perf_counter() - t0) / 3
    print("streams=%d threads/sub=%s: %d proofs per call: keygen+prove+fetch %.1f ms (%.0f/s), stage+verify %.1f ms (%.0f/s), both %.0f proofs/s"
          % (S, os.environ.get("STREAM_BENCH_THREADS", "default"), n, tp * 1e3, n / tp, tv * 1e3, n / tv, n / (tp + tv)))
    c.close()
