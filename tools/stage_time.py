import os, sys, time, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpcith_kyber_kosk_amd import api
for B in (46, 512):
    ctx = api.Kosk(kyber_k=3, max_batch=B)
    tapes = [hashlib.shake_256(("kosk-tape-v1:%d" % b).encode()).digest(ctx.tape_bytes) for b in range(B)]
    ctx.stage_prover_inputs(tapes)
    t0 = time.perf_counter()
    for _ in range(10):
        ctx.stage_prover_inputs(tapes)
    dt = (time.perf_counter() - t0) / 10
    print("B=%d stage_prover_inputs %.3f ms (incl. python join of tapes), lib-internal %.3f ms" % (B, dt * 1e3, ctx.phase_seconds()[0] * 1e3))
    ctx.close()
