"""Ablation driver for k_assemble_groups (KOSK_ASM_DBG, temporary): 276-proof key generations with proof only, no verifier (the images are wrong
by construction under a switch)."""
import os, sys, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpcith_kyber_kosk_amd import api
k, B = 3, 276
ctx = api.Kosk(kyber_k=k, max_batch=B)
tapes = [hashlib.shake_256(("kosk-tape-v1:%d" % b).encode()).digest(ctx.tape_bytes) for b in range(B)]
for _ in range(6):
    ctx.verifiable_keygen_resident(tapes)
ctx.synchronize()
print("done")
