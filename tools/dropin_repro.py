"""Standalone repro of bench.py's drop_in leg: KOSK_STREAMS=3 handle, 276 proofs per call on pageable / pinned buffers."""
import ctypes as C
import hashlib
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpcith_kyber_kosk_amd import api

k, B = 3, 46
lib = api.lib
os.environ["KOSK_STREAMS"] = sys.argv[1] if len(sys.argv) > 1 else "3"
h = api.Kosk(kyber_k=k, max_batch=3 * B)
del os.environ["KOSK_STREAMS"]
n = 6 * B
tapes = [hashlib.shake_256(("kosk-tape-v1:%d" % b).encode()).digest(h.tape_bytes) for b in range(B)]
blob = C.create_string_buffer(b"".join(tapes) * 6, h.tape_bytes * n)
pk = C.create_string_buffer(h.pk_bytes * n); sk = C.create_string_buffer(h.sk_bytes * n); ok = C.create_string_buffer(n)
buf = C.create_string_buffer(h.proof_bytes * n)
ref = None
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 4):
    assert lib.kosk_verifiable_keygen_batch(h.handle, n, blob, h.tape_bytes, pk, sk, buf) == 0, lib.kosk_last_error(h.handle)
    dig = [hashlib.sha3_256(buf.raw[i * h.proof_bytes:(i + 1) * h.proof_bytes]).hexdigest()[:8] for i in range(n)]
    if ref is None:
        ref = dig
    same_as_first_chunk = [dig[i] == dig[i % B] for i in range(n)]
    assert lib.kosk_verify_batch(h.handle, n, buf, pk, ok) == 0, lib.kosk_last_error(h.handle)
    bad = [i for i, b_ in enumerate(ok.raw) if b_ != 1]
    print("iter", it, "proofs equal to chunk 0's:", all(same_as_first_chunk), "chunks differing:", sorted({i // B for i, s_ in enumerate(same_as_first_chunk) if not s_}),
          "rejected:", len(bad), "chunks:", sorted({i // B for i in bad}), "masks:", sorted(set(h.fail_masks(n)[i] for i in bad))[:4], h.path_counts())
