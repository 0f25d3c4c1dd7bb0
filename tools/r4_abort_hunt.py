# (round 5: KOSK_REGISTER=2, the knob this script hunted, has been removed from the library; kept for the record of profiles/r04_abort_hunt.txt)
"""Bounded hunt for the silent SIGABRT of round 4's one aborted suite run (profiles/r04_gputest_aborted.log): the FIRST call on a
fresh handle, a multi-chunk compact host-buffer call that page-locks the whole pages inside a pageable Python buffer (KOSK_REGISTER=2, the
default of rounds 2-3), next to another live handle -- `iters` times, every buffer allocated anew.  Run it with AMD_LOG_LEVEL=1 and
LIBC_FATAL_STDERR_=1 so that a ROCclr queue error or a glibc heap check leaves its message on stderr.
    python tools/r4_abort_hunt.py [iters]"""
import ctypes as C, faulthandler, hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
faulthandler.enable()
os.environ["KOSK_REGISTER"] = "2"
from mpcith_kyber_kosk_amd import api
lib = api.lib
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 150
n = 5
direct = staged = 0
for it in range(iters):
    k = (2, 3, 4)[it % 3]
    tapes = [hashlib.shake_256(b"hunt-%d-%d" % (it, b)).digest(api.tape_bytes(k)) for b in range(n)]
    ref = api.Kosk(kyber_k=k, max_batch=n)          # the live neighbour of the suite's test
    ref.verifiable_keygen_resident(tapes)
    ctx = api.Kosk(kyber_k=k, max_batch=2)          # fresh handle: its first call is the chunked one
    cb = lib.kosk_compact_proof_bytes(k)
    pk = C.create_string_buffer(ctx.pk_bytes * n); sk = C.create_string_buffer(ctx.sk_bytes * n)
    out = C.create_string_buffer(cb * n); ok = C.create_string_buffer(n)
    rc = lib.kosk_verifiable_keygen_batch_compact(ctx.handle, n, C.c_char_p(b"".join(tapes)), ctx.tape_bytes, pk, sk, out)
    assert rc == 0, lib.kosk_last_error(ctx.handle)
    assert lib.kosk_verify_batch_compact(ctx.handle, n, out, pk, ok) == 0 and ok.raw == b"\x01" * n
    pc = ctx.path_counts()
    direct += pc["copy_direct"]; staged += pc["copy_staged"]
    ref.close(); ctx.close()
    if it % 25 == 24:
        print("iteration %d ok (direct copies so far %d, staged %d)" % (it + 1, direct, staged), flush=True)
print("r4_abort_hunt: %d fresh handles, first call chunked with page-locking of the caller's buffer: no abort; %d direct / %d staged chunk copies" % (iters, direct, staged))
