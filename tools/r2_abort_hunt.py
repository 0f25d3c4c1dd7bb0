"""Round-2 SIGABRT hunt: the streamed host-buffer path of the ROUND-2 TREE (git worktree _r2tree = 7171ff3^), many
fresh handles with KOSK_STREAMS=3, first calls included, under AMD_LOG_LEVEL=1 + LIBC_FATAL_STDERR_=1 so that a ROCclr queue abort or a
glibc heap abort would leave its message on stderr.  Run from the repository root:   python tools/r2_abort_hunt.py 200"""
import os, sys, hashlib, time
root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "_r2tree")
sys.path.insert(0, root)
os.chdir(root)
os.environ["KOSK_STREAMS"] = "3"
from mpcith_kyber_kosk_amd import api  # the round-2 library
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
k, n = 3, 7
tapes = [hashlib.shake_256(("kosk-tape-v1:%d" % (40 + b)).encode()).digest(api.tape_bytes(k)) for b in range(n)]
ref = None
t0 = time.time()
for it in range(iters):
    h = api.Kosk(kyber_k=k, max_batch=6)          # a fresh three-lane handle every iteration: its FIRST call is the suspect
    got = h.verifiable_keygen(tapes)
    if ref is None:
        ref = got
    assert got == ref, it
    assert h.verify(got[2], got[0]) == [True] * n, it
    if it % 4 == 0:                                 # and a second call on some of them
        assert h.verifiable_keygen(tapes) == ref, it
    h.close()
    if it % 20 == 19:
        print("iteration %d ok (%.0f s)" % (it + 1, time.time() - t0), flush=True)
print("r2_abort_hunt: %d fresh KOSK_STREAMS=3 handles, keygen + verify each: no abort, every proof identical" % iters)
