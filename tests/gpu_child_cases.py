"""Bodies of the GPU tests that run in a FRESH child Python process (tests/conftest.py: run_gpu_child): everything that puts
several library handles, lane threads and page-locked caller memory in play at once.  A crash in here is one failing test
with its stderr in the report, not the end of the pytest process and of every test behind it.

    python -c "from tests.gpu_child_cases import streamed_chunks; streamed_chunks(3)"
"""
import os

from tests import oracle_lib as oracle


def _kosk(k, max_batch, **kw):
    """a handle created with the given kosk_options fields (lower-case names: streams, combine, fs_mode, ...) and, for the few
    DEBUG knobs the library still reads from the environment at kosk_create (upper-case names: KOSK_REGISTER, KOSK_GRAPHS), under them"""
    from mpcith_kyber_kosk_amd import api
    opts = {n: v for n, v in kw.items() if n.islower()}
    env = {n: v for n, v in kw.items() if not n.islower()}
    old = {name: os.environ.get(name) for name in env}
    os.environ.update({name: str(v) for name, v in env.items()})
    try:
        return api.Kosk(kyber_k=k, max_batch=max_batch, **opts)
    finally:
        for name, v in old.items():
            if v is None:
                os.environ.pop(name, None)
            else:
                os.environ[name] = v


def streamed_chunks(k):
    """streams=3: a call longer than one sub-context is cut into chunks that run on three sub-contexts concurrently; n = 7
    over sub-batches of 2 leaves a ragged last chunk.  Same bytes as the single-context path and the oracle.  The library never
    page-locks the caller's (pageable) buffers -- KOSK_REGISTER=2 of rounds 2-4 is gone and is read as 1 -- so every chunk is staged."""
    from mpcith_kyber_kosk_amd import api
    n = 7
    tapes = [oracle.tape_bytes_for(k, 40 + b) for b in range(n)]
    plain = api.Kosk(kyber_k=k, max_batch=n)
    pks0, sks0, pis0 = plain.verifiable_keygen(tapes)
    assert plain.path_counts()["copy_direct"] == 0
    st = _kosk(k, 6, streams=3, KOSK_REGISTER=2)  # the retired value: behaves like the default
    assert st.streams == 3 and st.host_threads >= 1
    pks, sks, pis = st.verifiable_keygen(tapes)
    assert pks == pks0 and sks == sks0 and pis == pis0
    pc = st.path_counts()
    assert pc["copy_direct"] == 0 and pc["copy_staged"] == 4, pc  # 4 chunks of <= 2 proofs
    for b in (0, n - 1):
        opk, osk, opi, _, _ = oracle.verifiable_keygen(k, tapes[b])
        assert pks[b] == opk and sks[b] == osk and pis[b] == opi
    assert st.verify(pis, pks) == [True] * n
    # one bad proof in the ragged last chunk, one wrong key in the first: exactly those two fail, with the same masks as
    # on the single context
    bad = list(pis)
    p = oracle.params(k)
    flip = bytearray(bad[n - 1]); flip[p.off[0] + 5] ^= 1; bad[n - 1] = bytes(flip)
    keys = list(pks); keys[0] = pks[1]
    want = [False] + [True] * (n - 2) + [False]
    assert st.verify(bad, keys) == want
    m_st = st.fail_masks(n)
    assert plain.verify(bad, keys) == want
    assert plain.fail_masks(n) == m_st
    assert all((m != 0) == (not w) for m, w in zip(m_st, want))
    st2 = _kosk(k, 6, streams=3)
    pks2, sks2, pis2 = st2.verifiable_keygen(tapes)
    assert pks2 == pks0 and sks2 == sks0 and pis2 == pis0
    assert st2.verify(bad, keys) == want
    pc2 = st2.path_counts()
    assert pc2["copy_direct"] == 0 and pc2["copy_staged"] == 8, pc2
    for c in (plain, st, st2):
        c.close()
    print("streamed_chunks ok", k)


def streamed_loop(k, iters):
    """The round-2 abort's territory, many times over: several live handles (1, 2 and 3 lanes), streamed keygen + verify in
    a loop, caller buffers allocated and freed every iteration (rounds 2-4 page-locked spans of them per call; that is gone)."""
    from mpcith_kyber_kosk_amd import api
    n = 7
    tapes = [oracle.tape_bytes_for(k, 40 + b) for b in range(n)]
    plain = api.Kosk(kyber_k=k, max_batch=n)
    ref = plain.verifiable_keygen(tapes)
    hs = [_kosk(k, 6, streams=3), _kosk(k, 4, streams=2), _kosk(k, 6, streams=3, KOSK_REGISTER=0), _kosk(k, 3)]
    for it in range(iters):
        h = hs[it % len(hs)]
        got = h.verifiable_keygen(tapes)
        assert got == ref, it
        assert h.verify(got[2], got[0]) == [True] * n, it
        if it % 16 == 5:  # handles come and go while the others stay alive
            hs[1].close()
            hs[1] = _kosk(k, 4, streams=2)
    for h in hs + [plain]:
        h.close()
    print("streamed_loop ok", k, iters)


def errors_do_not_kill(k):
    """Error containment at the ABI: bad arguments, a handle whose creation fails, a batch call on a destroyed-and-recreated
    handle -- every failure is a return code with text, and the next call on a live handle works."""
    import ctypes as C
    from mpcith_kyber_kosk_amd import api
    lib = api.lib
    h = C.c_void_p()
    assert lib.kosk_create(C.byref(h), 0, 7, 4) != 0 and b"kyber_k" in lib.kosk_last_error(None)
    assert lib.kosk_create(C.byref(h), 0, k, 0) != 0 and b"max_batch" in lib.kosk_last_error(None)
    assert lib.kosk_create(C.byref(h), 99, k, 1) != 0 and lib.kosk_last_error(None) != b""
    st = _kosk(k, 4, streams=2)
    tapes = [oracle.tape_bytes_for(k, 90 + b) for b in range(5)]
    guard = C.create_string_buffer(64)
    assert lib.kosk_verifiable_keygen_batch(st.handle, 5, None, 0, None, guard, guard) != 0
    assert lib.kosk_verify_batch(st.handle, 5, guard, None, guard) != 0
    # a tape stride smaller than a tape: refused by every lane, reported once, handle still good
    blob = b"".join(tapes)
    pk = C.create_string_buffer(st.pk_bytes * 5); sk = C.create_string_buffer(st.sk_bytes * 5); pi = C.create_string_buffer(st.proof_bytes * 5)
    assert lib.kosk_verifiable_keygen_batch(st.handle, 5, C.c_char_p(blob), 16, pk, sk, pi) != 0
    assert b"tape_stride" in lib.kosk_last_error(st.handle)
    got = st.verifiable_keygen(tapes)
    assert st.verify(got[2], got[0]) == [True] * 5
    # fail masks: only of the last completed verify call
    assert st.fail_masks(5) == [0] * 5
    assert st.verify(got[2][:2], got[0][:2]) == [True, True]
    m = (C.c_uint32 * 5)()
    assert lib.kosk_verify_fail_masks(st.handle, m, 5) != 0 and lib.kosk_verify_fail_masks(st.handle, m, 2) == 0
    # pk == NULL on a handle that never generated keys: an error, not stale keys
    fresh = api.Kosk(kyber_k=k, max_batch=2)
    ok = C.create_string_buffer(2)
    assert lib.kosk_verify_resident_pk(fresh.handle, 2, None, ok) != 0 and b"resident public keys" in lib.kosk_last_error(fresh.handle)
    fresh.close()
    st.close()
    print("errors_do_not_kill ok", k)


def pinned_buffers(k):
    """Proof buffers from kosk_host_alloc: every chunk of a host-buffer call is copied straight to / from the caller's memory
    (single-chunk calls too), same bytes as through pageable memory; KOSK_REGISTER=0 keeps even those on the staging path."""
    import ctypes as C
    from mpcith_kyber_kosk_amd import api
    lib = api.lib
    n = 5
    tapes = [oracle.tape_bytes_for(k, 140 + b) for b in range(n)]
    plain = api.Kosk(kyber_k=k, max_batch=n)
    ref = plain.verifiable_keygen(tapes)
    blob = b"".join(tapes)
    for streams, cap, reg in ((1, n, 1), (2, 4, 1), (2, 4, 0)):
        h = _kosk(k, cap, streams=streams, KOSK_REGISTER=reg)
        nbytes = h.proof_bytes * n
        ptr = lib.kosk_host_alloc(nbytes)
        assert ptr
        pk = C.create_string_buffer(h.pk_bytes * n); sk = C.create_string_buffer(h.sk_bytes * n); ok = C.create_string_buffer(n)
        assert lib.kosk_verifiable_keygen_batch(h.handle, n, C.c_char_p(blob), h.tape_bytes, pk, sk, C.c_void_p(ptr)) == 0, lib.kosk_last_error(h.handle)
        got = C.string_at(ptr, nbytes)
        assert got == b"".join(ref[2]) and pk.raw == b"".join(ref[0]) and sk.raw == b"".join(ref[1])
        assert lib.kosk_verify_batch(h.handle, n, C.c_void_p(ptr), pk, ok) == 0 and ok.raw == b"\x01" * n
        bad = bytearray(got); bad[h.proof_bytes * 3 + oracle.params(k).off[0] + 5] ^= 1
        C.memmove(ptr, bytes(bad), nbytes)
        assert lib.kosk_verify_batch(h.handle, n, C.c_void_p(ptr), pk, ok) == 0 and ok.raw == b"\x01\x01\x01\x00\x01"
        pc = h.path_counts()
        chunks = -(-n // (cap // streams if streams > 1 else cap))
        if reg:
            assert pc["copy_staged"] == 0 and pc["copy_direct"] == 3 * chunks, (streams, pc)
        else:
            assert pc["copy_direct"] == 0 and pc["copy_staged"] == 3 * chunks, (streams, pc)
        lib.kosk_host_free(C.c_void_p(ptr))
        h.close()
    plain.close()
    print("pinned_buffers ok", k)


def big_batches():
    """The batch shapes of BASELINE configs[2..4] through the host-buffer calls (chunked: n above the context's capacity), the
    resident split, the compact staging and the second-level entry points, in one fresh process."""
    from mpcith_kyber_kosk_amd import api
    for k, n in ((3, 46), (4, 91), (3, 130)):
        ctx = api.Kosk(kyber_k=k, max_batch=min(n, 91))
        tapes = [oracle.tape_bytes_for(k, 1000 + b) for b in range(n)]
        pks, sks, pis = ctx.verifiable_keygen(tapes)
        opk, osk, opi, _, _ = oracle.verifiable_keygen(k, tapes[n - 1])
        assert (pks[-1], sks[-1], pis[-1]) == (opk, osk, opi)
        assert ctx.verify(pis, pks) == [True] * n
        m = min(n, 91)
        ctx.verifiable_keygen_resident(tapes[:m])
        assert ctx.verify_resident_pk(m) == [True] * m
        blobs = ctx.fetch_proofs_compact(m)
        ctx.stage_verifier_inputs_compact(blobs, pks[:m])
        assert ctx.verify_resident(m) == [True] * m
        ctx.close()
    # second-level entry points
    k = 2
    ctx = api.Kosk(kyber_k=k, max_batch=2)
    rnd = ctx.prepare_randomness(n=3)
    rng = ctx.prepare_range_proof(n=3)
    assert len(rnd) == 3 and len(rng) == 3
    ctx.close()
    print("big_batches ok")


def combined_calls(k, threads=6, rounds=4, min_merge=0.5):
    """combine=3: six caller threads, each with its own handle and its own small resident calls.  The calls of a cohort's
    members are served by merged pipeline runs; every caller must get exactly what an uncombined handle gives it -- pk, sk, proof
    images, resident digest tables, verify bits and fail masks byte for byte -- with host tapes, device tapes read by a merged
    run, a short (ragged) batch on one member, and given / resident public keys.  The oracle pins three of the proofs."""
    import ctypes as C
    import threading
    import torch
    from mpcith_kyber_kosk_amd import api
    lib = api.lib
    per = 3
    plain = api.Kosk(kyber_k=k, max_batch=per)
    hs = [_kosk(k, per, combine=3, combine_wait_us=200000, combine_idle_us=100000) for _ in range(threads)]
    stride = (plain.tape_bytes + 63) // 64 * 64
    # what every (thread, round) must produce: from the uncombined handle
    want = {}
    tapes = {}
    for t in range(threads):
        for r in range(rounds):
            n = per if not (t == threads - 1 and r == 1) else per - 1  # one ragged call: the last member of its cohort, round 1
            tp = [oracle.tape_bytes_for(k, 5000 + (t * rounds + r) * per + b) for b in range(n)]
            tapes[t, r] = tp
            plain.verifiable_keygen_resident(tp)
            pk, sk = plain.keys(n)
            assert plain.verify_resident_pk(n) == [True] * n
            want[t, r] = (pk, sk, plain.fetch_proofs(n),
                          [torch.as_tensor(plain.resident_digests(i, n), device="cuda").cpu().numpy().tobytes() for i in (0, 1)])
    for (t, r) in ((0, 0), (threads - 1, 1), (2, rounds - 1)):
        opk, osk, opi, _, _ = oracle.verifiable_keygen(k, tapes[t, r][-1])
        assert (want[t, r][0][-1], want[t, r][1][-1], want[t, r][2][-1]) == (opk, osk, opi)
    errs = []
    barrier = threading.Barrier(threads)

    base = [None] * threads

    def worker(t):
        try:
            h = hs[t]
            # two unchecked rounds first: nobody is expected at a cohort's very first call, so it cannot merge.  The threads then
            # run freely (no barriers: a caller parked at a barrier while its cohort waits for it inside the library would only
            # test this test); the combiner itself brings the members of a cohort into step (tools/combine_diag.py)
            barrier.wait()
            for _ in range(2):
                h.verifiable_keygen_resident(tapes[t, 0])
                assert h.verify_resident_pk(len(tapes[t, 0])) == [True] * len(tapes[t, 0])
            base[t] = h.combine_stats()
            for r in range(rounds):
                tp = tapes[t, r]
                n = len(tp)
                if r % 2 == 0:  # host tapes
                    h.verifiable_keygen_resident(tp)
                else:           # device tapes (a merged run copies them into its own tape block; a lone run reads them in place)
                    import numpy as np
                    host = np.zeros((n, stride), np.uint8)
                    for b, x in enumerate(tp):
                        host[b, :len(x)] = np.frombuffer(x, np.uint8)
                    dev = torch.from_numpy(host).to("cuda")
                    torch.cuda.synchronize()
                    h.verifiable_keygen_resident(dev.data_ptr(), n=n, tape_stride=stride)
                pk, sk = h.keys(n)
                assert (pk, sk) == want[t, r][:2], ("keys", t, r)
                ok = h.verify_resident_pk(n, pks=pk if r == 2 else None)
                assert ok == [True] * n and h.fail_masks(n) == [0] * n, ("verify", t, r, ok)
                assert h.fetch_proofs(n) == want[t, r][2], ("proofs", t, r)
                for i in (0, 1):
                    got = torch.as_tensor(h.resident_digests(i, n), device="cuda").cpu().numpy().tobytes()
                    assert got == want[t, r][3][i], ("digests", t, r, i)
                # a wrong key for this caller only: its own masks say so, the run's other callers are untouched
                if r == rounds - 1:
                    keys = list(pk)
                    if t == 1:
                        keys[0] = want[0, r][0][0]
                    ok = h.verify_resident_pk(n, pks=keys)
                    assert ok == ([False] + [True] * (n - 1) if t == 1 else [True] * n), ("wrong key", t, ok)
                    assert (h.fail_masks(n)[0] != 0) == (t == 1)
        except Exception as e:  # noqa: BLE001
            errs.append((t, repr(e)))
            try:
                barrier.abort()
            except Exception:
                pass
    ths = [threading.Thread(target=worker, args=(t,)) for t in range(threads)]
    for x in ths:
        x.start()
    for x in ths:
        x.join()
    assert not errs, errs
    calls = sum(h.combine_stats()[0] - base[t][0] for t, h in enumerate(hs))
    members = sum(h.combine_stats()[1] - base[t][1] for t, h in enumerate(hs))
    assert calls == threads * (2 * rounds + 1)
    # the barrier-aligned calls really merged: every cohort's present members in one run (3, 3 for six threads; 3, 2 for five)
    ideal = sum(min(3, threads - c0) ** 2 for c0 in range(0, threads, 3)) / threads
    # (how many is a matter of thread timing: the bytes above are what is asserted strictly)
    assert members / calls >= min_merge * ideal and members > calls, (calls, members, ideal)
    # entry points that never merge keep working on a member's own block, next to its neighbours' merged state
    got = hs[1].verifiable_keygen(tapes[1, 0])
    assert got[2] == want[1, 0][2]
    assert hs[1].verify(got[2], got[0]) == [True] * per
    assert hs[0].fetch_proofs(per) == want[0, rounds - 1][2]  # neighbour's resident proofs untouched
    # pk == NULL without resident keys on a member: an error for that caller alone
    fresh = _kosk(k, per, combine=3)
    okb = C.create_string_buffer(per)
    assert lib.kosk_verify_resident_pk(fresh.handle, per, None, okb) != 0 and b"resident public keys" in lib.kosk_last_error(fresh.handle)
    for h in hs + [plain, fresh]:
        h.close()
    print("combined_calls ok", k, "mean callers per run %.2f" % (members / calls))


def line_of_record_shape(k=3, per=46, callers=3, rounds=3, fs_device=0):
    """The shape bench.py's line of record runs (bench.py: Slot.step), checked byte for byte: `callers` caller threads of ONE cohort
    (combine=callers: 6 is the bench's default since round 5, 4 and 3 its side runs), each with 46 Kyber-768 proofs per call on DEVICE tapes with a 64-byte-aligned stride that a merged run reads in
    place (the tape pointer table of the first kernel), the raw resident entry points with the key generation's pk / sk staying
    resident for the verifier (pk == NULL) -- so every launch covers 138 proofs: the single-buffer instantiation of the commitment
    hash (3 174 waves), three rounds of row blocks in the expansion product.  Every caller's pk / sk / proof images / both digest
    tables equal an uncombined handle's, the first and last proof of every caller equal the oracle's, and the digest tables of
    those proofs equal the oracle's Tcomm / view commitments."""
    import ctypes as C
    import threading
    import numpy as np
    import torch
    from mpcith_kyber_kosk_amd import api
    lib = api.lib
    plain = api.Kosk(kyber_k=k, max_batch=per)
    # fs_device=1: the cohort runs its Fiat-Shamir rounds on the GPU (kosk_options::fs_mode);
    # `plain`, the handle everything is compared with, hashes on the host
    hs = [_kosk(k, per, combine=callers, combine_wait_us=5000000, combine_idle_us=2000000,
                fs_mode=fs_device) for _ in range(callers)]
    stride = (plain.tape_bytes + 63) // 64 * 64
    nsets = rounds
    tapes = {(t, r): [oracle.tape_bytes_for(k, 20000 + ((t * nsets) + r) * per + b) for b in range(per)] for t in range(callers) for r in range(nsets)}
    banks = []
    for t in range(callers):
        host = np.zeros((nsets, per, stride), np.uint8)
        for r in range(nsets):
            for b, x in enumerate(tapes[t, r]):
                host[r, b, :len(x)] = np.frombuffer(x, np.uint8)
        banks.append(torch.from_numpy(host).to("cuda"))
    torch.cuda.synchronize()
    want = {}
    for key, tp in tapes.items():
        plain.verifiable_keygen_resident(tp)
        pk, sk = plain.keys(per)
        assert plain.verify_resident_pk(per) == [True] * per
        want[key] = (pk, sk, plain.fetch_proofs(per),
                     [torch.as_tensor(plain.resident_digests(i, per), device="cuda").cpu().numpy().tobytes() for i in (0, 1)])
    # the oracle on the first and the last proof of every caller (last round): images, keys and both digest tables
    for t in range(callers):
        for b in (0, per - 1):
            opk, osk, opi, _, _, tr = oracle.verifiable_keygen(k, tapes[t, nsets - 1][b], trace=True)
            w = want[t, nsets - 1]
            assert (w[0][b], w[1][b], w[2][b]) == (opk, osk, opi), ("oracle", t, b)
            assert w[3][0][b * 1454 * 32:(b + 1) * 1454 * 32] == bytes(tr.tcomm), ("tcomm", t, b)
            assert w[3][1][b * 1454 * 32:(b + 1) * 1454 * 32] == bytes(tr.view_digest), ("view", t, b)
    errs, base = [], [None] * callers
    barrier = threading.Barrier(callers)

    def worker(t):
        try:
            h = hs[t]
            pk = C.create_string_buffer(h.pk_bytes * per); sk = C.create_string_buffer(h.sk_bytes * per); ok = C.create_string_buffer(per)
            ptrs = [C.c_void_p(banks[t][r].data_ptr()) for r in range(nsets)]

            def step(r):
                assert lib.kosk_verifiable_keygen_resident(h.handle, per, ptrs[r], stride, pk, sk) == 0, lib.kosk_last_error(h.handle)
                assert lib.kosk_verify_resident_pk(h.handle, per, None, ok) == 0, lib.kosk_last_error(h.handle)
                assert ok.raw == b"\x01" * per, ("verify bits", t, r)
            barrier.wait()
            for _ in range(2):  # nobody is expected at a cohort's very first call: two unchecked steps bring the callers into step
                step(0)
            base[t] = h.combine_stats()
            for r in range(nsets):
                step(r)
                w = want[t, r]
                assert [pk.raw[i * h.pk_bytes:(i + 1) * h.pk_bytes] for i in range(per)] == w[0], ("pk", t, r)
                assert [sk.raw[i * h.sk_bytes:(i + 1) * h.sk_bytes] for i in range(per)] == w[1], ("sk", t, r)
                assert h.fail_masks(per) == [0] * per
                assert h.fetch_proofs(per) == w[2], ("proofs", t, r)
                for i in (0, 1):
                    got = torch.as_tensor(h.resident_digests(i, per), device="cuda").cpu().numpy().tobytes()
                    assert got == w[3][i], ("digests", t, r, i)
        except Exception as e:  # noqa: BLE001
            errs.append((t, repr(e)))
            try:
                barrier.abort()
            except Exception:
                pass
    ths = [threading.Thread(target=worker, args=(t,)) for t in range(callers)]
    for x in ths:
        x.start()
    for x in ths:
        x.join()
    assert not errs, errs
    calls = sum(h.combine_stats()[0] - base[t][0] for t, h in enumerate(hs))
    members = sum(h.combine_stats()[1] - base[t][1] for t, h in enumerate(hs))
    assert calls == callers * 2 * nsets, calls
    assert members == callers * calls, (calls, members)  # every checked call ran in a merged run of all the cohort's callers
    pc = hs[0].path_counts()
    assert sum(h.path_counts()["hash_dma"] for h in hs) > 0 and all(h.path_counts()["hash_plain"] == 0 for h in hs), pc
    nfs_dev, nfs_host, ncopy = (sum(h.path_counts()[nm] for h in hs) for nm in ("fs_device", "fs_host", "digest_copy"))
    assert (nfs_dev >= 4 * nsets and nfs_host == 0 and ncopy == 0) if fs_device else (nfs_dev == 0 and nfs_host >= 4 * nsets), (nfs_dev, nfs_host, ncopy)
    assert plain.path_counts()["fs_device"] == 0
    for h in hs + [plain]:
        h.close()
    print("line_of_record_shape ok", k, per, callers, "callers per run %.2f" % (members / calls), "fs_device" if fs_device else "fs_host")


def member_big_batch_stays_in_its_block(k=3, per=3):
    """a cohort member's view spans its neighbours' blocks of the shared workspace, so its non-merged entry points must
    chunk by the member's OWN batch size.  Member 0 makes host-buffer calls of 2 * per + 1 proofs (keygen, verify, the compact pair,
    the second-level prepare calls) while members 1 and 2 loop resident calls: member 0's results equal a plain handle's, and the
    neighbours' keys, resident proofs and verify bits never change."""
    import threading
    from mpcith_kyber_kosk_amd import api
    plain = api.Kosk(kyber_k=k, max_batch=per)
    hs = [_kosk(k, per, combine=3, combine_wait_us=2000, combine_idle_us=500) for _ in range(3)]
    n_big = 2 * per + 1
    big_tapes = [oracle.tape_bytes_for(k, 7000 + b) for b in range(n_big)]
    want_big = plain.verifiable_keygen(big_tapes)
    nb_tapes = {t: [oracle.tape_bytes_for(k, 7100 + t * per + b) for b in range(per)] for t in (1, 2)}
    want_nb = {}
    for t in (1, 2):
        plain.verifiable_keygen_resident(nb_tapes[t])
        want_nb[t] = (plain.keys(per), plain.fetch_proofs(per))
    errs, stop = [], threading.Event()

    def neighbour(t):
        try:
            h = hs[t]
            it = 0
            while not stop.is_set() or it < 3:
                h.verifiable_keygen_resident(nb_tapes[t])
                assert h.keys(per) == want_nb[t][0], ("neighbour keys", t, it)
                assert h.verify_resident_pk(per) == [True] * per, ("neighbour verify", t, it)
                assert h.fail_masks(per) == [0] * per
                assert h.fetch_proofs(per) == want_nb[t][1], ("neighbour proofs", t, it)
                it += 1
        except Exception as e:  # noqa: BLE001
            errs.append((t, repr(e)))
    ths = [threading.Thread(target=neighbour, args=(t,)) for t in (1, 2)]
    for x in ths:
        x.start()
    try:
        h0 = hs[0]
        for it in range(4):
            got = h0.verifiable_keygen(big_tapes)
            assert got == want_big, ("member 0 keygen", it)
            assert h0.verify(got[2], got[0]) == [True] * n_big, ("member 0 verify", it)
            bad = list(got[2]); flip = bytearray(bad[n_big - 1]); flip[100] ^= 1; bad[n_big - 1] = bytes(flip)
            assert h0.verify(bad, got[0]) == [True] * (n_big - 1) + [False]
        # the second-level entry points chunk the same way
        mo = [oracle.main_order(k, oracle.tape_bytes_for(k, 7300 + b)) for b in range(n_big)]
        rands = h0.prepare_randomness([oracle.tape_bytes_for(k, 7300 + b)[:mo[b]["used"][0]] for b in range(n_big)])
        assert rands == [m["rand"] for m in mo]
    except Exception as e:  # noqa: BLE001
        errs.append((0, repr(e)))
    finally:
        stop.set()
        for x in ths:
            x.join()
    assert not errs, errs
    for h in hs + [plain]:
        h.close()
    print("member_big_batch_stays_in_its_block ok", k)


def cohort_round_hooks(k=2, per=3, rounds=4):
    """Round hooks inside a cohort: member 0 has no hook and leads the merged runs; member 1 has a hook and merges (the default): its
    hook fires from the merged run, on the LEADER's thread, with member 1's own block of the table; member 2 has a hook and
    kosk_options::hooks_unmerged = 1: its calls never merge and its hook always fires on its own thread.  Member 2 also sends a shorter
    batch.  Every caller's bytes equal an uncombined handle's."""
    import hashlib
    import threading
    import torch
    from mpcith_kyber_kosk_amd import api
    plain = api.Kosk(kyber_k=k, max_batch=per)
    ns = [per, per, per - 1]
    tapes = [[oracle.tape_bytes_for(k, 9500 + t * per + b) for b in range(ns[t])] for t in range(3)]
    want = []
    for t in range(3):
        plain.verifiable_keygen_resident(tapes[t])
        want.append((plain.keys(ns[t]), plain.fetch_proofs(ns[t]),
                     [hashlib.sha3_256(torch.as_tensor(plain.resident_digests(r, ns[t]), device="cuda").cpu().numpy().tobytes()).hexdigest() for r in (0, 1)]))
    hs = [_kosk(k, per, combine=3, combine_wait_us=2000000, combine_idle_us=1000000),
          _kosk(k, per, combine=3, combine_wait_us=2000000, combine_idle_us=1000000),
          _kosk(k, per, combine=3, combine_wait_us=2000000, combine_idle_us=1000000, hooks_unmerged=1)]
    seen = {1: [], 2: []}
    idents = {}

    def mk_hook(t):
        def hook(role, rnd, ptr, nbytes):
            tab = torch.as_tensor(api.DeviceView(ptr, (ns[t], 1454, 32)), device="cuda")
            seen[t].append((threading.get_ident(), role, rnd, nbytes, hashlib.sha3_256(tab.cpu().numpy().tobytes()).hexdigest()))
        return hook
    hs[1].set_round_hook(mk_hook(1))
    hs[2].set_round_hook(mk_hook(2))
    errs = []
    barrier = threading.Barrier(3)

    def worker(t):
        try:
            idents[t] = threading.get_ident()
            h = hs[t]
            for r in range(rounds):
                barrier.wait()
                h.verifiable_keygen_resident(tapes[t])
                assert h.keys(ns[t]) == want[t][0], ("keys", t, r)
                assert h.verify_resident_pk(ns[t]) == [True] * ns[t], ("bits", t, r)
                assert h.fetch_proofs(ns[t]) == want[t][1], ("proofs", t, r)
        except Exception as e:  # noqa: BLE001
            errs.append((t, repr(e)))
            try:
                barrier.abort()
            except Exception:
                pass
    th = [threading.Thread(target=worker, args=(t,)) for t in range(3)]
    [x.start() for x in th]; [x.join() for x in th]
    assert not errs, errs
    for t in (1, 2):
        assert len(seen[t]) == 4 * rounds, (t, len(seen[t]))
        for ident, role, rnd, nbytes, dig in seen[t]:
            assert nbytes == ns[t] * 1454 * 32 and role in (0, 1) and rnd in (0, 1), (t, role, rnd, nbytes)
            if role == 0:
                assert dig == want[t][2][rnd], (t, role, rnd)  # the member's OWN block of the prover's round table
    assert all(s_[0] == idents[2] for s_ in seen[2])                  # unmerged: always the handle's own thread
    assert any(s_[0] == idents[0] for s_ in seen[1]), "member 1's hook never fired from a merged run led by member 0"
    c2, m2 = hs[2].combine_stats()
    assert c2 > 0 and m2 == c2                                         # every call of member 2 ran alone
    c1, m1 = hs[1].combine_stats()
    assert m1 > c1                                                     # member 1 did merge (with member 0)
    for h in hs + [plain]:
        h.close()
    print("cohort_round_hooks ok", k)


def combined_members_come_and_go(k):
    """A cohort whose members are destroyed and re-created while the others keep calling: the freed block is reused by the next
    handle, members that are no neighbours any more run on their own, the workspace lives until the last member is gone, a
    second cohort opens when the first is full -- and every caller still gets the uncombined handle's bytes."""
    import threading
    from mpcith_kyber_kosk_amd import api
    per = 2
    plain = api.Kosk(kyber_k=k, max_batch=per)
    mk = lambda: _kosk(k, per, combine=3, combine_wait_us=100000, combine_idle_us=50000)
    tapes = [[oracle.tape_bytes_for(k, 9000 + t * per + b) for b in range(per)] for t in range(5)]
    want = []
    for tp in tapes:
        plain.verifiable_keygen_resident(tp)
        want.append((plain.keys(per), plain.fetch_proofs(per)))

    def round_of(handles, idx):
        """every handle of `handles` (list of (handle, tape set)) does keygen + verify concurrently; returns nothing, asserts bytes"""
        errs = []

        def w(h, t):
            try:
                h.verifiable_keygen_resident(tapes[t])
                assert h.keys(per) == want[t][0]
                assert h.verify_resident_pk(per) == [True] * per
                assert h.fetch_proofs(per) == want[t][1]
            except Exception as e:  # noqa: BLE001
                errs.append((idx, t, repr(e)))
        th = [threading.Thread(target=w, args=(h, t)) for h, t in handles]
        [x.start() for x in th]; [x.join() for x in th]
        assert not errs, errs
    a, b, c = mk(), mk(), mk()
    round_of([(a, 0), (b, 1), (c, 2)], 0)
    round_of([(a, 0), (b, 1), (c, 2)], 1)
    b.close()                                   # the middle member goes: a and c are no neighbours any more
    round_of([(a, 3), (c, 4)], 2)
    assert a.fetch_proofs(per) == want[3][1] and c.fetch_proofs(per) == want[4][1]
    d = mk()                                    # takes the freed block between them
    round_of([(a, 0), (d, 1), (c, 2)], 3)
    e = mk()                                    # the cohort is full: a second one opens (and is alone in it)
    round_of([(a, 3), (d, 4), (c, 0), (e, 1)], 4)
    merged_before = sum(h.combine_stats()[1] - h.combine_stats()[0] for h in (a, c, d))
    assert merged_before > 0                    # the three-member rounds really merged
    a.close(); d.close()
    round_of([(c, 2), (e, 3)], 5)               # the last member of the first cohort still owns a live workspace
    c.close(); e.close()
    f = mk()                                    # both cohorts are gone: a fresh one
    round_of([(f, 4)], 6)
    f.close(); plain.close()
    print("combined_members_come_and_go ok", k)
