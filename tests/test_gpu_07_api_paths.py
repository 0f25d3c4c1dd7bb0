"""GPU tests of the API paths around the kernels: randomness sources, context reuse, no stale state."""
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU: torch.cuda.is_available() is False")
    return torch


def test_randombytes_callback_consumes_tape_in_reference_order(oracle, torch_cuda):
    """kosk_set_randombytes: the library must draw 64, M x 32, then 302-byte blocks."""
    from mpcith_kyber_kosk_amd import api
    k = 3
    p = oracle.params(k)
    tapes = [oracle.tape_bytes_for(k, 20), oracle.tape_bytes_for(k, 21)]
    stream = b"".join(tapes)
    pos = [0]
    calls = []

    def rb(n):
        calls.append(n)
        out = stream[pos[0]:pos[0] + n]
        pos[0] += n
        return out
    ctx = api.Kosk(kyber_k=k, max_batch=2)
    ctx.set_randombytes(rb)
    pks, sks, pis = ctx.verifiable_keygen(None, n=2)
    assert pos[0] == 2 * p.tape_bytes and len(calls) == 2 * p.tape_calls
    assert calls[:2] == [64, 32] and calls[1 + p.M] == 302 and set(calls) == {64, 32, 302}
    for b in range(2):
        opk, osk, opi, _, _ = oracle.verifiable_keygen(k, tapes[b])
        assert (pks[b], sks[b], pis[b]) == (opk, osk, opi)
    # OS entropy (callback removed): proofs verify and differ
    ctx.set_randombytes(None)
    pks2, sks2, pis2 = ctx.verifiable_keygen(None, n=2)
    assert ctx.verify(pis2, pks2) == [True, True]
    assert pis2[0] != pis2[1] and pks2[0] != pks2[1] and pis2[0] != pis[0]
    ctx.close()


def test_fresh_context_verifies_foreign_proofs_without_stale_state(oracle, torch_cuda):
    from mpcith_kyber_kosk_amd import api
    k = 2
    p = oracle.params(k)
    prover = api.Kosk(kyber_k=k, max_batch=4)
    tapes = [oracle.tape_bytes_for(k, 30 + b) for b in range(4)]
    pks, sks, pis = prover.verifiable_keygen(tapes)
    prover.close()
    verifier = api.Kosk(kyber_k=k, max_batch=3)  # never proved anything; max_batch < n exercises chunking
    # tampered first: nothing correct may be lying around from an earlier call
    bad = []
    for f in (0, 6, 9, 13, 17, 19, 21, 4, 23, 5):
        t = bytearray(pis[0]); t[p.off[f] + 3] ^= 0x10; bad.append(bytes(t))
    got = verifier.verify(bad, [pks[0]] * len(bad))
    exp = [oracle.kosk_verify(k, t, pks[0])[0] for t in bad]
    assert got == exp and not any(got)
    masks = verifier.fail_masks(1)
    assert verifier.verify(pis, pks) == [True] * 4
    # swapped proofs / keys after a successful batch (stale CORRECT rows are now resident)
    assert verifier.verify([pis[1], pis[0], pis[2]], [pks[0], pks[1], pks[2]]) == [False, False, True]
    # and the prover side: the same context proves two different batches back to back
    ctx = api.Kosk(kyber_k=k, max_batch=2)
    a = ctx.verifiable_keygen(tapes[:2])
    b = ctx.verifiable_keygen(tapes[2:])
    assert a[2] == pis[:2] and b[2] == pis[2:]
    ctx.close()
    verifier.close()


def test_larger_batch_and_bench_shape(oracle, torch_cuda):
    """64 proofs in one call (> 46 of the bench, ragged against every tile size), spot-checked against the oracle."""
    from mpcith_kyber_kosk_amd import api
    k = 3
    n = 64
    ctx = api.Kosk(kyber_k=k, max_batch=n)
    tapes = [oracle.tape_bytes_for(k, 100 + b) for b in range(n)]
    pks, sks, pis = ctx.verifiable_keygen(tapes)
    assert ctx.verify(pis, pks) == [True] * n
    for b in (0, 31, 63):
        opk, osk, opi, _, _ = oracle.verifiable_keygen(k, tapes[b])
        assert (pks[b], sks[b], pis[b]) == (opk, osk, opi)
    assert len(set(pis)) == n
    ctx.close()


@pytest.mark.gpu
def test_pipeline_slots_stress(torch_cuda):
    """Three contexts proving and verifying concurrently from three host threads (bench.py's pipeline slots):
    every honest proof must verify and the proof bytes must not depend on what the other slots are doing."""
    import hashlib
    import threading
    from mpcith_kyber_kosk_amd import api
    S, B, N = 3, 46, 120  # 360 prove+verify steps: a kernel that misbehaves once per 1 000 launches under load shows up
    slots = [api.Kosk(kyber_k=3, max_batch=B, device=0) for _ in range(S)]
    ref = []
    for si, c in enumerate(slots):
        tapes = [hashlib.shake_256(b"kosk-tape-v1:%d" % (si * B + i)).digest(c.tape_bytes) for i in range(B)]
        c.stage_prover_inputs(tapes)
        c.prove_resident(B)
        assert all(c.verify_resident(B))
        ref.append(hashlib.sha3_256(b"".join(c.fetch_proofs(B))).hexdigest())
    bad = []

    def work(si):
        c = slots[si]
        for it in range(N):
            c.prove_resident(B)
            ok = c.verify_resident(B)
            if not all(ok):
                bad.append((si, it, ok.count(False), c.fail_masks(B)))
        if hashlib.sha3_256(b"".join(c.fetch_proofs(B))).hexdigest() != ref[si]:
            bad.append((si, "proof bytes changed"))
    th = [threading.Thread(target=work, args=(si,)) for si in range(S)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not bad, bad


def test_graph_replay_path_gives_identical_proofs(oracle, torch_cuda, monkeypatch):
    """KOSK_GRAPHS=1: every pipeline segment captured once and replayed as a hipGraph (also across a batch-size change,
    which forces a re-capture); proofs must not differ from the plain-launch path or the oracle."""
    from mpcith_kyber_kosk_amd import api
    k = 3
    tapes = [oracle.tape_bytes_for(k, 60 + i) for i in range(3)]
    plain = api.Kosk(kyber_k=k, max_batch=3)
    want = plain.verifiable_keygen(tapes)
    monkeypatch.setenv("KOSK_GRAPHS", "1")
    g = api.Kosk(kyber_k=k, max_batch=3)
    monkeypatch.delenv("KOSK_GRAPHS")
    for rep in range(3):  # first call captures, later calls replay
        got = g.verifiable_keygen(tapes)
        assert got == want
        assert g.verify(got[2], got[0]) == [True, True, True]
    got2 = g.verifiable_keygen(tapes[:2])  # different batch size: segments are re-captured
    assert got2[2] == want[2][:2]
    assert g.verify(got2[2], got2[0]) == [True, True]
    assert want[2][0] == oracle.verifiable_keygen(k, tapes[0])[2]
