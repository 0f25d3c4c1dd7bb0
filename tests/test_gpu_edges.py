"""GPU edge cases of the batch ABI: empty and ragged batches, the chunk-parallel host-buffer path (KOSK_STREAMS > 1), and
batches larger than the context (chunking), all against the single-context path and the CPU oracle, bit for bit."""
import ctypes as C

import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU: torch.cuda.is_available() is False")
    return torch


def test_empty_batches_are_no_ops(oracle, torch_cuda):
    from mpcith_kyber_kosk_amd import api
    ctx = api.Kosk(kyber_k=3, max_batch=2)
    lib, h = api.lib, ctx.handle
    guard = C.create_string_buffer(b"\xa5" * 64)
    assert lib.kosk_verifiable_keygen_batch(h, 0, None, 0, guard, guard, guard) == 0
    assert lib.kosk_verify_batch(h, 0, guard, guard, guard) == 0
    assert lib.kosk_verifiable_keygen_resident(h, 0, None, 0, guard, guard) != 0  # the resident calls want 1..max_batch
    assert b"kosk_verifiable_keygen_resident" in lib.kosk_last_error(h)
    assert guard.raw[:64] == b"\xa5" * 64  # nothing written
    # a negative count is an error, not a crash; the context stays usable afterwards
    assert lib.kosk_verify_batch(h, -1, guard, guard, guard) != 0
    assert b"kosk_verify_batch" in lib.kosk_last_error(h)
    tape = [oracle.tape_bytes_for(3, 0)]
    pks, sks, pis = ctx.verifiable_keygen(tape)
    assert ctx.verify(pis, pks) == [True]
    ctx.close()


@pytest.mark.parametrize("k", [2, 3])
def test_streamed_chunks_equal_single_context_and_oracle(k, oracle, torch_cuda, monkeypatch):
    """KOSK_STREAMS=3: a call longer than one sub-context is cut into chunks that run on three sub-contexts concurrently
    (caller buffers page-locked for the call); n = 7 over sub-batches of 2 leaves a ragged last chunk."""
    from mpcith_kyber_kosk_amd import api
    n = 7
    tapes = [oracle.tape_bytes_for(k, 40 + b) for b in range(n)]
    plain = api.Kosk(kyber_k=k, max_batch=n)
    pks0, sks0, pis0 = plain.verifiable_keygen(tapes)
    monkeypatch.setenv("KOSK_STREAMS", "3")
    st = api.Kosk(kyber_k=k, max_batch=6)
    monkeypatch.delenv("KOSK_STREAMS")
    assert st.streams == 3
    pks, sks, pis = st.verifiable_keygen(tapes)
    assert pks == pks0 and sks == sks0 and pis == pis0
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
    # the opt-out of page-locking gives the same bytes
    monkeypatch.setenv("KOSK_STREAMS", "3")
    monkeypatch.setenv("KOSK_REGISTER", "0")
    st2 = api.Kosk(kyber_k=k, max_batch=6)
    pks2, sks2, pis2 = st2.verifiable_keygen(tapes)
    assert pks2 == pks0 and sks2 == sks0 and pis2 == pis0
    assert st2.verify(bad, keys) == want
    for c in (plain, st, st2):
        c.close()


def test_batch_of_one_on_a_large_context_and_full_context(oracle, torch_cuda):
    """n = 1 on a context sized for 46 (the bench's slot) and n = max_batch exactly."""
    from mpcith_kyber_kosk_amd import api
    k = 3
    ctx = api.Kosk(kyber_k=k, max_batch=46)
    t = [oracle.tape_bytes_for(k, 7)]
    pks, sks, pis = ctx.verifiable_keygen(t)
    opk, osk, opi, _, _ = oracle.verifiable_keygen(k, t[0])
    assert (pks[0], sks[0], pis[0]) == (opk, osk, opi)
    assert ctx.verify(pis, pks) == [True]
    ctx.close()
    ctx = api.Kosk(kyber_k=k, max_batch=5)
    tapes = [oracle.tape_bytes_for(k, 60 + b) for b in range(5)]
    pks, sks, pis = ctx.verifiable_keygen(tapes)
    opk, osk, opi, _, _ = oracle.verifiable_keygen(k, tapes[4])
    assert (pks[4], sks[4], pis[4]) == (opk, osk, opi)
    assert ctx.verify(pis, pks) == [True] * 5
    ctx.close()


@pytest.mark.parametrize("k", [2, 3])
def test_degenerate_randomness_tapes(k, oracle, torch_cuda):
    """Tapes of constant bytes: every BE16 % q draw, every seed and every sampler sees its extreme input (0x0000 and 0xFFFF words,
    identical seeds for d, z and all PRF keys, all-equal share randoms).  Byte for byte against the oracle, and verified."""
    from mpcith_kyber_kosk_amd import api
    ctx = api.Kosk(kyber_k=k, max_batch=4)
    n = ctx.tape_bytes
    tapes = [b"\x00" * n, b"\xff" * n, (b"\x0d\x00" * n)[:n], bytes((i * 251 + 7) & 0xFF for i in range(n))]
    pks, sks, pis = ctx.verifiable_keygen(tapes)
    for b in range(len(tapes)):
        opk, osk, opi, _, _ = oracle.verifiable_keygen(k, tapes[b])
        assert pks[b] == opk and sks[b] == osk, b
        assert pis[b] == opi, b
    assert ctx.verify(pis, pks) == [True] * len(tapes)
    ctx.close()


def test_split_commitment_launches_give_identical_proofs(oracle, torch_cuda, monkeypatch):
    """KOSK_HASH_SPLIT=1 + KOSK_HASH_PRIMER=1 (opt-in): 46 proofs are hashed as 44 + 2 behind a placement primer.  Same bytes as
    the default single launch; the two proofs of the second launch also against the oracle."""
    from mpcith_kyber_kosk_amd import api
    k, n = 3, 46
    tapes = [oracle.tape_bytes_for(k, 300 + b) for b in range(n)]
    plain = api.Kosk(kyber_k=k, max_batch=n)
    assert plain.commit_launch_groups(n) == n
    ref = plain.verifiable_keygen(tapes)
    plain.close()
    monkeypatch.setenv("KOSK_HASH_SPLIT", "1")
    monkeypatch.setenv("KOSK_HASH_PRIMER", "1")
    ctx = api.Kosk(kyber_k=k, max_batch=n)
    assert ctx.commit_launch_groups(n) == 44 and ctx.commit_launch_groups(44) == 44 and ctx.commit_launch_groups(3) == 3
    got = ctx.verifiable_keygen(tapes)
    assert got == ref
    for b in (44, 45):
        opk, osk, opi, _, _ = oracle.verifiable_keygen(k, tapes[b])
        assert (got[0][b], got[1][b], got[2][b]) == (opk, osk, opi)
    assert ctx.verify(got[2], got[0]) == [True] * n
    ctx.close()


@pytest.mark.parametrize("knob", ["KOSK_TABLE_GEMM=0", "KOSK_HASH_DMA=0", "KOSK_LINCOMB_FUSED=0", "KOSK_NTT_FP32=1", "KOSK_BLOCKING_SYNC=1",
                                  "KOSK_GRAPHS=1"])
def test_documented_knobs_do_not_change_results(knob, oracle, torch_cuda, monkeypatch):
    """Every runtime knob of INTEGRATION.md 5 selects another kernel or another way of waiting, never other bytes: proofs, keys
    and verify bits equal the default context's (which the other tests pin to the oracle), for K = 3 and a K = 4 spot check."""
    from mpcith_kyber_kosk_amd import api
    name, val = knob.split("=")
    for k, n in ((3, 3), (4, 1)):
        tapes = [oracle.tape_bytes_for(k, 500 + b) for b in range(n)]
        base = api.Kosk(kyber_k=k, max_batch=n)
        ref = base.verifiable_keygen(tapes)
        base.close()
        monkeypatch.setenv(name, val)
        ctx = api.Kosk(kyber_k=k, max_batch=n)
        monkeypatch.delenv(name)
        got = ctx.verifiable_keygen(tapes)
        assert got == ref
        assert ctx.verify(got[2], got[0]) == [True] * n
        bad = bytearray(got[2][0]); bad[oracle.params(k).off[0] + 7] ^= 4  # an f share of an opened party: always read
        assert ctx.verify([bytes(bad)], [got[0][0]]) == [False]
        ctx.close()
