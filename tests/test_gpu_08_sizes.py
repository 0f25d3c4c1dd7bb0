"""Parity at BASELINE.json's sizes: configs[1] (Kyber-512, 46 proofs = 66 884 party lanes) end to end, and the graded kernel
entry points at exactly 65 536 lanes / 65 536 polynomials for K = 2 and K = 3, checked on
sampled lanes against hashlib / the oracle -- the same helpers bench.py's 65 536-lane leg asserts with."""
import hashlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

LANES = 65536


@pytest.fixture(scope="module")
def torch_cuda():
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU: torch.cuda.is_available() is False")
    return torch


def bench_rows(k, words, lanes=LANES):
    """SURVEY.md 8(d) config 2/3 input: u16 values in [0, q) from the byte stream SHAKE256("kosk-bench-v1:k<K>")"""
    raw = hashlib.shake_256(("kosk-bench-v1:k%d" % k).encode()).digest(2 * words * lanes)
    return (np.frombuffer(raw, "<u2").astype(np.uint32) * 3329 >> 16).astype(np.uint16).reshape(words, lanes)


def sample_lanes(lanes, count=64):
    rng = np.random.default_rng(lanes)
    return sorted(set([0, 1, 63, 64, lanes - 65, lanes - 1] + rng.integers(0, lanes, count - 6).tolist()))


@pytest.mark.parametrize("k", [2, 3])
def test_commit_hash_at_65536_lanes(k, torch_cuda, oracle):
    """kosk_commit_hash_lanes at exactly 65 536 lanes: Tcomm messages (308 / 320 B), then view messages (452 / 472 B) whose
    first 32 bytes are the Tcomm digest of the same lane (mlwe_prover.cpp:116-127, :397-444; fips202.c:745-754)."""
    torch = torch_cuda
    from mpcith_kyber_kosk_amd import api
    p = oracle.params(k)
    tc_words, vw_words = p.tcomm_msg_bytes // 2, (p.view_msg_bytes - 32) // 2
    assert (p.tcomm_msg_bytes, p.view_msg_bytes) == {2: (308, 452), 3: (320, 472)}[k]
    rows = bench_rows(k, vw_words)
    d_rows = torch.from_numpy(rows.view(np.int16)).cuda()
    d_tc = torch.zeros((LANES, 32), dtype=torch.uint8, device="cuda")
    d_vw = torch.zeros((LANES, 32), dtype=torch.uint8, device="cuda")
    c = api.Kosk(kyber_k=k, max_batch=1)
    torch_cuda.cuda.synchronize()  # torch fills / copies run on the null stream; the library streams are not ordered against it
    c.commit_hash_lanes(d_rows.data_ptr(), LANES, LANES, 0, 0, d_tc.data_ptr())          # rows 0..tc_words-1
    torch_cuda.cuda.synchronize()  # torch fills / copies run on the null stream; the library streams are not ordered against it
    c.commit_hash_lanes(d_rows.data_ptr(), LANES, LANES, d_tc.data_ptr(), 1, d_vw.data_ptr())
    c.synchronize()
    assert c.path_counts()["hash_dma"] == 2          # the kernel the bench times (LDS-DMA staged), not a fallback
    tc, vw = d_tc.cpu().numpy(), d_vw.cpu().numpy()
    for l in sample_lanes(LANES):
        t = hashlib.sha3_256(rows[:tc_words, l].astype("<u2").tobytes()).digest()
        assert tc[l].tobytes() == t, l
        assert vw[l].tobytes() == hashlib.sha3_256(t + rows[:, l].astype("<u2").tobytes()).digest(), l
    # size-independent property: every digest differs from its neighbours' (no lane wrote another lane's slot)
    assert len({bytes(x) for x in vw[::257]}) == len(vw[::257])
    c.close()


def test_ntt256_at_65536_polynomials(torch_cuda, oracle):
    """kosk_ntt256_batch on 65 536 polynomials (ntt.c:80-95 + poly.c:261-265), sampled against the oracle's poly_ntt, plus
    linearity over the whole batch: NTT(a) + NTT(b) == NTT(a + b) mod q."""
    torch = torch_cuda
    from mpcith_kyber_kosk_amd import api
    c = api.Kosk(kyber_k=3, max_batch=1)
    a = bench_rows(3, 256).T.copy().astype(np.int16)   # [65536][256], values in [0, q)
    b = np.roll(a, 1, axis=0)
    s = ((a.astype(np.int32) + b) % 3329).astype(np.int16)
    outs = []
    for x in (a, b, s):
        d_in = torch.from_numpy(np.ascontiguousarray(x)).cuda()
        d_out = torch.zeros_like(d_in)
        torch_cuda.cuda.synchronize()  # torch fills / copies run on the null stream; the library streams are not ordered against it
        c.ntt256_batch(d_in.data_ptr(), d_out.data_ptr(), LANES)
        c.synchronize()
        outs.append(d_out.cpu().numpy())
    for i in sample_lanes(LANES):
        assert np.array_equal(outs[0][i], oracle.poly_ntt(a[i])), i
    assert outs[0].min() >= -1664 and outs[0].max() <= 1664
    lin = (outs[0].astype(np.int32) + outs[1] - outs[2]) % 3329
    assert not lin.any()
    c.close()


def test_config2_kyber512_46_proofs(oracle, torch_cuda):
    """BASELINE.json configs[1]: K = 2, 46 proofs = 66 884 party lanes in one batch.  Every verify bit, three proofs byte for
    byte against the oracle, the reference-recorded digest of proof 0 (tape "kosk-tape-v1:0")."""
    from mpcith_kyber_kosk_amd import api
    k, n = 2, 46
    ctx = api.Kosk(kyber_k=k, max_batch=n)
    tapes = [oracle.tape_bytes_for(k, b) for b in range(n)]
    ctx.verifiable_keygen_resident(tapes)
    pks, sks = ctx.keys(n)
    pis = ctx.fetch_proofs(n)
    assert hashlib.sha3_256(pis[0]).hexdigest() == "e8252bad44ae1e49bdb9e5f5d75bbabb98e2453a32909013c2b2aed8d425d330"
    assert hashlib.sha3_256(pks[0]).hexdigest() == "5303acc35b8f721f343bdfe43cafec16c69ca1ac5c5e79bf0379fc9d902508f2"
    assert hashlib.sha3_256(sks[0]).hexdigest() == "ae6d5d9158c3f86f990c6c7d397e9e91e00b73ee4960f2d4a402aaa97e4e6404"
    for b in (0, 22, 45):
        opk, osk, opi, _, _ = oracle.verifiable_keygen(k, tapes[b])
        assert pks[b] == opk and sks[b] == osk and pis[b] == opi, b
    assert ctx.verify_resident_pk(n) == [True] * n
    # the drop-in calls on host buffers give the same bytes and bits; one tampered proof in the middle is the only reject
    pks2, sks2, pis2 = ctx.verifiable_keygen(tapes)
    assert pks2 == pks and sks2 == sks and pis2 == pis
    bad = list(pis)
    t = bytearray(bad[31]); t[oracle.params(k).off[13] + 9] ^= 1; bad[31] = bytes(t)
    assert ctx.verify(bad, pks) == [i != 31 for i in range(n)]
    assert oracle.kosk_verify(k, bad[31], pks[31])[0] is False
    ctx.close()
