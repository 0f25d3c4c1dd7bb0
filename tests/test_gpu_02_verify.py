"""GPU parity: HIP verifier verify bit vs the CPU oracle on honest and tampered proofs."""
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU: torch.cuda.is_available() is False")
    return torch


@pytest.mark.parametrize("k", [2, 3, 4])
def test_verify_honest_and_tampered(k, oracle, torch_cuda):
    from mpcith_kyber_kosk_amd import api
    p = oracle.params(k)
    ctx = api.Kosk(kyber_k=k, max_batch=13)
    tapes = [oracle.tape_bytes_for(k, b) for b in range(2)]
    pks, sks, pis = ctx.verifiable_keygen(tapes)
    assert ctx.verify(pis, pks) == [True, True]
    # the oracle accepts what the GPU proved, and the GPU accepts what the oracle proved
    opk, osk, opi, _, _ = oracle.verifiable_keygen(k, tapes[0])
    assert oracle.kosk_verify(k, pis[0], pks[0])[0]
    assert ctx.verify([opi], [opk]) == [True]
    # wrong public key
    assert ctx.verify([pis[0]], [pks[1]]) == [False]
    assert not oracle.kosk_verify(k, pis[0], pks[1])[0]

    # flip one bit in every one of the 24 proof fields (first, middle and last element)
    bad, where = [], []
    for f in range(24):
        for pos in (0, p.size[f] // 2, p.size[f] - 1):
            t = bytearray(pis[0])
            t[p.off[f] + pos] ^= 1
            bad.append(bytes(t))
            where.append((f, pos))
    got = ctx.verify(bad, [pks[0]] * len(bad))
    accepted = []
    for (f, pos), g, t in zip(where, got, bad):
        exp, why = oracle.kosk_verify(k, t, pks[0])
        assert g == exp, f"field {f} byte {pos}: gpu={g} oracle={exp} ({why})"
        if g:
            accepted.append(f)
    # The reference never looks at beta/gamma (fields 2,3), t (8) and eta (15,16) shares of unopened
    # parties beyond the first 407 (mlwe_verifier.cpp:106-107, :321-323, :390-394), nor at u shares
    # (21,22) of unopened parties beyond party 812 (:503-507, :555-556): flips there are accepted by the
    # reference and therefore, bit for bit, by this verifier.  Every other field rejects.
    assert set(accepted) <= {2, 3, 8, 15, 16, 21, 22}, accepted
    rejected_fields = {f for (f, _), g in zip(where, got) if not g}
    assert rejected_fields >= set(range(24)) - {8, 15, 16}, rejected_fields
    ctx.close()


def test_verify_malformed_opened_list(oracle, torch_cuda):
    from mpcith_kyber_kosk_amd import api
    k = 2
    p = oracle.params(k)
    ctx = api.Kosk(kyber_k=k, max_batch=4)
    tapes = [oracle.tape_bytes_for(k, 5)]
    pks, sks, pis = ctx.verifiable_keygen(tapes)
    off = p.off[5]
    dup = bytearray(pis[0]); dup[off + 2:off + 4] = dup[off:off + 2]          # I[1] = I[0]
    big = bytearray(pis[0]); big[off:off + 2] = (1454).to_bytes(2, "little")  # I[0] out of range
    noncanon = bytearray(pis[0])
    so = p.off[13]
    v = int.from_bytes(noncanon[so:so + 2], "little") + 3329                    # same residue, non-canonical
    noncanon[so:so + 2] = v.to_bytes(2, "little")
    got = ctx.verify([bytes(dup), bytes(big), bytes(noncanon), pis[0]], pks * 4)
    assert got == [False, False, False, True]
    assert not oracle.kosk_verify(k, bytes(noncanon), pks[0])[0]
    ctx.close()


BYTE_FIELDS = (4, 23)  # Tcomm, comm digests; field 5 is the list I; every other field holds u16 elements of GF(3329)


@pytest.mark.parametrize("k", [2, 3, 4])
def test_verify_random_corruptions_match_oracle(k, oracle, torch_cuda):
    """Differential test at random positions: single-bit flips and substitutions of whole elements, anywhere in the proof
    image, always leaving CANONICAL field elements (< q) behind.  Whatever the reference's verifier does not read (see
    test_verify_honest_and_tampered) must be accepted here too, everything else rejected -- the oracle decides, proof by proof;
    the fail masks are zero exactly for the accepted."""
    import random
    from mpcith_kyber_kosk_amd import api
    p = oracle.params(k)
    rnd = random.Random(20260 + k)
    ctx = api.Kosk(kyber_k=k, max_batch=16)
    tape = [oracle.tape_bytes_for(k, 90 + k)]
    pks, sks, pis = ctx.verifiable_keygen(tape)
    pi = pis[0]
    bad, what = [], []
    for _ in range(22):  # bit flips, weighted towards the large share fields where unread regions exist
        f = rnd.choice([2, 3, 8, 15, 16, 21, 22] * 2 + list(range(24)))
        t = bytearray(pi)
        if f in BYTE_FIELDS or f == 5:
            pos = rnd.randrange(p.size[f])
            t[p.off[f] + pos] ^= 1 << rnd.randrange(8)
        else:
            pos = rnd.randrange(p.size[f] // 2) * 2
            old = int.from_bytes(t[p.off[f] + pos:p.off[f] + pos + 2], "little")
            new = old ^ (1 << rnd.randrange(12))
            while new >= 3329:
                new = old ^ (1 << rnd.randrange(12))
            t[p.off[f] + pos:p.off[f] + pos + 2] = new.to_bytes(2, "little")
        bad.append(bytes(t)); what.append(("flip", f, pos))
    for _ in range(12):  # another canonical field element in place of a u16
        f = rnd.choice([0, 1, 2, 3, 8, 9, 10, 15, 16, 17, 19, 21, 22])
        pos = rnd.randrange(p.size[f] // 2) * 2
        t = bytearray(pi)
        old = int.from_bytes(t[p.off[f] + pos:p.off[f] + pos + 2], "little")
        new = (old + 1 + rnd.randrange(3328)) % 3329
        t[p.off[f] + pos:p.off[f] + pos + 2] = new.to_bytes(2, "little")
        bad.append(bytes(t)); what.append(("subst", f, pos))
    got = ctx.verify(bad, [pks[0]] * len(bad))
    masks = ctx.fail_masks(len(bad))
    n_acc = 0
    for w, g, m, t in zip(what, got, masks, bad):
        exp, why = oracle.kosk_verify(k, t, pks[0])
        assert g == exp, f"{w}: gpu={g} oracle={exp} ({why})"
        assert (m == 0) == g, f"{w}: verify bit {g} but fail mask {m:#x}"
        n_acc += g
    assert 0 < n_acc < len(bad), n_acc  # the sample holds both kinds
    ctx.close()


def test_verify_non_canonical_elements(oracle, torch_cuda):
    """u16 values >= q never come out of an honest prover.  Where the reference never reads a record, any bytes are accepted,
    by the reference, the oracle and this verifier alike.  Where it does read, this verifier rejects the proof outright (fail
    bit 0) -- the reference computes on the out-of-range value with gf3329's non-reducing add/sub and rejects or (for some
    congruent encodings) accepts; the cases below are ones it rejects as well."""
    from mpcith_kyber_kosk_amd import api
    k = 3
    p = oracle.params(k)
    ctx = api.Kosk(kyber_k=k, max_batch=8)
    pks, sks, pis = ctx.verifiable_keygen([oracle.tape_bytes_for(k, 77)])
    pi = pis[0]
    I = [int.from_bytes(pi[p.off[5] + 2 * i:p.off[5] + 2 * i + 2], "little") for i in range(150)]
    rest = [q for q in range(1454) if q not in set(I)]
    K, E, Z, NCHK = k, 5, 4, 70

    def put(f, idx, val):
        t = bytearray(pi)
        t[p.off[f] + 2 * idx:p.off[f] + 2 * idx + 2] = val.to_bytes(2, "little")
        return bytes(t)
    i_hi = next(i for i, q in enumerate(rest) if q >= 407)       # first unopened party with index >= 407
    i_lo = next(i for i, q in enumerate(rest) if q < 407)
    unread = [put(2, i_hi * NCHK + 3, 0xFFFF),                   # beta share of a party recon_secrets_ddeg never reaches
              put(3, (len(rest) - 1) * NCHK, 4096),              # gamma, last unopened party
              put(8, 407 * K, 3329),                             # t share of the 408th unopened party (not a node)
              put(15, (1000 * K + 1) * E + 2, 65535),            # eta share beyond the 407 nodes
              put(21, (813 * K) * Z, 40000),                     # u share of the 814th unopened party
              put(22, ((len(rest) - 1) * K + 2) * Z + 3, 3329)]
    read = [put(2, i_lo * NCHK + 3, 3329 + 5),                   # beta share of an unopened party below 407
            put(13, 5 * K, int.from_bytes(pi[p.off[13] + 10 * K:p.off[13] + 10 * K + 2], "little") + 3329),  # s + r share, compared raw
            put(8, 406 * K, 5000),                               # t share of the last node
            put(21, (812 * K) * Z, 3329)]                        # u share of the last node
    got = ctx.verify(unread + read, [pks[0]] * 10)
    masks = ctx.fail_masks(10)
    assert got == [True] * 6 + [False] * 4, got
    assert all(m == 0 for m in masks[:6]) and all(m & 1 for m in masks[6:]), [hex(m) for m in masks]
    for t in unread:
        assert oracle.kosk_verify(k, t, pks[0])[0]
    assert not oracle.kosk_verify(k, read[1], pks[0])[0]
    ctx.close()
