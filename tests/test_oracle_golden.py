"""The oracle against the golden vectors recorded from the compiled reference."""
import hashlib
import json
import os

import numpy as np
import pytest

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "kosk_tape_v1.json")))


def test_tape_definition(oracle):
    t = oracle.tape_bytes_for(2, 0)
    assert t[:16].hex() == GOLD["reference"]["tape0_first16"]


@pytest.mark.parametrize("k", [2, 3, 4])
def test_oracle_matches_reference_digests(k, oracle):
    ref = GOLD["reference"][str(k)]
    p = oracle.params(k)
    assert p.proof_bytes == ref["proof_bytes"] and p.tape_bytes == ref["tape_bytes"] and p.tape_calls == ref["tape_calls"]
    pk, sk, pi, calls, pos = oracle.verifiable_keygen(k, oracle.tape_bytes_for(k, 0))
    assert (calls, pos) == (ref["tape_calls"], ref["tape_bytes"])
    assert hashlib.sha3_256(pk).hexdigest() == ref["sha3_pk"]
    assert hashlib.sha3_256(sk).hexdigest() == ref["sha3_sk"]
    assert hashlib.sha3_256(pi).hexdigest() == ref["sha3_pi"]
    I = np.frombuffer(pi[p.off[5]:p.off[5] + 300], dtype="<u2")
    assert list(I[:8]) == ref["I_first8"]
    assert len(set(I.tolist())) == 150 and I.max() < 1454
    # verify accepts, tampered rejects (survey: "verify / tampered" = 1 / 0)
    ok, why = oracle.kosk_verify(k, pi, pk)
    assert ok, why
    bad = bytearray(pi); bad[p.off[13] + 10] ^= 1
    ok, why = oracle.kosk_verify(k, bytes(bad), pk)
    assert not ok and "share error" in why
    # per-field digests (localise regressions)
    g = GOLD["oracle"]["k%d_tape0" % k]
    assert [p.off[i] for i in range(24)] == g["field_offsets"]
    assert [hashlib.sha3_256(pi[p.off[i]:p.off[i] + p.size[i]]).hexdigest()[:16] for i in range(24)] == g["fields_sha3"]


def test_second_tape_and_trace(oracle):
    k = 3
    pk, sk, pi, calls, pos, tr = oracle.verifiable_keygen(k, oracle.tape_bytes_for(k, 1), trace=True)
    g = GOLD["oracle"]["k3_tape1"]
    assert hashlib.sha3_256(pi).hexdigest() == g["sha3_pi"]
    assert bytes(tr.h1).hex() == g["h1"] and bytes(tr.ch).hex() == g["ch"] and list(tr.alpha)[:4] == g["alpha_first4"]
    # stage consistency: h1 is the digest of the Tcomm table
    assert hashlib.sha3_256(bytes(tr.tcomm)).digest() == bytes(tr.h1)
    assert hashlib.sha3_256(bytes(tr.view_digest)).digest() == bytes(tr.ch)


def test_k2_survey_pins_on_proof_fields(oracle):
    """pi.f_shares[0][0..3] = 518 1840 2941 965, pi.Tcomm[0][0..7], pi.comm[0][0..7] for K=2."""
    k = 2
    ref = GOLD["reference"]["2"]
    p = oracle.params(k)
    pk, sk, pi, _, _ = oracle.verifiable_keygen(k, oracle.tape_bytes_for(k, 0))
    assert list(np.frombuffer(pi[p.off[0]:p.off[0] + 8], dtype="<u2")) == ref["f_shares_0_first4"]
    assert pi[p.off[4]:p.off[4] + 8].hex() == ref["tcomm0_first8"]
    assert pi[p.off[23]:p.off[23] + 8].hex() == ref["comm0_first8"]
