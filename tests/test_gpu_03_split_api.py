"""GPU tests of the reference's second-level entry points (mlwe_prover.hpp:77-99, mlwe_verifier.hpp:14-15) through the
C ABI: prepare_randomness / prepare_range_proof / prove / verify on the reference's struct layouts, in main.cpp's call
order, bit-exact against the oracle, and consistent with the fused kyber_verifiable_keygen path."""
import ctypes as C
import hashlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU: torch.cuda.is_available() is False")
    return torch


def _keygen_inst(api, k, seed64):
    A = np.zeros((k, k, 256), np.int16); s = np.zeros((k, 256), np.int16); e = np.zeros((k, 256), np.int16); t = np.zeros((k, 256), np.int16)
    pk = C.create_string_buffer(api.pk_bytes(k)); sk = C.create_string_buffer(api.sk_bytes(k))
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    assert api.lib.kosk_keygen(k, seed64, pk, sk, vp(A), vp(s), vp(e), vp(t)) == 0
    return A.tobytes() + t.tobytes() + s.tobytes() + e.tobytes(), pk.raw


@pytest.mark.parametrize("fs", [0, 1])  # Fiat-Shamir hashes on the host / on the device (kosk_options::fs_mode)
@pytest.mark.parametrize("k", [2, 3, 4])
def test_main_cpp_order_matches_oracle(oracle, torch_cuda, k, fs):
    """main.cpp:21-47: prepare_randomness, prepare_range_proof, kyber_keygen, prove, verify on ONE tape."""
    from mpcith_kyber_kosk_amd import api
    tape = oracle.tape_bytes_for(k, 31)
    ref = oracle.main_order(k, tape)
    assert ref["verify"]
    u = ref["used"]
    ctx = api.Kosk(kyber_k=k, max_batch=2, fs_mode=fs)
    assert api.lib.kosk_randomness_bytes(k) == len(ref["rand"])
    assert api.lib.kosk_range_proof_bytes(k) == len(ref["range"])
    assert api.lib.kosk_mlwe_inst_bytes(k) == len(ref["inst"])
    rand = ctx.prepare_randomness([tape[0:u[0]]])[0]
    assert rand == ref["rand"], "mpcith_randomness differs from the oracle"
    rng = ctx.prepare_range_proof([tape[u[0]:u[1]]])[0]
    assert rng == ref["range"], "mpcith_range_proof differs from the oracle"
    inst, pk = _keygen_inst(api, k, tape[u[1]:u[2]])
    assert inst == ref["inst"] and pk == ref["pk"]
    pi = ctx.prove_prepared([inst], [rand], [rng], [tape[u[2]:u[3]]])[0]
    assert pi == ref["pi"], "proof differs from the oracle's prove() on the same structs"
    assert ctx.verify_inst([pi], [inst]) == [True]
    assert ctx.verify([pi], [pk]) == [True]
    for pos in (10, (len(pi) // 3) & ~1):  # a checked share (f_shares) and one the reference never reads (a late beta share)
        bad = bytearray(pi); bad[pos] ^= 1   # low byte of a u16: stays canonical unless the value was q-1
        if int.from_bytes(bad[pos:pos + 2], "little") < 3329:
            assert ctx.verify_inst([bytes(bad)], [inst]) == [oracle.kosk_verify(k, bytes(bad), pk)[0]]
    bad = bytearray(pi); bad[10] ^= 1
    assert ctx.verify_inst([bytes(bad)], [inst]) == [False]
    other, _ = _keygen_inst(api, k, hashlib.sha3_512(b"another key").digest())
    assert ctx.verify_inst([pi], [other]) == [False]


def test_split_equals_fused(oracle, torch_cuda):
    """kyber_verifiable_keygen draws keygen's 64 bytes first; feeding the same bytes to the split calls gives the same proof."""
    from mpcith_kyber_kosk_amd import api
    k = 3
    p = oracle.params(k)
    ctx = api.Kosk(kyber_k=k, max_batch=3)
    tapes = [oracle.tape_bytes_for(k, 40 + i) for i in range(3)]
    pks, sks, pis = ctx.verifiable_keygen(tapes)
    a = 64 + 32 * p.M + 2 * p.M * 302
    b = a + 2 * k * p.E * 302
    rands = ctx.prepare_randomness([t[64:a] for t in tapes])
    rngs = ctx.prepare_range_proof([t[a:b] for t in tapes])
    insts = [_keygen_inst(api, k, t[:64])[0] for t in tapes]
    pis2 = ctx.prove_prepared(insts, rands, rngs, [t[b:] for t in tapes])
    assert pis2 == pis
    # the offline material can be banked and reused later on a fresh context (persisted offline/online split)
    ctx2 = api.Kosk(kyber_k=k, max_batch=1)
    assert ctx2.prove_prepared(insts[2:], rands[2:], rngs[2:], [tapes[2][b:]]) == pis[2:]
    assert ctx2.verify_inst(pis[2:], insts[2:]) == [True]


def test_split_randombytes_call_order(oracle, torch_cuda):
    """Without tapes the split calls draw through the callback with the reference's sizes: M x 32, 2M x 302 | 2K(2eta+1) x 302 | rest."""
    from mpcith_kyber_kosk_amd import api
    k = 2
    p = oracle.params(k)
    tape = oracle.tape_bytes_for(k, 50)
    ref = oracle.main_order(k, tape)
    u = ref["used"]
    ctx = api.Kosk(kyber_k=k, max_batch=1)
    calls = []
    stream = {"pos": 0, "buf": tape[:u[1]] + tape[u[2]:u[3]]}  # everything but keygen's 64 bytes

    def rb(n):
        calls.append(n)
        stream["pos"] += n
        return stream["buf"][stream["pos"] - n:stream["pos"]]
    ctx.set_randombytes(rb)
    rand = ctx.prepare_randomness(n=1)[0]
    assert calls == [32] * p.M + [302] * (2 * p.M)
    rng = ctx.prepare_range_proof(n=1)[0]
    assert calls[3 * p.M:] == [302] * (2 * k * p.E)
    n0 = len(calls)
    pi = ctx.prove_prepared([ref["inst"]], [rand], [rng])[0]
    assert calls[n0:] == [302] * (3 * k + 4 * k * p.eta1)
    assert pi == ref["pi"]
