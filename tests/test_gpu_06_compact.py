"""Compact wire format: GPU pack / unpack against the host codec and the plain image path."""
import ctypes as C

import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU: torch.cuda.is_available() is False")
    return torch


@pytest.mark.parametrize("k", [2, 3, 4])
def test_compact_roundtrip_and_verify(oracle, torch_cuda, k):
    from mpcith_kyber_kosk_amd import api
    n = 3
    ctx = api.Kosk(kyber_k=k, max_batch=n)
    tapes = [oracle.tape_bytes_for(k, 70 + i) for i in range(n)]
    pks, sks, pis = ctx.verifiable_keygen(tapes)         # leaves the proofs resident
    cb = api.lib.kosk_compact_proof_bytes(k)
    assert cb % 16 == 0 and 0.77 * len(pis[0]) < cb < 0.79 * len(pis[0])
    blobs = ctx.fetch_proofs_compact(n)                  # packed on the GPU in front of the D2H copy
    for pi, blob in zip(pis, blobs):
        host = C.create_string_buffer(cb)
        assert api.lib.kosk_proof_compress(k, pi, host) == 0
        assert host.raw == blob, "GPU packing differs from the host codec"
        back = C.create_string_buffer(len(pi))
        assert api.lib.kosk_proof_decompress(k, blob, back) == 0
        assert back.raw == pi, "compact format is not lossless"
    # verifier fed with compact bytes: unpacked on the GPU behind the H2D copy
    ctx2 = api.Kosk(kyber_k=k, max_batch=n)
    ctx2.stage_verifier_inputs_compact(blobs, pks)
    assert ctx2.verify_resident(n) == [True] * n
    assert ctx2.fetch_proofs(n) == pis                   # the resident images are the original ones
    bad = bytearray(blobs[1]); bad[100] ^= 0x10
    ctx2.stage_verifier_inputs_compact([blobs[0], bytes(bad), blobs[2]], pks)
    assert ctx2.verify_resident(n) == [True, False, True]


def test_compress_rejects_unrepresentable_values(oracle, torch_cuda):
    from mpcith_kyber_kosk_amd import api
    k = 2
    p = oracle.params(k)
    img = bytearray(p.proof_bytes)
    out = C.create_string_buffer(api.lib.kosk_compact_proof_bytes(k))
    assert api.lib.kosk_proof_compress(k, bytes(img), out) == 0
    img[1] = 0x10                                        # first u16 = 4096
    assert api.lib.kosk_proof_compress(k, bytes(img), out) == -1
