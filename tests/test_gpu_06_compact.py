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


@pytest.mark.parametrize("k", [2, 3, 4])
def test_compact_host_buffer_calls_chunked(k, oracle, torch_cuda):
    """kosk_verifiable_keygen_batch_compact / kosk_verify_batch_compact: n = 5 through a context of 2 (three chunks, the last
    ragged): the compact bytes equal the host codec of the image call's proofs; the verifier accepts them and rejects a
    tampered one with the same fail mask as the image path."""
    from mpcith_kyber_kosk_amd import api
    lib = api.lib
    n = 5
    tapes = [oracle.tape_bytes_for(k, 170 + i) for i in range(n)]
    ref = api.Kosk(kyber_k=k, max_batch=n)
    pks, sks, pis = ref.verifiable_keygen(tapes)
    cb = lib.kosk_compact_proof_bytes(k)
    ctx = api.Kosk(kyber_k=k, max_batch=2, fs_mode=k & 1)  # K = 3 with the Fiat-Shamir hashes on the device (`ref` above hashes on the host)
    pk = C.create_string_buffer(ctx.pk_bytes * n); sk = C.create_string_buffer(ctx.sk_bytes * n)
    out = C.create_string_buffer(cb * n); ok = C.create_string_buffer(n)
    assert lib.kosk_verifiable_keygen_batch_compact(ctx.handle, n, C.c_char_p(b"".join(tapes)), ctx.tape_bytes, pk, sk, out) == 0
    assert pk.raw == b"".join(pks) and sk.raw == b"".join(sks)
    for b in range(n):
        host = C.create_string_buffer(cb)
        assert lib.kosk_proof_compress(k, pis[b], host) == 0
        assert out.raw[b * cb:(b + 1) * cb] == host.raw, b
    assert lib.kosk_verify_batch_compact(ctx.handle, n, out, pk, ok) == 0 and ok.raw == b"\x01" * n
    bad = bytearray(out.raw); bad[4 * cb + 200] ^= 0x20
    assert lib.kosk_verify_batch_compact(ctx.handle, n, bytes(bad), pk, ok) == 0 and ok.raw == b"\x01" * 4 + b"\x00"
    masks = ctx.fail_masks(n)
    img = C.create_string_buffer(ctx.proof_bytes)
    assert lib.kosk_proof_decompress(k, bytes(bad[4 * cb:5 * cb]), img) == 0
    assert ref.verify(pis[:4] + [img.raw], pks) == [True] * 4 + [False] and ref.fail_masks(n) == masks
    assert lib.kosk_verify_batch_compact(ctx.handle, 0, out, pk, ok) == 0
    ref.close(); ctx.close()
