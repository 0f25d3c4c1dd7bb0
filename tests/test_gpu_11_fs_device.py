"""Fiat-Shamir aggregation on the GPU (round 6; csrc/kosk_fs_kernels.hip, csrc/kosk_fs_dev.hpp): the wave-cooperative sponge against
hashlib, the two derivations against the library's host functions, and handles in device mode (kosk_options::fs_mode) against the
oracle and against host-mode handles, byte for byte.  mlwe_prover.cpp:130-153, :445-474; mlwe_verifier.cpp:37-65, :634-683."""
import ctypes as C
import hashlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

NPARTY, NOPEN, NREST, SEL = 1454, 150, 1304, 1312
SEL_WIN, SEL_OSORT, SEL_OPOS, NWIN = 160, 192, 352, 23


@pytest.fixture(scope="module")
def torch():
    t = pytest.importorskip("torch")
    if not t.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU: torch.cuda.is_available() is False")
    return t


@pytest.fixture(scope="module")
def ctx(torch):
    from mpcith_kyber_kosk_amd import api
    c = api.Kosk(kyber_k=3, max_batch=8)
    yield c
    c.close()


def _dev(torch, arr):
    return torch.from_numpy(np.ascontiguousarray(arr)).cuda()


@pytest.mark.parametrize("length", [0, 1, 7, 8, 9, 16, 135, 136, 137, 143, 144, 271, 272, 273, 1000, 4000, 46528])
def test_wave_sponge_sha3_256_matches_hashlib(length, torch, ctx):
    """kosk_sha3_256_batch_wave: one Keccak state per wave (a 32-bit word of the bit-interleaved state per lane, theta / pi / chi through
    LDS), message lengths around every block and word boundary and the digest table's 46 528 bytes; ragged message counts"""
    rng = np.random.default_rng(7000 + length)
    stride = max(8, (length + 7) // 8 * 8) + 8
    for n in (1, 5, 67):
        msgs = rng.integers(0, 256, size=(n, stride), dtype=np.uint8)
        d_in = _dev(torch, msgs)
        d_out = torch.zeros((n + 1, 32), dtype=torch.uint8, device="cuda")  # one guard row behind the last digest
        torch.cuda.synchronize()  # torch fills / copies run on the null stream; the library streams are not ordered against it
        ctx.sha3_256_batch_wave(d_in.data_ptr(), stride, length, d_out.data_ptr(), n)
        ctx.synchronize()
        out = d_out.cpu().numpy()
        for i in range(n):
            assert out[i].tobytes() == hashlib.sha3_256(msgs[i, :length].tobytes()).digest(), (length, n, i)
        assert not out[n].any()


def test_wave_sponge_refuses_unaligned_input(torch, ctx):
    from mpcith_kyber_kosk_amd import api
    d_in = torch.zeros(4096, dtype=torch.uint8, device="cuda")
    d_out = torch.zeros(64, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    with pytest.raises(api.KoskError):
        ctx.sha3_256_batch_wave(d_in.data_ptr() + 4, 136, 100, d_out.data_ptr(), 1)
    with pytest.raises(api.KoskError):
        ctx.sha3_256_batch_wave(d_in.data_ptr(), 132, 100, d_out.data_ptr(), 2)


@pytest.mark.parametrize("k", [2, 3, 4])
def test_fs_alpha_device_matches_host(k, torch):
    """k_fs_chain<FS_ALPHA> on random digest tables against kosk_fs_alpha (the host function the default mode runs) and hashlib"""
    from mpcith_kyber_kosk_amd import api
    c = api.Kosk(kyber_k=k, max_batch=2)
    n, J = 37, 70 + 2 * k
    rng = np.random.default_rng(7100 + k)
    tables = rng.integers(0, 256, size=(n, NPARTY * 32), dtype=np.uint8)
    d_t = _dev(torch, tables)
    d_a = torch.full((n, 80), -1, dtype=torch.int16, device="cuda")  # 0xFFFF everywhere: the entries behind J must be zeroed
    d_h = torch.zeros((n, 32), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    c.fs_alpha_device(d_t.data_ptr(), NPARTY * 32, n, d_a.data_ptr(), d_h.data_ptr())
    c.synchronize()
    got = d_a.cpu().numpy().view(np.uint16)
    h1 = d_h.cpu().numpy()
    for b in range(n):
        want = (C.c_uint16 * 80)()
        assert api.lib.kosk_fs_alpha(k, tables[b].tobytes(), want) == 0
        assert list(got[b, :J]) == list(want[:J]), b
        assert not got[b, J:].any()
        assert h1[b].tobytes() == hashlib.sha3_256(tables[b].tobytes()).digest()
        prf = hashlib.shake_256(h1[b].tobytes() + b"\x01").digest(2 * J)
        assert list(got[b, :J]) == [((prf[2 * i] << 8) | prf[2 * i + 1]) % 3329 for i in range(J)]
    c.close()


def test_fs_opened_device_matches_host(torch, ctx):
    """k_fs_chain<FS_OPENED>: I with the reference's "+inc, rescan" probing (every random table has a handful of colliding candidates),
    the ascending complement, the window boundaries and the sorted opened list, against kosk_fs_opened and a restatement of
    fs_opened_batch's derived tables"""
    from mpcith_kyber_kosk_amd import api
    n = 96
    rng = np.random.default_rng(7200)
    tables = rng.integers(0, 256, size=(n, NPARTY * 32), dtype=np.uint8)
    d_t = _dev(torch, tables)
    d_sel = torch.zeros((n, SEL), dtype=torch.int16, device="cuda")
    d_rest = torch.zeros((n, SEL), dtype=torch.int16, device="cuda")
    d_ch = torch.zeros((n, 32), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    ctx.fs_opened_device(d_t.data_ptr(), NPARTY * 32, n, d_sel.data_ptr(), d_rest.data_ptr(), SEL, d_ch.data_ptr())
    ctx.synchronize()
    sel, rest, ch = d_sel.cpu().numpy().view(np.uint16), d_rest.cpu().numpy().view(np.uint16), d_ch.cpu().numpy()
    collided = 0
    for b in range(n):
        I = (C.c_uint16 * NOPEN)()
        R = (C.c_uint16 * NREST)()
        assert api.lib.kosk_fs_opened(tables[b].tobytes(), I, R) == 0
        assert list(sel[b, :NOPEN]) == list(I), b
        assert list(rest[b, :NREST]) == list(R), b
        assert ch[b].tobytes() == hashlib.sha3_256(tables[b].tobytes()).digest()
        prf = hashlib.shake_256(ch[b].tobytes() + b"\x01").digest(300)
        cand = [((prf[2 * i] << 8) | prf[2 * i + 1]) % NPARTY for i in range(NOPEN)]
        collided += cand != list(I)
        assert len(set(I)) == NOPEN
        win = [sum(1 for q in R if q < 64 * w) for w in range(NWIN + 1)]
        assert list(sel[b, SEL_WIN:SEL_WIN + NWIN + 1]) == win, b
        osort = sorted(I)
        pos = {q: i for i, q in enumerate(I)}
        assert list(sel[b, SEL_OSORT:SEL_OSORT + NOPEN]) == osort, b
        assert list(sel[b, SEL_OPOS:SEL_OPOS + NOPEN]) == [pos[q] for q in osort], b
    assert collided > n // 2, collided  # the probing ran: most tables have colliding candidates


@pytest.mark.parametrize("k", [2, 3, 4])
def test_device_fiat_shamir_handle_matches_oracle(k, oracle, torch):
    """a handle created with fs_mode = KOSK_FS_DEVICE: pk / sk / proof images byte-identical to the oracle's (and so to a host-mode
    handle's), verify bits, a tampered opened list caught by the device's own I' == I (fail bit 11), no digest table copied"""
    from mpcith_kyber_kosk_amd import api
    n = 5
    c = api.Kosk(kyber_k=k, max_batch=n, fs_mode=api.FS_DEVICE)
    tapes = [oracle.tape_bytes_for(k, 700 + 10 * k + b) for b in range(n)]
    pks, sks, pis = c.verifiable_keygen(tapes)
    p = oracle.params(k)
    for b in range(n):
        opk, osk, opi, _, _ = oracle.verifiable_keygen(k, tapes[b])
        assert pks[b] == opk and sks[b] == osk
        if pis[b] != opi:
            bad = [i for i in range(24) if pis[b][p.off[i]:p.off[i] + p.size[i]] != opi[p.off[i]:p.off[i] + p.size[i]]]
            pytest.fail(f"proof {b} differs from the oracle in fields {bad}")
    assert c.verify(pis, pks) == [True] * n
    assert c.fail_masks(n) == [0] * n
    # the opened list swapped / a view commitment flipped / a Tcomm digest flipped: rejected, with the bits the host mode reports
    t1 = bytearray(pis[1]); o = p.off[5]; t1[o:o + 2], t1[o + 2:o + 4] = t1[o + 2:o + 4], t1[o:o + 2]
    t2 = bytearray(pis[2]); t2[p.off[23] + 40] ^= 1
    t3 = bytearray(pis[3]); t3[p.off[4] + 5] ^= 0x80
    bad = [pis[0], bytes(t1), bytes(t2), bytes(t3), pis[4]]
    host = api.Kosk(kyber_k=k, max_batch=n, fs_mode=api.FS_HOST)
    assert host.verify(bad, pks) == [True, False, False, False, True]
    want_masks = host.fail_masks(n)
    assert c.verify(bad, pks) == [True, False, False, False, True]
    assert c.fail_masks(n) == want_masks
    assert want_masks[2] & (1 << 11)
    pc, ph = c.path_counts(), host.path_counts()
    assert pc["fs_device"] > 0 and pc["fs_host"] == 0 and pc["digest_copy"] == 0, pc
    assert ph["fs_host"] > 0 and ph["fs_device"] == 0, ph
    host.close()
    c.close()


def test_device_fiat_shamir_resident_calls_and_hooks(oracle, torch):
    """the resident pair in device mode: resident digest tables equal the oracle's, the round hook still fires four times per step with
    complete tables (the host waits for a table only when somebody asked to be told)"""
    from mpcith_kyber_kosk_amd import api
    k, n = 3, 4
    c = api.Kosk(kyber_k=k, max_batch=n, fs_mode=api.FS_DEVICE)
    tapes = [oracle.tape_bytes_for(k, 760 + b) for b in range(n)]
    seen = []

    def hook(role, rnd, ptr, nbytes):
        t = torch.as_tensor(api.DeviceView(ptr, (n, NPARTY, 32)), device="cuda")
        seen.append((role, rnd, nbytes, hashlib.sha3_256(t.cpu().numpy().tobytes()).hexdigest()))
    c.set_round_hook(hook)
    c.verifiable_keygen_resident(tapes)
    assert c.verify_resident_pk(n) == [True] * n
    c.set_round_hook(None)
    assert [(s[0], s[1], s[2]) for s in seen] == [(0, 0, n * NPARTY * 32), (0, 1, n * NPARTY * 32), (1, 0, n * NPARTY * 32), (1, 1, n * NPARTY * 32)]
    assert seen[0][3] == seen[2][3] and seen[1][3] == seen[3][3]  # the verifier rebuilt the prover's tables
    tabs = [torch.as_tensor(c.resident_digests(r, n), device="cuda").cpu().numpy() for r in (0, 1)]
    for b in (0, n - 1):
        opk, osk, opi, _, _, tr = oracle.verifiable_keygen(k, tapes[b], trace=True)
        assert tabs[0][b].tobytes() == bytes(tr.tcomm) and tabs[1][b].tobytes() == bytes(tr.view_digest)
    assert c.fetch_proofs(n)[n - 1] == opi
    c.close()


@pytest.mark.parametrize("callers", [3, 6])
def test_line_of_record_shape_with_device_fiat_shamir(callers, torch, gpu_child):
    """tests/gpu_child_cases.py: line_of_record_shape with the cohort in device mode -- merged runs of 138 / 276 proofs, every caller's
    pk / sk / images / digest tables equal a HOST-mode handle's and the oracle's"""
    out = gpu_child("from tests.gpu_child_cases import line_of_record_shape; line_of_record_shape(callers=%d, fs_device=1)" % callers)
    assert "line_of_record_shape ok 3 46 %d callers per run %d.00 fs_device" % (callers, callers) in out


def test_sixteen_callers_per_cohort_in_device_mode(torch, gpu_child):
    """the arrangement bench.py runs in device mode since round 6 (three cohorts of sixteen): ONE cohort of sixteen callers, 736 proofs per
    launch -- every caller's pk / sk (assembled in one host job over the whole run) / images / digest tables equal a host-mode handle's and
    the oracle's"""
    out = gpu_child("from tests.gpu_child_cases import line_of_record_shape; line_of_record_shape(callers=16, fs_device=1, rounds=2)", timeout=900)
    assert "line_of_record_shape ok 3 46 16 callers per run 16.00 fs_device" in out
