"""Oracle primitives against oracle/_ref: the reference's OWN kyber/*.c and utils/gf3329.c compiled in
place (oracle/Makefile).  Skipped when that build is absent (it cannot be rebuilt without /root/reference,
but the prebuilt .so travels to the GPU box)."""
import ctypes as C
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAMES = {2: "pqcrystals_kyber512_ref_", 3: "pqcrystals_kyber768_ref_", 4: "pqcrystals_kyber1024_ref_"}


def _ref(k):
    path = os.path.join(ROOT, "oracle", "_ref", "libkyber_ref_k%d.so" % k)
    if not os.path.exists(path):
        pytest.skip("oracle/_ref not built (reference tree absent)")
    return C.CDLL(path)


@pytest.mark.parametrize("k", [2, 3, 4])
def test_ring_arithmetic_and_sampling(k, oracle):
    ref, pre = _ref(k), NAMES[k]
    rng = np.random.default_rng(k)
    vp = C.c_void_p
    for trial in range(20):
        a = rng.integers(-3328, 3329, size=256).astype(np.int16)
        r1, r2 = a.copy(), a.copy()
        getattr(ref, pre + "poly_ntt")(r1.ctypes.data_as(vp))
        oracle.lib.ko_poly_ntt(r2.ctypes.data_as(vp))
        assert np.array_equal(r1, r2)
        r1, r2 = a.copy(), a.copy()
        getattr(ref, pre + "ntt")(r1.ctypes.data_as(vp))
        oracle.lib.ko_ntt(r2.ctypes.data_as(vp))
        assert np.array_equal(r1, r2)
        r1, r2 = a.copy(), a.copy()
        getattr(ref, pre + "poly_tomont")(r1.ctypes.data_as(vp))
        oracle.lib.ko_poly_tomont(r2.ctypes.data_as(vp))
        assert np.array_equal(r1, r2)
        A = rng.integers(0, 3329, size=k * 256).astype(np.int16)
        B = rng.integers(-1664, 1665, size=k * 256).astype(np.int16)
        o1, o2 = np.zeros(256, np.int16), np.zeros(256, np.int16)
        getattr(ref, pre + "polyvec_basemul_acc_montgomery")(o1.ctypes.data_as(vp), A.ctypes.data_as(vp), B.ctypes.data_as(vp))
        oracle.lib.ko_polyvec_basemul_acc(o2.ctypes.data_as(vp), A.ctypes.data_as(vp), B.ctypes.data_as(vp), k)
        assert np.array_equal(o1, o2)
        t = rng.integers(-3328, 3329, size=256).astype(np.int16)
        b1, b2 = C.create_string_buffer(384), C.create_string_buffer(384)
        getattr(ref, pre + "poly_tobytes")(b1, t.ctypes.data_as(vp))
        oracle.lib.ko_poly_tobytes(b2, t.ctypes.data_as(vp))
        assert b1.raw == b2.raw
        p1, p2 = np.zeros(256, np.int16), np.zeros(256, np.int16)
        getattr(ref, pre + "poly_frombytes")(p1.ctypes.data_as(vp), b1)
        oracle.lib.ko_poly_frombytes(p2.ctypes.data_as(vp), b1)
        assert np.array_equal(p1, p2)
        seed = bytes(rng.integers(0, 256, size=32, dtype=np.uint8))
        eta1 = 3 if k == 2 else 2
        n1 = np.zeros(256, np.int16)
        getattr(ref, pre + "poly_getnoise_eta1")(n1.ctypes.data_as(vp), C.c_char_p(seed), C.c_uint8(trial))
        prf = oracle.shake256(seed + bytes([trial]), eta1 * 64)
        n2 = np.zeros(256, np.int16)
        oracle.lib.ko_cbd(n2.ctypes.data_as(vp), C.c_char_p(prf), eta1)
        assert np.array_equal(n1, n2)
        for transposed in (0, 1):
            m1, m2 = np.zeros(k * k * 256, np.int16), np.zeros(k * k * 256, np.int16)
            getattr(ref, pre + "gen_matrix")(m1.ctypes.data_as(vp), C.c_char_p(seed), transposed)
            oracle.lib.ko_gen_matrix(m2.ctypes.data_as(vp), C.c_char_p(seed), transposed, k)
            assert np.array_equal(m1, m2)


def test_fips202_and_gf3329(oracle):
    ref = _ref(2)
    rng = np.random.default_rng(9)
    for n in (0, 1, 33, 135, 136, 137, 500, 46528):
        d = bytes(rng.integers(0, 256, size=n, dtype=np.uint8))
        o1 = C.create_string_buffer(32)
        ref.pqcrystals_kyber_fips202_ref_sha3_256(o1, C.c_char_p(d), C.c_size_t(n))
        assert o1.raw == oracle.sha3_256(d)
        o1 = C.create_string_buffer(333)
        ref.pqcrystals_kyber_fips202_ref_shake256(o1, C.c_size_t(333), C.c_char_p(d), C.c_size_t(n))
        assert o1.raw == oracle.shake256(d, 333)
    for f in ("gf3329_add", "gf3329_sub", "gf3329_mul"):
        getattr(ref, f).restype = C.c_uint16
        getattr(ref, f).argtypes = [C.c_uint16, C.c_uint16]
    ref.gf3329_inv.restype = C.c_uint16
    ref.gf3329_inv.argtypes = [C.c_uint16]
    ref.encode_to_gf3329.restype = C.c_uint16
    ref.encode_to_gf3329.argtypes = [C.c_int16]
    ref.decode_from_gf3329.restype = C.c_int16
    ref.decode_from_gf3329.argtypes = [C.c_uint16]
    oracle.lib.ko_gf_encode.restype = C.c_uint16
    oracle.lib.ko_gf_encode.argtypes = [C.c_int16]
    oracle.lib.ko_gf_decode.restype = C.c_int16
    oracle.lib.ko_gf_decode.argtypes = [C.c_uint16]
    for a in list(range(0, 3329, 13)) + [3328]:
        assert ref.gf3329_inv(a) == oracle.lib.ko_gf_inv(a)
        assert ref.decode_from_gf3329(a) == oracle.lib.ko_gf_decode(a)
        for b in (0, 1, 1664, 1665, 3328, (a * 5 + 1) % 3329):
            assert ref.gf3329_add(a, b) == oracle.lib.ko_gf_add(a, b)
            assert ref.gf3329_sub(a, b) == oracle.lib.ko_gf_sub(a, b)
            assert ref.gf3329_mul(a, b) == oracle.lib.ko_gf_mul(a, b)
    for a in range(-1664, 1665, 7):
        assert ref.encode_to_gf3329(a) == oracle.lib.ko_gf_encode(a)
