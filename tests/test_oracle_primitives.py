"""Known-answer tests of the oracle's primitives that do not depend on the reference at all:
hashlib, exhaustive field arithmetic, schoolbook ring products, mathematical table properties."""
import ctypes as C
import hashlib

import numpy as np
import pytest

Q = 3329


@pytest.mark.parametrize("n", [0, 1, 32, 33, 71, 72, 73, 135, 136, 137, 167, 168, 169, 272, 1000, 46528])
def test_sha3_shake_vs_hashlib(n, oracle):
    d = bytes((i * 131 + 7) & 255 for i in range(n))
    assert oracle.sha3_256(d) == hashlib.sha3_256(d).digest()
    assert oracle.shake256(d, 512) == hashlib.shake_256(d).digest(512)
    out = C.create_string_buffer(64)
    oracle.lib.ko_sha3_512(out, C.c_char_p(d), n)
    assert out.raw == hashlib.sha3_512(d).digest()
    out = C.create_string_buffer(700)
    oracle.lib.ko_shake128(out, 700, C.c_char_p(d), n)
    assert out.raw == hashlib.shake_128(d).digest(700)


def test_gf3329_exhaustive(oracle):
    a = np.arange(Q, dtype=np.int64)
    A, B = np.meshgrid(a, a, indexing="ij")
    # sample rows exhaustively in b for a stride of a (full 3329^2 through ctypes would be slow)
    for x in list(range(0, Q, 97)) + [1, Q - 1]:
        for y in (0, 1, 2, 1664, 1665, Q - 1, (x * 7 + 3) % Q):
            assert oracle.lib.ko_gf_add(x, y) == (x + y) % Q
            assert oracle.lib.ko_gf_sub(x, y) == (x - y) % Q
            assert oracle.lib.ko_gf_mul(x, y) == (x * y) % Q
    for x in range(1, Q):
        assert oracle.lib.ko_gf_mul(x, oracle.lib.ko_gf_inv(x)) == 1
    assert oracle.lib.ko_gf_inv(0) == 0


def test_barrett_is_centred_on_all_int16(oracle):
    for a in range(-32768, 32768):
        r = oracle.lib.ko_barrett_reduce(a)
        assert -1664 <= r <= 1664 and (r - a) % Q == 0


def test_montgomery_reduce(oracle):
    rng = np.random.default_rng(1)
    rinv = pow(1 << 16, -1, Q)
    for a in rng.integers(-Q * 32768, Q * 32768, size=2000):
        r = oracle.lib.ko_montgomery_reduce(int(a))
        assert -Q < r < Q and (r - int(a) * rinv) % Q == 0


def _bitrev7(i):
    return int(format(i, "07b")[::-1], 2)


def test_zetas_and_ntt_vs_definition(oracle):
    z = [oracle.lib.ko_zetas()[i] for i in range(128)]
    for i in range(128):
        assert (z[i] - pow(17, _bitrev7(i), Q) * (1 << 16)) % Q == 0 and -1664 <= z[i] <= 1664
    # NTT definition: f^[2i] + f^[2i+1] X = f mod (X^2 - 17^(2 brv(i) + 1))
    rng = np.random.default_rng(2)
    f = rng.integers(0, Q, size=256)
    out = oracle.poly_ntt(f.astype(np.int16))
    assert out.min() >= -1664 and out.max() <= 1664
    for i in (0, 1, 17, 63, 127):
        root = pow(17, 2 * _bitrev7(i) + 1, Q)
        c0 = sum(int(f[2 * j]) * pow(root, j, Q) for j in range(128)) % Q
        c1 = sum(int(f[2 * j + 1]) * pow(root, j, Q) for j in range(128)) % Q
        assert (int(out[2 * i]) - c0) % Q == 0 and (int(out[2 * i + 1]) - c1) % Q == 0


def test_basemul_is_negacyclic_product(oracle):
    """NTT^-1 is not on the path; check instead that pointwise basemul of two NTTs equals the NTT of the
    schoolbook product in Z_q[X]/(X^256+1), up to the Montgomery factors of the reference pipeline."""
    rng = np.random.default_rng(3)
    a = rng.integers(0, Q, size=256)
    b = rng.integers(-2, 3, size=256)
    prod = np.zeros(256, dtype=np.int64)
    for i in range(256):
        for j in range(256):
            if i + j < 256:
                prod[i + j] += int(a[i]) * int(b[j])
            else:
                prod[i + j - 256] -= int(a[i]) * int(b[j])
    prod %= Q
    A = oracle.poly_ntt(a.astype(np.int16))
    B = oracle.poly_ntt(b.astype(np.int16))
    r = np.zeros(256, np.int16)
    Aenc = np.where(A < 0, A + Q, A).astype(np.int16)
    oracle.lib.ko_polyvec_basemul_acc(r.ctypes.data_as(C.c_void_p), Aenc.ctypes.data_as(C.c_void_p), B.ctypes.data_as(C.c_void_p), 1)
    oracle.lib.ko_poly_tomont(r.ctypes.data_as(C.c_void_p))  # basemul leaves R^-1, tomont restores
    P = oracle.poly_ntt(prod.astype(np.int16))
    assert np.all((r.astype(np.int64) - P.astype(np.int64)) % Q == 0)


def test_lagrange_tables(oracle):
    from tests.test_oracle_golden import GOLD
    pins = GOLD["reference"]["lagrange_tables_sha256_prefix_suffix"]
    for which, name in enumerate(["share_ddeg", "recon_ddeg", "recon_2ddeg"]):
        t = oracle.table(which)
        h = hashlib.sha256(t.astype("<u2").tobytes()).hexdigest()
        assert [h[:8], h[-8:]] == pins[name]
        assert np.all(t.astype(np.int64).sum(axis=1) % Q == 1)  # Lagrange basis sums to 1
    assert list(oracle.table(0)[0][:4]) == GOLD["reference"]["share_ddeg_row0_first4"]
    # independent check of one coefficient by the product formula
    t = oracle.table(0)
    x, j = 5, 9
    num, den = 1, 1
    for m in range(407):
        if m != j:
            num = num * ((407 + x) - m) % Q
            den = den * (j - m) % Q
    assert t[x][j] == num * pow(den, -1, Q) % Q


def test_sharing_round_trips(oracle):
    rng = np.random.default_rng(4)
    y = rng.integers(0, Q, size=407).astype(np.uint16)
    sh = oracle.recompute_shares(y)
    assert np.array_equal(sh[:151], y[256:])                    # parties 0..150 hold the random tail
    assert np.array_equal(oracle.recon(sh), y[:256])            # encode -> reconstruct
    y2 = rng.integers(0, Q, size=407).astype(np.uint16)
    sh2 = oracle.recompute_shares(y2)
    prod = (sh.astype(np.uint32) * sh2 % Q).astype(np.uint16)
    assert np.array_equal(oracle.recon(prod, True), (y[:256].astype(np.uint32) * y2[:256] % Q).astype(np.uint16))
    # linearity
    s3 = oracle.recompute_shares(((y.astype(np.uint32) + y2) % Q).astype(np.uint16))
    assert np.array_equal(s3, ((sh.astype(np.uint32) + sh2) % Q).astype(np.uint16))
    # interpolation through arbitrary nodes reproduces the polynomial (stands in for NTL)
    xs = np.sort(rng.choice(np.arange(256, 1710), size=407, replace=False)).astype(np.uint16)
    full = np.concatenate([y[:256], sh])                         # values at points 0..1709
    ys = full[xs]
    out = np.zeros(407, np.uint16)
    oracle.lib.ko_interp_eval(out.ctypes.data_as(C.c_void_p), 407, xs.ctypes.data_as(C.c_void_p), ys.ctypes.data_as(C.c_void_p), 407)
    assert np.array_equal(out, full[:407])


def test_oracle_main_cpp_call_order(oracle):
    """main.cpp:21-47 order (prepare_randomness, prepare_range_proof, kyber_keygen, prove, verify): the whole tape is
    consumed in the documented pieces, the proof verifies, and the struct images have the reference's sizes."""
    for k, sizes in ((2, (950400, 163072, 5120)), (3, (975744, 174720, 9216))):
        p = oracle.params(k)
        r = oracle.main_order(k, oracle.tape_bytes_for(k, 7))
        assert r["verify"]
        a = 32 * p.M + 2 * p.M * 302
        b = a + 2 * k * p.E * 302
        assert r["used"] == [a, b, b + 64, p.tape_bytes]
        assert (len(r["rand"]), len(r["range"]), len(r["inst"])) == sizes
        # f -> NTT f inside the randomness struct
        import numpy as np
        f0 = np.frombuffer(r["rand"][:512], np.uint16).astype(np.int32)
        f0 = np.where(f0 > 1664, f0 - 3329, f0).astype(np.int16)
        ntt0 = np.frombuffer(r["rand"][p.M * 512:p.M * 512 + 512], np.uint16)
        assert np.array_equal(np.mod(oracle.poly_ntt(f0).astype(np.int32), 3329).astype(np.uint16), ntt0)
