"""ctypes access to oracle/libkosk_oracle.so (the CPU restatement) -- tests only."""
import ctypes as C
import hashlib
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = C.CDLL(os.path.join(ROOT, "oracle", "libkosk_oracle.so"))


class Params(C.Structure):
    _fields_ = [("K", C.c_int), ("eta1", C.c_int), ("M", C.c_int), ("V", C.c_int), ("E", C.c_int), ("Z", C.c_int),
                ("pk_bytes", C.c_size_t), ("sk_bytes", C.c_size_t), ("proof_bytes", C.c_size_t), ("tape_bytes", C.c_size_t),
                ("tape_calls", C.c_int), ("tcomm_msg_bytes", C.c_size_t), ("view_msg_bytes", C.c_size_t),
                ("off", C.c_size_t * 24), ("size", C.c_size_t * 24)]


class Tape(C.Structure):
    _fields_ = [("buf", C.c_char_p), ("len", C.c_size_t), ("pos", C.c_size_t), ("calls", C.c_size_t), ("overrun", C.c_int)]


class Trace(C.Structure):
    _fields_ = [("tcomm", C.c_uint8 * (1454 * 32)), ("h1", C.c_uint8 * 32), ("alpha", C.c_uint16 * 78),
                ("view_digest", C.c_uint8 * (1454 * 32)), ("ch", C.c_uint8 * 32),
                ("sr_rec", C.c_uint16 * (4 * 256)), ("er_rec", C.c_uint16 * (4 * 256))]


lib.ko_table_share_ddeg.restype = C.POINTER(C.c_uint16)
lib.ko_table_recon_ddeg.restype = C.POINTER(C.c_uint16)
lib.ko_table_recon_2ddeg.restype = C.POINTER(C.c_uint16)
lib.ko_zetas.restype = C.POINTER(C.c_int16)
lib.ko_gf_add.restype = lib.ko_gf_sub.restype = lib.ko_gf_mul.restype = lib.ko_gf_inv.restype = C.c_uint16
lib.ko_gf_add.argtypes = lib.ko_gf_sub.argtypes = lib.ko_gf_mul.argtypes = [C.c_uint16, C.c_uint16]
lib.ko_gf_inv.argtypes = [C.c_uint16]
lib.ko_barrett_reduce.restype = C.c_int16
lib.ko_barrett_reduce.argtypes = [C.c_int16]
lib.ko_montgomery_reduce.restype = C.c_int16
lib.ko_montgomery_reduce.argtypes = [C.c_int32]


def params(k):
    p = Params()
    assert lib.ko_get_params(k, C.byref(p)) == 0
    return p


def tape_bytes_for(k, index, prefix="kosk-tape-v1:"):
    """tape b = SHAKE256("kosk-tape-v1:<b>") byte stream."""
    return hashlib.shake_256((prefix + str(index)).encode()).digest(params(k).tape_bytes)


def verifiable_keygen(k, tape, trace=False):
    p = params(k)
    t = Tape(tape, len(tape), 0, 0, 0)
    pk = C.create_string_buffer(p.pk_bytes); sk = C.create_string_buffer(p.sk_bytes); pi = C.create_string_buffer(p.proof_bytes)
    tr = Trace() if trace else None
    lib.ko_verifiable_keygen(k, C.byref(t), pk, sk, pi, C.byref(tr) if trace else None)
    assert not t.overrun
    out = (pk.raw, sk.raw, pi.raw, t.calls, t.pos)
    return out + (tr,) if trace else out


def crafted_verifiable_keygen(k, tape, items):
    """ko_verifiable_keygen by the oracle's CRAFTING prover (kosk_oracle.c: ko_craft_add): items = [(kind, idx, party, mult)] adds
    mult * q to share idx of the party (kind 0 s, 1 e, 2 f, 3 NTT f) before the prover commits to its shares, so the proof's
    Fiat-Shamir hashes are consistent with the non-canonical u16 it holds.  Returns (pk, sk, pi)."""
    lib.ko_craft_clear()
    try:
        for it in items:
            assert lib.ko_craft_add(*it) == 0
        return verifiable_keygen(k, tape)[:3]
    finally:
        lib.ko_craft_clear()


def kosk_verify(k, pi, pk):
    why = C.create_string_buffer(256)
    ok = lib.ko_kosk_verify(k, C.c_char_p(pi), C.c_char_p(pk), why, 256)
    return bool(ok), why.value.decode()


def sha3_256(data):
    out = C.create_string_buffer(32)
    lib.ko_sha3_256(out, C.c_char_p(bytes(data)), len(data))
    return out.raw


def sha3_512(data):
    out = C.create_string_buffer(64)
    lib.ko_sha3_512(out, C.c_char_p(bytes(data)), len(data))
    return out.raw


def shake256(data, n):
    out = C.create_string_buffer(n)
    lib.ko_shake256(out, n, C.c_char_p(bytes(data)), len(data))
    return out.raw


def poly_ntt(a):
    a = np.ascontiguousarray(a, dtype=np.int16).copy()
    lib.ko_poly_ntt(a.ctypes.data_as(C.c_void_p))
    return a


def recompute_shares(y407):
    y = np.ascontiguousarray(y407, dtype=np.uint16)
    out = np.zeros(1454, np.uint16)
    lib.ko_recompute_share_secrets_ddeg(out.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p))
    return out


def recon(shares, two_d=False):
    s = np.ascontiguousarray(shares, dtype=np.uint16)
    out = np.zeros(256, np.uint16)
    (lib.ko_recon_secrets_2ddeg if two_d else lib.ko_recon_secrets_ddeg)(out.ctypes.data_as(C.c_void_p), s.ctypes.data_as(C.c_void_p))
    return out


def table(which):
    fn, shape = [(lib.ko_table_share_ddeg, (1303, 407)), (lib.ko_table_recon_ddeg, (256, 407)), (lib.ko_table_recon_2ddeg, (256, 813))][which]
    return np.ctypeslib.as_array(fn(), shape=shape).copy()


class Mlwe(C.Structure):  # ko_mlwe, oracle/kosk_oracle.h
    _fields_ = [("A", C.c_int16 * (4 * 4 * 256)), ("t", C.c_int16 * (4 * 256)), ("s", C.c_int16 * (4 * 256)), ("e", C.c_int16 * (4 * 256))]


lib.ko_pre_alloc.restype = C.c_void_p
lib.ko_pre_free.argtypes = [C.c_void_p]
for _n in ("ko_pre_f", "ko_pre_ntt_f", "ko_pre_f_shares", "ko_pre_ntt_f_shares"):
    getattr(lib, _n).restype = C.POINTER(C.c_uint16)
    getattr(lib, _n).argtypes = [C.c_void_p, C.c_int]
for _n in ("ko_pre_s_eta_shares", "ko_pre_e_eta_shares"):
    getattr(lib, _n).restype = C.POINTER(C.c_uint16)
    getattr(lib, _n).argtypes = [C.c_void_p, C.c_int, C.c_int]
lib.ko_prepare_randomness.argtypes = lib.ko_prepare_range_proof.argtypes = [C.c_int, C.c_void_p, C.c_void_p]
lib.ko_prove.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
lib.ko_keygen.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
lib.ko_verify.argtypes = [C.c_int, C.c_char_p, C.c_void_p, C.c_char_p, C.c_size_t]


def mlwe_inst_bytes(k, raw):
    """ko_mlwe -> the reference's mlwe_inst image for KYBER_K = k: A[K][K][256], t, s, e (mlwe_prover.hpp:34-37)"""
    A = np.ctypeslib.as_array(raw.A).reshape(4, 4, 256)[:k, :k]
    parts = [A] + [np.ctypeslib.as_array(getattr(raw, f)).reshape(4, 256)[:k] for f in ("t", "s", "e")]
    return b"".join(np.ascontiguousarray(x, dtype=np.int16).tobytes() for x in parts)


def main_order(k, tape):
    """The call order of the reference's main.cpp:21-47 on one tape: prepare_randomness, prepare_range_proof, kyber_keygen,
    prove, verify.  Returns the offline material in the reference's struct layouts, the raw instance, the proof, and how
    many tape bytes each call consumed."""
    p = params(k)
    t = Tape(tape, len(tape), 0, 0, 0)
    pre = lib.ko_pre_alloc()
    used = []
    lib.ko_prepare_randomness(k, C.byref(t), pre); used.append(t.pos)
    lib.ko_prepare_range_proof(k, C.byref(t), pre); used.append(t.pos)
    raw = Mlwe()
    pk = C.create_string_buffer(p.pk_bytes); sk = C.create_string_buffer(p.sk_bytes)
    lib.ko_keygen(k, C.byref(t), pk, sk, C.byref(raw)); used.append(t.pos)
    pi = C.create_string_buffer(p.proof_bytes)
    lib.ko_prove(k, C.byref(t), pi, C.byref(raw), pre, None); used.append(t.pos)
    assert not t.overrun
    why = C.create_string_buffer(256)
    ok = bool(lib.ko_verify(k, pi.raw, C.byref(raw), why, 256))
    M, E = p.M, p.E

    def share_vec(ptr):
        y = np.ctypeslib.as_array(ptr, shape=(1454,)).astype(np.uint16)
        return (1454).to_bytes(8, "little") + (np.arange(1454, dtype=np.uint16) + 256).tobytes() + y.tobytes()
    f = b"".join(np.ctypeslib.as_array(lib.ko_pre_f(pre, i), shape=(256,)).tobytes() for i in range(M))
    nf = b"".join(np.ctypeslib.as_array(lib.ko_pre_ntt_f(pre, i), shape=(256,)).tobytes() for i in range(M))
    fs = b"".join(share_vec(lib.ko_pre_f_shares(pre, i)) for i in range(M))
    nfs = b"".join(share_vec(lib.ko_pre_ntt_f_shares(pre, i)) for i in range(M))
    rand = f + nf + fs + nfs
    rng = b"".join(share_vec(lib.ko_pre_s_eta_shares(pre, i, j)) for i in range(k) for j in range(E)) + \
        b"".join(share_vec(lib.ko_pre_e_eta_shares(pre, i, j)) for i in range(k) for j in range(E))
    lib.ko_pre_free(pre)
    return {"rand": rand, "range": rng, "inst": mlwe_inst_bytes(k, raw), "pk": pk.raw, "sk": sk.raw, "pi": pi.raw, "verify": ok,
            "used": used}
