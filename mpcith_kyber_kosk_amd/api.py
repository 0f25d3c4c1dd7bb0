"""ctypes binding of libkosk_mi355x.so -- the host-side mirror of the reference
API (kosk.hpp:18-24) plus the kernel-level entry points used by tests/bench.

There is NO CPU fallback: constructing :class:`Kosk` without a usable HIP device
raises, and importing this module without the built library raises.
"""
import ctypes as C
import os

from . import build as _build

_HERE = os.path.dirname(os.path.abspath(__file__))
# KOSK_LIB_PATH: another build of the same library (A/B measurements of two trees on one GPU box); never a different product
LIB_PATH = os.environ.get("KOSK_LIB_PATH") or os.path.join(_HERE, "libkosk_mi355x.so")


class KoskError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        _build.build()
    lib = C.CDLL(LIB_PATH)
    u8p, u16p, i16p, sz, vp = C.POINTER(C.c_uint8), C.POINTER(C.c_uint16), C.POINTER(C.c_int16), C.c_size_t, C.c_void_p
    sig = {
        "kosk_pk_bytes": (sz, [C.c_int]), "kosk_sk_bytes": (sz, [C.c_int]),
        "kosk_proof_bytes": (sz, [C.c_int]), "kosk_tape_bytes": (sz, [C.c_int]),
        "kosk_proof_field": (C.c_int, [C.c_int, C.c_int, C.POINTER(sz), C.POINTER(sz)]),
        "kosk_create": (C.c_int, [C.POINTER(vp), C.c_int, C.c_int, C.c_int]),
        "kosk_options_init": (None, [vp]),
        "kosk_create_ex": (C.c_int, [C.POINTER(vp), C.c_int, C.c_int, C.c_int, vp]),
        "kosk_destroy": (None, [vp]),
        "kosk_last_error": (C.c_char_p, [vp]),
        "kosk_set_randombytes": (C.c_int, [vp, vp, vp]),
        "kosk_verifiable_keygen_batch": (C.c_int, [vp, C.c_int, vp, sz, vp, vp, vp]),
        "kosk_verify_batch": (C.c_int, [vp, C.c_int, vp, vp, vp]),
        "kosk_verify_fail_masks": (C.c_int, [vp, vp, C.c_int]),
        "kosk_randomness_bytes": (sz, [C.c_int]),
        "kosk_range_proof_bytes": (sz, [C.c_int]),
        "kosk_mlwe_inst_bytes": (sz, [C.c_int]),
        "kosk_prepare_randomness": (C.c_int, [vp, C.c_int, vp, sz, vp]),
        "kosk_prepare_range_proof": (C.c_int, [vp, C.c_int, vp, sz, vp]),
        "kosk_prove_prepared": (C.c_int, [vp, C.c_int, vp, vp, vp, vp, sz, vp]),
        "kosk_verify_inst": (C.c_int, [vp, C.c_int, vp, vp, vp]),
        "kosk_compact_proof_bytes": (sz, [C.c_int]),
        "kosk_proof_compress": (C.c_int, [C.c_int, vp, vp]),
        "kosk_proof_decompress": (C.c_int, [C.c_int, vp, vp]),
        "kosk_fetch_proofs_compact": (C.c_int, [vp, C.c_int, vp]),
        "kosk_verifiable_keygen_batch_compact": (C.c_int, [vp, C.c_int, vp, sz, vp, vp, vp]),
        "kosk_verify_batch_compact": (C.c_int, [vp, C.c_int, vp, vp, vp]),
        "kosk_stage_verifier_inputs_compact": (C.c_int, [vp, C.c_int, vp, vp]),
        "kosk_stage_prover_inputs": (C.c_int, [vp, C.c_int, vp, sz, vp, vp]),
        "kosk_prove_resident": (C.c_int, [vp, C.c_int]),
        "kosk_fetch_proofs": (C.c_int, [vp, C.c_int, vp]),
        "kosk_stage_verifier_inputs": (C.c_int, [vp, C.c_int, vp, vp]),
        "kosk_verify_resident": (C.c_int, [vp, C.c_int, vp]),
        "kosk_verifiable_keygen_resident": (C.c_int, [vp, C.c_int, vp, sz, vp, vp]),
        "kosk_verify_resident_pk": (C.c_int, [vp, C.c_int, vp, vp]),
        "kosk_resident_digests": (C.c_int, [vp, C.c_int, C.POINTER(vp), C.POINTER(sz)]),
        "kosk_set_round_hook": (C.c_int, [vp, vp, vp]),
        "kosk_phase_seconds": (C.c_int, [vp, C.POINTER(C.c_double), C.c_int]),
        "kosk_path_count": (C.c_int, [vp, C.c_int, C.POINTER(C.c_long)]),
        "kosk_host_threads": (C.c_int, [vp]),
        "kosk_host_alloc": (vp, [sz]),
        "kosk_host_free": (None, [vp]),
        "kosk_sha3_256_batch": (C.c_int, [vp, vp, sz, sz, vp, C.c_int]),
        "kosk_shake256_batch": (C.c_int, [vp, vp, sz, sz, vp, sz, C.c_int]),
        "kosk_sha3_256_batch_pair": (C.c_int, [vp, vp, sz, sz, vp, C.c_int]),
        "kosk_sha3_256_batch_wave": (C.c_int, [vp, vp, sz, sz, vp, C.c_int]),
        "kosk_fs_alpha_device": (C.c_int, [vp, vp, sz, C.c_int, vp, vp]),
        "kosk_fs_opened_device": (C.c_int, [vp, vp, sz, C.c_int, vp, vp, C.c_int, vp]),
        "kosk_commit_hash_lanes": (C.c_int, [vp, vp, sz, C.c_int, vp, C.c_int, vp]),
        "kosk_ntt256_batch": (C.c_int, [vp, vp, vp, C.c_int]),
        "kosk_lagrange_expand": (C.c_int, [vp, vp, vp, C.c_int]),
        "kosk_recon_secrets": (C.c_int, [vp, vp, vp, C.c_int, C.c_int]),
        "kosk_profile_enable": (C.c_int, [vp, C.c_int]),
        "kosk_profile_read": (C.c_int, [vp, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_long)]),
        "kosk_profile_read_units": (C.c_int, [vp, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_long), C.POINTER(C.c_long)]),
        "kosk_combine_stats": (C.c_int, [vp, C.POINTER(C.c_long), C.POINTER(C.c_long)]),
        "kosk_stream_timer_start": (C.c_int, [vp]),
        "kosk_stream_timer_stop": (C.c_int, [vp, C.POINTER(C.c_double)]),
        "kosk_device_synchronize": (C.c_int, [vp]),
        "kosk_streams": (C.c_int, [vp]),
        "kosk_resident_proofs": (C.c_int, [vp, C.POINTER(vp), C.POINTER(sz)]),
        "kosk_keygen": (C.c_int, [C.c_int, vp, vp, vp, vp, vp, vp, vp]),
        "kosk_fs_alpha": (C.c_int, [C.c_int, vp, vp]),
        "kosk_fs_opened": (C.c_int, [vp, vp, vp]),
        "kosk_host_sha3_256": (None, [vp, vp, sz]),
        "kosk_host_shake256": (None, [vp, sz, vp, sz]),
        "kosk_host_sha3_256_multi": (C.c_int, [vp, vp, sz, sz, C.c_int, C.c_int]),
        "kosk_lagrange_table": (C.c_int, [C.c_int, vp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)  # AttributeError if the header and the library ever disagree
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()
EXPORTS = ["kosk_pk_bytes", "kosk_sk_bytes", "kosk_proof_bytes", "kosk_tape_bytes", "kosk_proof_field", "kosk_create",
           "kosk_destroy", "kosk_last_error", "kosk_set_randombytes", "kosk_verifiable_keygen_batch", "kosk_verify_batch",
           "kosk_verify_fail_masks", "kosk_randomness_bytes", "kosk_range_proof_bytes", "kosk_mlwe_inst_bytes",
           "kosk_prepare_randomness", "kosk_prepare_range_proof", "kosk_prove_prepared", "kosk_verify_inst", "kosk_compact_proof_bytes",
           "kosk_proof_compress", "kosk_proof_decompress", "kosk_fetch_proofs_compact", "kosk_stage_verifier_inputs_compact", "kosk_stage_prover_inputs", "kosk_prove_resident", "kosk_fetch_proofs",
           "kosk_stage_verifier_inputs", "kosk_verify_resident", "kosk_verifiable_keygen_resident", "kosk_verify_resident_pk",
           "kosk_resident_digests", "kosk_set_round_hook", "kosk_phase_seconds", "kosk_path_count", "kosk_host_threads", "kosk_sha3_256_batch_pair", "kosk_verifiable_keygen_batch_compact", "kosk_verify_batch_compact", "kosk_host_alloc", "kosk_host_free", "kosk_sha3_256_batch",
           "kosk_shake256_batch", "kosk_commit_hash_lanes", "kosk_ntt256_batch", "kosk_lagrange_expand",
           "kosk_recon_secrets", "kosk_profile_enable", "kosk_profile_read", "kosk_profile_read_units", "kosk_combine_stats", "kosk_stream_timer_start", "kosk_stream_timer_stop", "kosk_device_synchronize", "kosk_streams", "kosk_resident_proofs", "kosk_keygen", "kosk_fs_alpha",
           "kosk_fs_opened", "kosk_host_sha3_256", "kosk_host_shake256", "kosk_host_sha3_256_multi", "kosk_lagrange_table",
           "kosk_options_init", "kosk_create_ex", "kosk_sha3_256_batch_wave", "kosk_fs_alpha_device", "kosk_fs_opened_device"]


class KoskOptions(C.Structure):
    """kosk_options of include/kosk_mi355x.h (per-handle configuration, round 6); build with options(**fields)"""
    _fields_ = [("size", C.c_uint32), ("streams", C.c_int32), ("combine", C.c_int32), ("combine_wait_us", C.c_int32),
                ("combine_idle_us", C.c_int32), ("combine_prewake_us", C.c_int32), ("strict_encoding", C.c_int32), ("fs_mode", C.c_int32),
                ("host_threads", C.c_int32), ("blocking_sync", C.c_int32), ("hooks_unmerged", C.c_int32), ("reserved", C.c_int32 * 6)]


FS_HOST, FS_DEVICE = 0, 1


def options(**fields):
    o = KoskOptions()
    lib.kosk_options_init(C.byref(o))
    for k_, v_ in fields.items():
        if k_ not in dict(KoskOptions._fields_) or k_ in ("size", "reserved"):
            raise KoskError("unknown option " + k_)
        setattr(o, k_, int(v_))
    return o


def pk_bytes(k): return lib.kosk_pk_bytes(k)
def sk_bytes(k): return lib.kosk_sk_bytes(k)
def proof_bytes(k): return lib.kosk_proof_bytes(k)
def tape_bytes(k): return lib.kosk_tape_bytes(k)


def proof_field(k, idx):
    off, size = C.c_size_t(), C.c_size_t()
    if lib.kosk_proof_field(k, idx, C.byref(off), C.byref(size)):
        raise KoskError("bad field")
    return off.value, size.value


def _buf(b):
    """bytes-like -> (ctypes pointer value, keepalive)"""
    if isinstance(b, (bytes, bytearray)):
        arr = (C.c_uint8 * len(b)).from_buffer_copy(b) if isinstance(b, bytes) else (C.c_uint8 * len(b)).from_buffer(b)
        return C.cast(arr, C.c_void_p), arr
    raise TypeError(type(b))


def host_keygen(k, seed64):
    """kyber_keygen (kosk.cpp:4-70) on the host; returns pk, sk, A, s, e, t (lists of int)."""
    import numpy as np
    pk = C.create_string_buffer(pk_bytes(k)); sk = C.create_string_buffer(sk_bytes(k))
    A = np.zeros(k * k * 256, np.int16); s = np.zeros(k * 256, np.int16); e = np.zeros(k * 256, np.int16); t = np.zeros(k * 256, np.int16)
    r = lib.kosk_keygen(k, C.c_char_p(bytes(seed64)), pk, sk, A.ctypes.data, s.ctypes.data, e.ctypes.data, t.ctypes.data)
    if r:
        raise KoskError("kosk_keygen failed")
    return pk.raw, sk.raw, A, s, e, t


def host_sha3_256(data):
    out = C.create_string_buffer(32)
    lib.kosk_host_sha3_256(out, C.c_char_p(bytes(data)), len(data))
    return out.raw


def host_shake256(data, outlen):
    out = C.create_string_buffer(outlen)
    lib.kosk_host_shake256(out, outlen, C.c_char_p(bytes(data)), len(data))
    return out.raw


class DeviceView:
    """A window on library-owned HBM for torch (torch.as_tensor(view, device="cuda") is zero-copy): uint8, C-contiguous."""

    def __init__(self, ptr, shape):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": "|u1", "data": (int(ptr), False), "version": 2, "strides": None}


class Kosk:
    """One library context = one GPU, one parameter set, up to max_batch proofs in flight.

    Mirrors the reference's top-level API (kosk.hpp):
      verifiable_keygen(tapes) -> (pk, sk, pi) lists      kyber_verifiable_keygen
      verify(pi, pk) -> list[bool]                         kyber_kosk_verify
    """

    def __init__(self, kyber_k=2, max_batch=1, device=0, **opts):
        """opts: fields of kosk_options (streams, combine, strict_encoding, fs_mode, host_threads, blocking_sync, ...): the handle is
        then created with kosk_create_ex; without any, with kosk_create (library defaults and the KOSK_* environment)"""
        self.k = kyber_k
        self.max_batch = max_batch
        self._h = C.c_void_p()
        if opts:
            self._opts = options(**opts)
            rc = lib.kosk_create_ex(C.byref(self._h), device, kyber_k, max_batch, C.byref(self._opts))
        else:
            rc = lib.kosk_create(C.byref(self._h), device, kyber_k, max_batch)
        if rc:
            raise KoskError("kosk_create: " + lib.kosk_last_error(None).decode())
        self.pk_bytes, self.sk_bytes = pk_bytes(kyber_k), sk_bytes(kyber_k)
        self.proof_bytes, self.tape_bytes = proof_bytes(kyber_k), tape_bytes(kyber_k)
        self._cb = None
        self._hook = None
        self._pk = self._sk = None

    def close(self):
        if self._h:
            lib.kosk_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, r, what):
        if r:
            raise KoskError(what + ": " + lib.kosk_last_error(self._h).decode())

    @property
    def handle(self):
        return self._h

    def set_randombytes(self, fn):
        """fn(nbytes) -> bytes, called in the reference's randombytes order."""
        if fn is None:
            self._cb = None
            self._chk(lib.kosk_set_randombytes(self._h, None, None), "set_randombytes")
            return
        CB = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_uint8), C.c_size_t)

        def tramp(_user, out, n):
            data = fn(n)
            C.memmove(out, data, n)
        self._cb = CB(tramp)
        self._chk(lib.kosk_set_randombytes(self._h, C.cast(self._cb, C.c_void_p), None), "set_randombytes")

    def verifiable_keygen(self, tapes=None, n=None):
        """tapes: list of bytes (one randomness tape per instance) or None (callback / OS entropy)."""
        if tapes is not None:
            n = len(tapes)
            stride = self.tape_bytes
            blob = b"".join(t[:stride].ljust(stride, b"\0") for t in tapes)
            for t in tapes:
                if len(t) < stride:
                    raise KoskError("tape shorter than kosk_tape_bytes")
            tp = C.c_char_p(blob)
        else:
            if n is None:
                n = 1
            stride, tp = 0, None
        pk = C.create_string_buffer(self.pk_bytes * n); sk = C.create_string_buffer(self.sk_bytes * n)
        pi = C.create_string_buffer(self.proof_bytes * n)
        self._chk(lib.kosk_verifiable_keygen_batch(self._h, n, tp, stride, pk, sk, pi), "verifiable_keygen")
        cut = lambda b, s: [b.raw[i * s:(i + 1) * s] for i in range(n)]
        return cut(pk, self.pk_bytes), cut(sk, self.sk_bytes), cut(pi, self.proof_bytes)

    def verify(self, pis, pks):
        n = len(pis)
        ok = C.create_string_buffer(n)
        self._chk(lib.kosk_verify_batch(self._h, n, C.c_char_p(b"".join(pis)), C.c_char_p(b"".join(pks)), ok), "verify")
        return [b == 1 for b in ok.raw]

    def fail_masks(self, n):
        m = (C.c_uint32 * n)()
        self._chk(lib.kosk_verify_fail_masks(self._h, m, n), "fail_masks")
        return list(m)

    # compact wire format
    def fetch_proofs_compact(self, n):
        size = lib.kosk_compact_proof_bytes(self.k)
        out = C.create_string_buffer(size * n)
        self._chk(lib.kosk_fetch_proofs_compact(self._h, n, out), "fetch_proofs_compact")
        return [out.raw[i * size:(i + 1) * size] for i in range(n)]

    def stage_verifier_inputs_compact(self, blobs, pks):
        self._chk(lib.kosk_stage_verifier_inputs_compact(self._h, len(blobs), b"".join(blobs), b"".join(pks)), "stage_verifier_inputs_compact")

    # second-level entry points (reference structs as bytes; see include/kosk_mi355x.h)
    def stage_verifier_inputs(self, pis, pks):
        """proof images and public keys into HBM for verify_resident (kosk_stage_verifier_inputs)"""
        self._chk(lib.kosk_stage_verifier_inputs(self._h, len(pis), b"".join(pis), b"".join(pks)), "stage_verifier_inputs")

    def prepare_randomness(self, tapes=None, n=None):
        n = len(tapes) if tapes is not None else n
        size = lib.kosk_randomness_bytes(self.k)
        out = C.create_string_buffer(size * n)
        blob = b"".join(tapes) if tapes is not None else None
        self._chk(lib.kosk_prepare_randomness(self._h, n, blob, len(tapes[0]) if tapes is not None else 0, out), "prepare_randomness")
        return [out.raw[i * size:(i + 1) * size] for i in range(n)]

    def prepare_range_proof(self, tapes=None, n=None):
        n = len(tapes) if tapes is not None else n
        size = lib.kosk_range_proof_bytes(self.k)
        out = C.create_string_buffer(size * n)
        blob = b"".join(tapes) if tapes is not None else None
        self._chk(lib.kosk_prepare_range_proof(self._h, n, blob, len(tapes[0]) if tapes is not None else 0, out), "prepare_range_proof")
        return [out.raw[i * size:(i + 1) * size] for i in range(n)]

    def prove_prepared(self, insts, rands, ranges, tapes=None):
        n = len(insts)
        pi = C.create_string_buffer(self.proof_bytes * n)
        blob = b"".join(tapes) if tapes is not None else None
        self._chk(lib.kosk_prove_prepared(self._h, n, b"".join(insts), b"".join(rands), b"".join(ranges), blob,
                                          len(tapes[0]) if tapes is not None else 0, pi), "prove_prepared")
        return [pi.raw[i * self.proof_bytes:(i + 1) * self.proof_bytes] for i in range(n)]

    def verify_inst(self, proofs, insts):
        n = len(proofs)
        ok = (C.c_uint8 * n)()
        self._chk(lib.kosk_verify_inst(self._h, n, b"".join(proofs), b"".join(insts), ok), "verify_inst")
        return [bool(x) for x in ok]

    # resident split (bench)
    def stage_prover_inputs(self, tapes):
        n = len(tapes)
        blob = b"".join(t[:self.tape_bytes] for t in tapes)
        self._pk = C.create_string_buffer(self.pk_bytes * n); self._sk = C.create_string_buffer(self.sk_bytes * n)
        self._chk(lib.kosk_stage_prover_inputs(self._h, n, C.c_char_p(blob), self.tape_bytes, self._pk, self._sk), "stage_prover_inputs")
        return n

    def prove_resident(self, n):
        self._chk(lib.kosk_prove_resident(self._h, n), "prove_resident")

    def verify_resident(self, n):
        ok = C.create_string_buffer(n)
        self._chk(lib.kosk_verify_resident(self._h, n, ok), "verify_resident")
        return [b == 1 for b in ok.raw]

    def fetch_proofs(self, n):
        pi = C.create_string_buffer(self.proof_bytes * n)
        self._chk(lib.kosk_fetch_proofs(self._h, n, pi), "fetch_proofs")
        return [pi.raw[i * self.proof_bytes:(i + 1) * self.proof_bytes] for i in range(n)]

    def verifiable_keygen_resident(self, tapes, n=None, tape_stride=None):
        """kyber_verifiable_keygen as one resident call: key generation + prove, pk/sk returned, proofs stay in HBM.
        tapes: list of bytes, or an int DEVICE pointer (with n and tape_stride), or None (callback / OS entropy, with n)."""
        if isinstance(tapes, int):
            tp, stride = C.c_void_p(tapes), tape_stride
        elif tapes is None:
            tp, stride = None, 0
        else:
            n = len(tapes)
            blob = b"".join(t[:self.tape_bytes] for t in tapes)
            tp, stride = C.c_char_p(blob), self.tape_bytes
        if getattr(self, "_pk", None) is None or len(self._pk) != self.pk_bytes * n:
            self._pk = C.create_string_buffer(self.pk_bytes * n); self._sk = C.create_string_buffer(self.sk_bytes * n)
        self._chk(lib.kosk_verifiable_keygen_resident(self._h, n, tp, stride, self._pk, self._sk), "verifiable_keygen_resident")
        return n

    def keys(self, n):
        """pk, sk lists of the last stage_prover_inputs / verifiable_keygen_resident"""
        return ([self._pk.raw[i * self.pk_bytes:(i + 1) * self.pk_bytes] for i in range(n)],
                [self._sk.raw[i * self.sk_bytes:(i + 1) * self.sk_bytes] for i in range(n)])

    def verify_resident_pk(self, n, pks=None):
        """kyber_kosk_verify on the resident proofs, pk decoding (polyvec_frombytes + gen_matrix) included; pks None = the
        pk bytes the key generation left in HBM"""
        ok = C.create_string_buffer(n)
        self._chk(lib.kosk_verify_resident_pk(self._h, n, C.c_char_p(b"".join(pks)) if pks is not None else None, ok), "verify_resident_pk")
        return [b == 1 for b in ok.raw]

    def resident_digests(self, rnd, n):
        """DeviceView of the round's digest table [n][1454][32] (round 0 Tcomm, 1 view commitments)"""
        d, stride = C.c_void_p(), C.c_size_t()
        self._chk(lib.kosk_resident_digests(self._h, rnd, C.byref(d), C.byref(stride)), "resident_digests")
        assert stride.value == 1454 * 32
        return DeviceView(d.value, (n, 1454, 32))

    def set_round_hook(self, fn):
        """fn(role, round, device_ptr, nbytes) on the calling thread when a round's digest table is complete in HBM"""
        if fn is None:
            self._hook = None
            self._chk(lib.kosk_set_round_hook(self._h, None, None), "set_round_hook")
            return
        CB = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t)
        self._hook = CB(lambda _u, role, rnd, ptr, nbytes: fn(role, rnd, ptr, nbytes))
        self._chk(lib.kosk_set_round_hook(self._h, C.cast(self._hook, C.c_void_p), None), "set_round_hook")

    def phase_seconds(self):
        out = (C.c_double * 16)()
        lib.kosk_phase_seconds(self._h, out, 16)
        return list(out)

    PROFILE_IDS = ["hash_tcomm", "hash_view", "gemm_expand1", "gemm_expand2", "lincomb", "ntt_f", "assemble",
                   "v_hash_tcomm", "v_hash_view", "v_interp_build", "v_gemm_interp", "v_gemm_expand", "v_gemm_recon", "v_lincomb",
                   "fs_alpha", "fs_opened", "v_fs_alpha", "v_fs_opened"]

    def profile_enable(self, on=True):
        self._chk(lib.kosk_profile_enable(self._h, int(on)), "profile_enable")

    def profile_read(self):
        """{name: (total_ms, launches)} since profile_enable()"""
        out = {}
        for i, name in enumerate(self.PROFILE_IDS):
            ms, cnt = C.c_double(), C.c_long()
            lib.kosk_profile_read(self._h, i, C.byref(ms), C.byref(cnt))
            out[name] = (ms.value, cnt.value)
        return out

    def profile_read_units(self):
        """{name: (total_ms, launches, proofs served by those launches)}: a merged run of a cohort (KOSK_COMBINE) serves several
        callers' batches per launch and is timed on the handle that led it"""
        out = {}
        for i, name in enumerate(self.PROFILE_IDS):
            ms, cnt, units = C.c_double(), C.c_long(), C.c_long()
            lib.kosk_profile_read_units(self._h, i, C.byref(ms), C.byref(cnt), C.byref(units))
            out[name] = (ms.value, cnt.value, units.value)
        return out

    def combine_stats(self):
        """(resident calls of this handle served through the combiner, sum over them of the members their run served)"""
        a, b = C.c_long(), C.c_long()
        self._chk(lib.kosk_combine_stats(self._h, C.byref(a), C.byref(b)), "combine_stats")
        return a.value, b.value

    PATH_IDS = ["hash_dma", "hash_plain", "table_gemm", "limb_gemm", "copy_direct", "copy_staged", "graph_replay", "digest_copy", "small_copy_kernel",
                "fs_device", "fs_host"]

    def path_counts(self):
        """{name: launches / copies} of the alternative kernel and copy paths on this handle since it was created"""
        out = {}
        for i, name in enumerate(self.PATH_IDS):
            v = C.c_long()
            self._chk(lib.kosk_path_count(self._h, i, C.byref(v)), "path_count")
            out[name] = v.value
        return out

    @property
    def host_threads(self):
        return lib.kosk_host_threads(self._h)

    def timer_start(self):
        self._chk(lib.kosk_stream_timer_start(self._h), "timer_start")

    def timer_stop_ms(self):
        ms = C.c_double()
        self._chk(lib.kosk_stream_timer_stop(self._h, C.byref(ms)), "timer_stop")
        return ms.value

    @property
    def streams(self):
        return lib.kosk_streams(self._h)

    def synchronize(self):
        self._chk(lib.kosk_device_synchronize(self._h), "synchronize")

    # kernel-level (device pointers as ints, e.g. torch tensor .data_ptr())
    def sha3_256_batch(self, d_in, in_stride, inlen, d_out, n):
        self._chk(lib.kosk_sha3_256_batch(self._h, d_in, in_stride, inlen, d_out, n), "sha3_256_batch")

    def sha3_256_batch_pair(self, d_in, in_stride, inlen, d_out, n):
        self._chk(lib.kosk_sha3_256_batch_pair(self._h, d_in, in_stride, inlen, d_out, n), "sha3_256_batch_pair")

    def sha3_256_batch_wave(self, d_in, in_stride, inlen, d_out, n):
        self._chk(lib.kosk_sha3_256_batch_wave(self._h, d_in, in_stride, inlen, d_out, n), "sha3_256_batch_wave")

    def fs_alpha_device(self, d_tables, table_stride, n, d_alpha, d_h1=None):
        self._chk(lib.kosk_fs_alpha_device(self._h, d_tables, table_stride, n, d_alpha, d_h1), "fs_alpha_device")

    def fs_opened_device(self, d_tables, table_stride, n, d_sel, d_rest, sel_stride, d_ch=None):
        self._chk(lib.kosk_fs_opened_device(self._h, d_tables, table_stride, n, d_sel, d_rest, sel_stride, d_ch), "fs_opened_device")

    def shake256_batch(self, d_in, in_stride, inlen, d_out, outlen, n):
        self._chk(lib.kosk_shake256_batch(self._h, d_in, in_stride, inlen, d_out, outlen, n), "shake256_batch")

    def commit_hash_lanes(self, d_rows, row_stride, n_lanes, d_prefix, with_prefix, d_out):
        self._chk(lib.kosk_commit_hash_lanes(self._h, d_rows, row_stride, n_lanes, d_prefix, int(with_prefix), d_out), "commit_hash_lanes")

    def ntt256_batch(self, d_in, d_out, n):
        self._chk(lib.kosk_ntt256_batch(self._h, d_in, d_out, n), "ntt256_batch")

    def lagrange_expand(self, d_y407, d_shares, n):
        self._chk(lib.kosk_lagrange_expand(self._h, d_y407, d_shares, n), "lagrange_expand")

    def recon_secrets(self, d_shares, d_secrets, n, two_d=False):
        self._chk(lib.kosk_recon_secrets(self._h, d_shares, d_secrets, n, int(two_d)), "recon_secrets")
