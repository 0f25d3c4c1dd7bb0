"""Host-side logic of the PRODUCT library (no GPU needed): ABI surface, keygen, Fiat-Shamir derivations,
tables.  The oracle is only the checker here."""
import ctypes as C
import hashlib
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def api():
    from mpcith_kyber_kosk_amd import api
    return api


def test_library_exports_every_declared_symbol(api):
    hdr = open(os.path.join(ROOT, "include", "kosk_mi355x.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(kosk_[a-z0-9_]+)\s*\(", hdr)) - {"kosk_randombytes_fn"}
    assert len(declared) >= 30
    for name in sorted(declared):
        assert hasattr(api.lib, name), name
    assert declared == set(api.EXPORTS), declared ^ set(api.EXPORTS)


@pytest.mark.parametrize("k", [2, 3, 4])
def test_sizes_and_field_layout(k, api, oracle):
    p = oracle.params(k)
    assert api.pk_bytes(k) == p.pk_bytes and api.sk_bytes(k) == p.sk_bytes
    assert api.proof_bytes(k) == p.proof_bytes == {2: 664340, 3: 680980, 4: 744148}[k]
    assert api.tape_bytes(k) == p.tape_bytes
    for i in range(24):
        assert api.proof_field(k, i) == (p.off[i], p.size[i])
    assert api.lib.kosk_proof_bytes(5) == 0


def test_create_without_gpu_fails_loudly(api):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(api.KoskError, match="no HIP device|no CPU fallback|hip"):
        api.Kosk(kyber_k=2, max_batch=1)


@pytest.mark.parametrize("k", [2, 3, 4])
def test_host_keygen_matches_oracle(k, api, oracle):
    for idx in range(3):
        tape = oracle.tape_bytes_for(k, idx)
        t = oracle.Tape(tape, 64, 0, 0, 0)
        pk = C.create_string_buffer(api.pk_bytes(k)); sk = C.create_string_buffer(api.sk_bytes(k))
        raw = np.zeros(16 * 256 + 3 * 4 * 256, np.int16)
        oracle.lib.ko_keygen(k, C.byref(t), pk, sk, raw.ctypes.data_as(C.c_void_p))
        pk2, sk2, A, s, e, tt = api.host_keygen(k, tape[:64])
        assert pk2 == pk.raw and sk2 == sk.raw
        A_or = raw[:16 * 256].reshape(4, 4, 256)[:k, :k].reshape(-1)
        assert np.array_equal(A, A_or)
        t_or = raw[16 * 256:20 * 256].reshape(4, 256)[:k].reshape(-1)
        s_or = raw[20 * 256:24 * 256].reshape(4, 256)[:k].reshape(-1)
        e_or = raw[24 * 256:28 * 256].reshape(4, 256)[:k].reshape(-1)
        assert np.array_equal(tt, t_or) and np.array_equal(s, s_or) and np.array_equal(e, e_or)
        assert np.abs(s).max() <= (3 if k == 2 else 2)
        # sk quirk of kosk.cpp:67-69: the last 32 bytes are the noise seed, H(pk) before it
        assert sk2[-64:-32] == hashlib.sha3_256(pk2).digest()


@pytest.mark.parametrize("n", [0, 1, 135, 136, 137, 1000, 46528])
def test_host_keccak(n, api):
    d = bytes((i * 29 + 1) & 255 for i in range(n))
    assert api.host_sha3_256(d) == hashlib.sha3_256(d).digest()
    assert api.host_shake256(d, 302) == hashlib.shake_256(d).digest(302)


@pytest.mark.parametrize("count,length,threads", [(1, 46528, 1), (46, 46528, 4), (13, 137, 2), (9, 0, 1), (5, 136, 3)])
def test_host_multibuffer_sha3(count, length, threads, api):
    data = bytes(os.urandom(count * length)) if length else b""
    out = C.create_string_buffer(32 * count)
    w = api.lib.kosk_host_sha3_256_multi(out, C.c_char_p(data), length, length, count, threads)
    assert w in (1, 4, 8)
    for i in range(count):
        assert out.raw[32 * i:32 * i + 32] == hashlib.sha3_256(data[i * length:(i + 1) * length]).digest()


def _ref_opened(dig):
    ch = hashlib.sha3_256(dig).digest()
    I_ = hashlib.shake_256(ch + b"\x01").digest(300)
    I = [((I_[2 * i] << 8) | I_[2 * i + 1]) % 1454 for i in range(150)]
    for i in range(1, 150):  # mlwe_prover.cpp:459-474, transcribed loop for loop
        inc = 0
        while True:
            dup = False
            for j in range(i):
                if (I[i] + inc) % 1454 == I[j]:
                    dup = True
                    inc += 1
                    break
            if not dup:
                break
        I[i] = (I[i] + inc) % 1454
    return I


def test_fs_opened_matches_reference_loop_including_collisions(api):
    hits = 0
    for seed in range(40):
        dig = hashlib.shake_256(b"dig%d" % seed).digest(1454 * 32)
        I = (C.c_uint16 * 150)(); rest = (C.c_uint16 * 1304)()
        api.lib.kosk_fs_opened(C.c_char_p(dig), I, rest)
        ref = _ref_opened(dig)
        assert list(I) == ref
        assert sorted(set(range(1454)) - set(ref)) == list(rest)
        ch = hashlib.sha3_256(dig).digest()
        I_ = hashlib.shake_256(ch + b"\x01").digest(300)
        raw = [((I_[2 * i] << 8) | I_[2 * i + 1]) % 1454 for i in range(150)]
        hits += len(raw) != len(set(raw))
    assert hits > 10  # the de-duplication path was really exercised (birthday bound ~ 7.4 collisions / 1000 draws)


@pytest.mark.parametrize("k", [2, 3, 4])
def test_fs_alpha(k, api):
    dig = hashlib.shake_256(b"tcomm%d" % k).digest(1454 * 32)
    J = 70 + 2 * k
    alpha = (C.c_uint16 * 80)()
    api.lib.kosk_fs_alpha(k, C.c_char_p(dig), alpha)
    a_ = hashlib.shake_256(hashlib.sha3_256(dig).digest() + b"\x01").digest(2 * J)
    assert list(alpha)[:J] == [((a_[2 * i] << 8) | a_[2 * i + 1]) % 3329 for i in range(J)]


def test_lagrange_tables_match_oracle(api, oracle):
    for which, shape in enumerate([(1303, 407), (256, 407), (256, 813)]):
        t = np.zeros(shape, np.uint16)
        assert api.lib.kosk_lagrange_table(which, t.ctypes.data) == 0
        assert np.array_equal(t, oracle.table(which))


def test_compat_header_proof_size_macro(api):
    """include/kosk_compat.hpp restates sizeof(mpcith_proof) as a macro; check it for K=2,3,4 with the host compiler."""
    import subprocess, tempfile
    src = '#include "kosk_compat.hpp"\n#include <cstdio>\nextern "C" void randombytes(uint8_t*, size_t) {}\nint main(){ printf("%zu %d %d", (size_t)MPCITH_PROOF_SIZE, (int)KYBER_PUBLICKEYBYTES, (int)KYBER_SECRETKEYBYTES); }\n'
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "t.cpp"), "w").write(src)
        for k in (2, 3, 4):
            exe = os.path.join(d, "t%d" % k)
            subprocess.check_call(["g++", "-std=c++17", "-DKYBER_K=%d" % k, "-I" + os.path.join(ROOT, "include"), os.path.join(d, "t.cpp"),
                                   "-L" + os.path.join(ROOT, "mpcith_kyber_kosk_amd"), "-lkosk_mi355x",
                                   "-Wl,-rpath," + os.path.join(ROOT, "mpcith_kyber_kosk_amd"), "-o", exe])
            out = subprocess.check_output([exe], text=True).split()
            assert [int(x) for x in out] == [api.proof_bytes(k), api.pk_bytes(k), api.sk_bytes(k)]


def test_second_level_struct_sizes():
    """sizeof(mpcith_randomness), sizeof(mpcith_range_proof), sizeof(mlwe_inst) per KYBER_K."""
    from mpcith_kyber_kosk_amd import api
    want = {2: (950400, 163072, 5120), 3: (975744, 174720, 9216), 4: (1001088, 232960, 14336)}
    for k, (r, g, i) in want.items():
        assert api.lib.kosk_randomness_bytes(k) == r
        assert api.lib.kosk_range_proof_bytes(k) == g
        assert api.lib.kosk_mlwe_inst_bytes(k) == i
    assert api.lib.kosk_randomness_bytes(5) == 0


def test_compact_codec_roundtrip_on_oracle_proof(oracle):
    """Host codec of the compact wire format: sizes, lossless on a real proof, values >= 4096 refused."""
    import ctypes as C
    from mpcith_kyber_kosk_amd import api
    want = {2: 664340, 3: 680980, 4: 744148}
    for k in (2, 3, 4):
        cb = api.lib.kosk_compact_proof_bytes(k)
        assert cb % 16 == 0 and 0.77 * want[k] < cb < 0.79 * want[k]
    k = 2
    pi = oracle.verifiable_keygen(k, oracle.tape_bytes_for(k, 3))[2]
    out = C.create_string_buffer(api.lib.kosk_compact_proof_bytes(k))
    assert api.lib.kosk_proof_compress(k, pi, out) == 0
    back = C.create_string_buffer(len(pi))
    assert api.lib.kosk_proof_decompress(k, out, back) == 0
    assert back.raw == pi
    bad = bytearray(pi); bad[1] = 0x10
    assert api.lib.kosk_proof_compress(k, bytes(bad), out) == -1


def test_float_reciprocal_reduction_is_exact_for_every_u32(tmp_path):
    """The device's mod-q reduction of the MFMA epilogues (gf_reduce_u32, csrc/kosk_limb_dev.hpp) converts to fp32, multiplies by a
    rounded (1 - 2^-22) / q and truncates: tools/float_reduce_check.c runs the same IEEE operations over all 2^32 inputs."""
    import re
    src = os.path.join(ROOT, "tools", "float_reduce_check.c")
    hdr = open(os.path.join(ROOT, "mpcith_kyber_kosk_amd", "csrc", "kosk_limb_dev.hpp")).read()
    const = re.search(r"\(float\)x \* (0x[0-9a-f.]+p-12f)", hdr).group(1)
    assert const in open(src).read(), "the checked constant is not the kernel's"
    exe = str(tmp_path / "frc")
    subprocess.check_call(["gcc", "-O2", "-o", exe, src])
    out = subprocess.run([exe], stdout=subprocess.PIPE, text=True, timeout=300).stdout
    assert "bad=0" in out and int(re.search(r"maxr=(\d+)", out).group(1)) < 2 * 3329, out
