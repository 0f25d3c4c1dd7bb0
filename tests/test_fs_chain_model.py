"""CPU: the lane-level model of the wave-cooperative sponge (tools/fs_chain_model.py -- the per-lane tables, the two LDS exchanges and
the bit-interleaved rotations of csrc/kosk_fs_dev.hpp, restated on numpy arrays of 64 lanes) against hashlib; the options struct of
kosk_create_ex; kosk_create_ex without a GPU."""
import ctypes as C
import hashlib
import os
import re
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_lane_model_matches_hashlib():
    import fs_chain_model as m
    assert m.self_check()


def test_lane_model_of_the_bpermute_variant_matches_hashlib():
    """variant B of csrc/kosk_fs_dev.hpp (the default): five-lane columns inside 16-lane rows, column sums by three DPP row shifts, both
    exchanges as ds_bpermute gathers"""
    import fs_chain_model as m
    assert m.self_check_b()
    t = m.tables_b()
    act = t["act"] == 1
    assert act.sum() == 50
    for key in ("sCm", "sCp", "s0", "s1", "s2"):  # an active lane never gathers from an idle one
        assert act[t[key][act]].all(), key
    # a column's five lanes share a 16-lane row (the DPP row shifts never cross one)
    for h in range(2):
        for x in range(5):
            assert len({m.lane_b(x, y, h) >> 4 for y in range(5)}) == 1


def test_lane_model_of_gen_matrix_on_the_wave_sponge():
    """csrc/kosk_keygen_wave_dev.hpp (kw_gen_matrix) lane by lane: the seed hash whose digest words stay in their lanes as the XOF's first
    words, the padding words of SHAKE128's 168-byte rate, and the one-step rejection parse (56 lanes x one 3-byte group, two ballots)
    against hashlib + the scalar rej_uniform of kyber/indcpa.c:124-145 -- K = 2, 3, 4, several (i, j), and the block guard"""
    import fs_chain_model as m
    assert m.self_check_gen_matrix()


def test_lane_tables_match_the_device_header():
    """the LDS map constants of csrc/kosk_fs_dev.hpp are the model's"""
    import fs_chain_model as m
    hdr = open(os.path.join(ROOT, "mpcith_kyber_kosk_amd", "csrc", "kosk_fs_dev.hpp")).read()
    got = {k: int(v) for k, v in re.findall(r"constexpr int (FSW_[A-Z]+) = (\d+);", hdr)}
    assert got == {"FSW_T": m.T_OFF, "FSW_B": m.B_OFF, "FSW_ZERO": m.ZERO_OFF, "FSW_JUNK": m.JUNK_OFF, "FSW_WORDS": m.LDS_WORDS}
    t = m.tables()
    act = t["act"] == 1
    assert act.sum() == 50
    # 16-byte reads of the theta exchange are 16-byte aligned; the zero pad is too
    assert not (t["rTm"] % 4).any() and not (t["rTp"] % 4).any() and m.ZERO_OFF % 4 == 0
    # nobody but the idle lanes touches the zero pad or the junk words; every pi destination (and its ghost) is its own word
    for key in ("wT", "wB"):
        assert (t[key][act] < m.ZERO_OFF).all() and (t[key][~act] >= m.JUNK_OFF).all()
    assert (t["wB"][~act] + 5 < m.LDS_WORDS).all()
    assert len(set(t["wB"][act])) == 50 and len(set(t["wB"][act] + 5)) == 50 and not set(t["wB"][act]) & set(t["wB"][act] + 5)
    # a chi read never reaches a scratch word: words x .. x + 2 of a row of (5 values, 2 ghosts, 3 scratch)
    assert (((t["rB"][act] - m.B_OFF) % 10) + 2 <= 6).all()


def test_interleaved_padding_words():
    """the constants the kernel pads with: SHAKE256-PRF(key, nonce 1) absorbs 33 bytes -- word 4 = 0x1F01, word 16 = 0x80 << 56"""
    import fs_chain_model as m
    key = bytes(range(32))
    w = m.Wave()
    w.absorb_words([int.from_bytes(key[8 * i:8 * i + 8], "little") for i in range(4)] + [0x1F01] + [0] * 11 + [0x80 << 56])
    w.permute()
    out = b"".join(w.word(i).to_bytes(8, "little") for i in range(17))
    assert out == hashlib.shake_256(key + b"\x01").digest(136)


def test_options_struct_defaults_and_size():
    from mpcith_kyber_kosk_amd import api
    o = api.options()
    assert o.size == C.sizeof(api.KoskOptions) == 68
    assert (o.streams, o.combine, o.host_threads) == (0, 0, 0)
    assert (o.combine_wait_us, o.combine_idle_us, o.combine_prewake_us, o.strict_encoding, o.fs_mode, o.blocking_sync, o.hooks_unmerged) == (-1,) * 7
    assert list(o.reserved) == [0] * 6
    with pytest.raises(api.KoskError):
        api.options(no_such_field=1)
    # the header declares the same fields in the same order
    hdr = open(os.path.join(ROOT, "include", "kosk_mi355x.h")).read()
    body = re.search(r"typedef struct kosk_options \{(.*?)\} kosk_options;", hdr, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = re.findall(r"(?:uint32_t|int32_t)\s+([a-z_]+)(?:\[\d+\])?;", body)
    assert names == [f[0] for f in api.KoskOptions._fields_]


def test_create_ex_without_gpu_or_with_bad_options_fails_loudly():
    import torch
    from mpcith_kyber_kosk_amd import api
    h = C.c_void_p()
    bad = api.KoskOptions()  # size never set
    assert api.lib.kosk_create_ex(C.byref(h), 0, 3, 1, C.byref(bad)) != 0 and not h.value
    assert b"options.size" in api.lib.kosk_last_error(None)
    if not torch.cuda.is_available():
        with pytest.raises(api.KoskError, match="no HIP device|no CPU fallback|hip"):
            api.Kosk(kyber_k=2, max_batch=1, fs_mode=api.FS_DEVICE)
