"""GPU parity of the individual kernels (through the C-ABI, device pointers) vs hashlib / the oracle."""
import hashlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


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


@pytest.mark.parametrize("length", [0, 1, 31, 32, 33, 135, 136, 137, 271, 272, 320, 472, 1000])
def test_sha3_256_and_shake256_message_major(length, torch, ctx):
    n = 257  # ragged: not a multiple of the wave size
    rng = np.random.default_rng(length)
    stride = max(8, (length + 7) // 8 * 8)
    msgs = rng.integers(0, 256, size=(n, stride), dtype=np.uint8)
    d_in = _dev(torch, msgs)
    d_out = torch.zeros((n, 32), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()  # torch fills / copies run on the null stream; the library streams are not ordered against it
    ctx.sha3_256_batch(d_in.data_ptr(), stride, length, d_out.data_ptr(), n)
    d_x = torch.zeros((n, 200), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()  # torch fills / copies run on the null stream; the library streams are not ordered against it
    ctx.shake256_batch(d_in.data_ptr(), stride, length, d_x.data_ptr(), 200, n)
    ctx.synchronize()
    out, xof = d_out.cpu().numpy(), d_x.cpu().numpy()
    for i in range(n):
        m = msgs[i, :length].tobytes()
        assert out[i].tobytes() == hashlib.sha3_256(m).digest()
        assert xof[i].tobytes() == hashlib.shake_256(m).digest(200)


@pytest.mark.parametrize("length", [0, 1, 3, 4, 5, 135, 136, 137, 271, 272, 320, 472, 1000])
def test_sha3_256_lane_pair_sponge(length, torch, ctx):
    """kosk_sha3_256_batch_pair: the lane-pair ("warp-cooperative") Keccak layout, 32 messages per wave, ragged counts (idle
    pairs in the last wave, a single message, one more than a wave) against hashlib."""
    rng = np.random.default_rng(1000 + length)
    stride = max(8, (length + 7) // 8 * 8)
    for n in (257, 1, 33):
        msgs = rng.integers(0, 256, size=(n, stride), dtype=np.uint8)
        d_in = _dev(torch, msgs)
        d_out = torch.zeros((n + 1, 32), dtype=torch.uint8, device="cuda")   # one guard row behind the last digest
        torch.cuda.synchronize()  # torch fills / copies run on the null stream; the library streams are not ordered against it
        ctx.sha3_256_batch_pair(d_in.data_ptr(), stride, length, d_out.data_ptr(), n)
        ctx.synchronize()
        out = d_out.cpu().numpy()
        for i in range(n):
            assert out[i].tobytes() == hashlib.sha3_256(msgs[i, :length].tobytes()).digest(), (n, i)
        assert not out[n].any()


@pytest.mark.parametrize("k", [2, 3, 4])
@pytest.mark.parametrize("with_prefix", [0, 1])
def test_commit_hash_column_layout(k, with_prefix, torch, oracle):
    from mpcith_kyber_kosk_amd import api
    p = oracle.params(k)
    words = (p.view_msg_bytes - 32) // 2 if with_prefix else p.tcomm_msg_bytes // 2
    n = 1454 + 77  # ragged lane count
    rng = np.random.default_rng(100 * k + with_prefix)
    stride = 1600
    rows = rng.integers(0, 3329, size=(words, stride), dtype=np.uint16)
    prefix = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    c = api.Kosk(kyber_k=k, max_batch=1)
    d_rows, d_pre = _dev(torch, rows), _dev(torch, prefix)
    d_out = torch.zeros((n, 32), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()  # torch fills / copies run on the null stream; the library streams are not ordered against it
    c.commit_hash_lanes(d_rows.data_ptr(), stride, n, d_pre.data_ptr(), with_prefix, d_out.data_ptr())
    c.synchronize()
    out = d_out.cpu().numpy()
    for l in list(range(0, n, 97)) + [n - 1]:
        msg = (prefix[l].tobytes() if with_prefix else b"") + rows[:, l].astype("<u2").tobytes()
        assert out[l].tobytes() == hashlib.sha3_256(msg).digest(), l
    c.close()


def test_ntt256_matches_oracle(torch, ctx, oracle):
    rng = np.random.default_rng(7)
    n = 1000 + 3  # not a multiple of 16 polynomials per workgroup
    a = rng.integers(-3328, 3329, size=(n, 256), dtype=np.int16)
    a[0] = 3328; a[1] = -3328; a[2] = 0           # extremes of the reference's input range
    a[3] = np.arange(256) % 3329
    d_in = _dev(torch, a)
    d_out = torch.zeros_like(d_in)
    torch.cuda.synchronize()  # torch fills / copies run on the null stream; the library streams are not ordered against it
    ctx.ntt256_batch(d_in.data_ptr(), d_out.data_ptr(), n)
    ctx.synchronize()
    out = d_out.cpu().numpy()
    for i in list(range(0, n, 37)) + [0, 1, 2, 3, n - 1]:
        assert np.array_equal(out[i], oracle.poly_ntt(a[i])), i
    assert out.min() >= -1664 and out.max() <= 1664



@pytest.mark.parametrize("n", [300, 9982])  # 9 982 rows = 46 proofs: every wave walks 10-11 table chunks (the pipelined epilogue's steady state)
def test_lagrange_expand_and_recon_match_oracle(n, torch, ctx, oracle):
    rng = np.random.default_rng(11)
    y = rng.integers(0, 3329, size=(n, 407), dtype=np.uint16)
    y[0] = 0; y[1] = 3328
    y[2] = 1664; y[3] = 1665  # the largest centred magnitudes: the limb products' sums are at their extremes for a constant row
    y[4, 0::2] = 1664; y[4, 1::2] = 1665
    d_y = _dev(torch, y)
    d_sh = torch.zeros((n, 1454), dtype=torch.int16, device="cuda")
    torch.cuda.synchronize()  # torch fills / copies run on the null stream; the library streams are not ordered against it
    ctx.lagrange_expand(d_y.data_ptr(), d_sh.data_ptr(), n)
    d_sec = torch.zeros((n, 256), dtype=torch.int16, device="cuda")
    torch.cuda.synchronize()  # torch fills / copies run on the null stream; the library streams are not ordered against it
    ctx.recon_secrets(d_sh.data_ptr(), d_sec.data_ptr(), n, False)
    ctx.synchronize()
    sh = d_sh.cpu().numpy().view(np.uint16)
    sec = d_sec.cpu().numpy().view(np.uint16)
    for i in list(range(0, n, 29 if n < 1000 else 499)) + [0, 1, 2, 3, 4, n - 1]:
        assert np.array_equal(sh[i], oracle.recompute_shares(y[i])), i
    # encode -> erase -> decode round trip on every row: the packed secrets come back
    assert np.array_equal(sec, y[:, :256])
    # degree-2d reconstruction of share-wise products = product of the packed secrets
    prod = (sh.astype(np.uint32)[0::2] * sh.astype(np.uint32)[1::2] % 3329).astype(np.uint16)
    d_p = _dev(torch, prod)
    d_s2 = torch.zeros((prod.shape[0], 256), dtype=torch.int16, device="cuda")
    torch.cuda.synchronize()  # torch fills / copies run on the null stream; the library streams are not ordered against it
    ctx.recon_secrets(d_p.data_ptr(), d_s2.data_ptr(), prod.shape[0], True)
    ctx.synchronize()
    s2 = d_s2.cpu().numpy().view(np.uint16)
    exp = (y[0::2, :256].astype(np.uint32) * y[1::2, :256].astype(np.uint32) % 3329).astype(np.uint16)
    assert np.array_equal(s2, exp)
    assert np.array_equal(s2[3], oracle.recon(prod[3], True))
