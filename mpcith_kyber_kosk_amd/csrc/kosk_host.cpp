// Host-side crypto and helpers of the product path (see kosk_host.hpp).
#include "kosk_host.hpp"

#include <linux/futex.h>
#include <sched.h>
#include <sys/random.h>
#include <sys/syscall.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <mutex>
#include <stdexcept>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <thread>

#include "kosk_math.hpp"

namespace kosk {

// ------------------------------------------------------------------ sponge --
namespace {

inline uint64_t load64(const uint8_t *p)
{
    uint64_t v;
    memcpy(&v, p, 8); // little-endian host (x86-64)
    return v;
}

struct Sponge {
    uint64_t s[25];
    size_t rate;
    Sponge(size_t r) : rate(r) { memset(s, 0, sizeof s); }
    void absorb_once(const uint8_t *in, size_t len, uint8_t dom)
    {
        while (len >= rate) {
            for (size_t i = 0; i < rate / 8; i++) s[i] ^= load64(in + 8 * i);
            keccak_f1600(s);
            in += rate;
            len -= rate;
        }
        uint8_t last[200];
        memset(last, 0, rate);
        memcpy(last, in, len);
        last[len] = dom;
        last[rate - 1] |= 0x80;
        for (size_t i = 0; i < rate / 8; i++) s[i] ^= load64(last + 8 * i);
    }
    void squeeze(uint8_t *out, size_t len)
    {
        while (len) {
            keccak_f1600(s);
            const size_t n = len < rate ? len : rate;
            memcpy(out, s, n);
            out += n;
            len -= n;
        }
    }
};

} // namespace

void sha3_256(uint8_t out[32], const uint8_t *in, size_t inlen) { Sponge k(136); k.absorb_once(in, inlen, 0x06); k.squeeze(out, 32); }
void sha3_512(uint8_t out[64], const uint8_t *in, size_t inlen) { Sponge k(72); k.absorb_once(in, inlen, 0x06); k.squeeze(out, 64); }
void shake128(uint8_t *out, size_t outlen, const uint8_t *in, size_t inlen) { Sponge k(168); k.absorb_once(in, inlen, 0x1F); k.squeeze(out, outlen); }
void shake256(uint8_t *out, size_t outlen, const uint8_t *in, size_t inlen) { Sponge k(136); k.absorb_once(in, inlen, 0x1F); k.squeeze(out, outlen); }
void shake256_prf(uint8_t *out, size_t outlen, const uint8_t key[32], uint8_t nonce)
{
    uint8_t ext[33];
    memcpy(ext, key, 32);
    ext[32] = nonce;
    shake256(out, outlen, ext, 33);
}

// ------------------------------------------------------------ Kyber keygen --
namespace {

// ntt.c:80-95 followed by poly_reduce (poly.c:261-265)
void poly_ntt(int16_t *r)
{
    int k = 1;
    for (int len = 128; len >= 2; len >>= 1)
        for (int start = 0; start < 256; start += 2 * len) {
            const int32_t z = kZetas.z[k++];
            for (int j = start; j < start + len; j++) {
                const int32_t t = fqmul(z, r[j + len]);
                r[j + len] = (int16_t)(r[j] - t);
                r[j] = (int16_t)(r[j] + t);
            }
        }
    for (int j = 0; j < 256; j++) r[j] = (int16_t)barrett_reduce(r[j]);
}

// polyvec.c:202-214 + poly.c:307-313
void matvec_row(int16_t *r, const int16_t *Arow, const int16_t *v, int K)
{
    constexpr int32_t f = (int32_t)((1ULL << 32) % Q);
    for (int p = 0; p < 128; p++) {
        const int32_t zeta = (p & 1) ? -(int32_t)kZetas.z[64 + (p >> 1)] : (int32_t)kZetas.z[64 + (p >> 1)];
        int32_t r0 = 0, r1 = 0;
        for (int l = 0; l < K; l++) {
            const int32_t a0 = Arow[l * 256 + 2 * p], a1 = Arow[l * 256 + 2 * p + 1];
            const int32_t b0 = v[l * 256 + 2 * p], b1 = v[l * 256 + 2 * p + 1];
            r0 += fqmul(fqmul(a1, b1), zeta) + fqmul(a0, b0);
            r1 += fqmul(a0, b1) + fqmul(a1, b0);
        }
        r[2 * p] = (int16_t)montgomery_reduce(barrett_reduce((int16_t)r0) * f);
        r[2 * p + 1] = (int16_t)montgomery_reduce(barrett_reduce((int16_t)r1) * f);
    }
}

// poly.c:124-139
void poly_tobytes(uint8_t *r, const int16_t *a)
{
    for (int i = 0; i < 128; i++) {
        const uint16_t t0 = (uint16_t)(a[2 * i] + ((a[2 * i] >> 15) & Q));
        const uint16_t t1 = (uint16_t)(a[2 * i + 1] + ((a[2 * i + 1] >> 15) & Q));
        r[3 * i] = (uint8_t)t0;
        r[3 * i + 1] = (uint8_t)((t0 >> 8) | (t1 << 4));
        r[3 * i + 2] = (uint8_t)(t1 >> 4);
    }
}

// indcpa.c:168-193 (gen_matrix, not transposed) with rej_uniform :124-145
void gen_matrix(int16_t *A, const uint8_t seed[32], int K)
{
    uint8_t ext[34];
    std::vector<uint8_t> buf(168 * 4);
    memcpy(ext, seed, 32);
    for (int i = 0; i < K; i++)
        for (int j = 0; j < K; j++) {
            ext[32] = (uint8_t)j;
            ext[33] = (uint8_t)i;
            size_t have = 168 * 4;
            shake128(buf.data(), have, ext, 34);
            int16_t *r = A + ((size_t)i * K + j) * 256;
            int ctr = 0;
            size_t pos = 0;
            for (;;) {
                for (; ctr < 256 && pos + 3 <= have; pos += 3) {
                    const uint16_t v0 = ((buf[pos] >> 0) | ((uint16_t)buf[pos + 1] << 8)) & 0xFFF;
                    const uint16_t v1 = ((buf[pos + 1] >> 4) | ((uint16_t)buf[pos + 2] << 4)) & 0xFFF;
                    if (v0 < Q) r[ctr++] = (int16_t)v0;
                    if (ctr < 256 && v1 < Q) r[ctr++] = (int16_t)v1;
                }
                if (ctr == 256) break;
                // like the reference (indcpa.c:139-144) keep squeezing, without a bound: the XOF output is a prefix-consistent
                // byte stream, so squeezing a longer prefix again continues where the parser stopped (168 = 56 whole triples)
                have += 168 * 4;
                if (buf.size() < have) buf.resize(have);
                shake128(buf.data(), have, ext, 34);
            }
        }
}

// cbd.c:58-107
void cbd(int16_t *r, const uint8_t *buf, int eta)
{
    if (eta == 2) {
        for (int i = 0; i < 32; i++) {
            uint32_t t;
            memcpy(&t, buf + 4 * i, 4);
            const uint32_t d = (t & 0x55555555u) + ((t >> 1) & 0x55555555u);
            for (int j = 0; j < 8; j++) r[8 * i + j] = (int16_t)(((d >> (4 * j)) & 3) - ((d >> (4 * j + 2)) & 3));
        }
    } else {
        for (int i = 0; i < 64; i++) {
            const uint32_t t = buf[3 * i] | ((uint32_t)buf[3 * i + 1] << 8) | ((uint32_t)buf[3 * i + 2] << 16);
            const uint32_t d = (t & 0x00249249u) + ((t >> 1) & 0x00249249u) + ((t >> 2) & 0x00249249u);
            for (int j = 0; j < 4; j++) r[4 * i + j] = (int16_t)(((d >> (6 * j)) & 7) - ((d >> (6 * j + 3)) & 7));
        }
    }
}

} // namespace

void host_keygen(const Params &P, const uint8_t seed64[64], uint8_t *pk, uint8_t *sk, HostKey &key)
{
    const int K = P.K;
    uint8_t buf[64], in[33];
    memcpy(in, seed64, 32);
    in[32] = (uint8_t)K;
    sha3_512(buf, in, 33); // kosk.cpp:12-14
    const uint8_t *public_seed = buf, *noise_seed = buf + 32;
    gen_matrix(key.A, public_seed, K);
    uint8_t nb[192];
    for (int i = 0; i < 2 * K; i++) { // nonce 0..K-1: s, K..2K-1: e   kosk.cpp:17-20
        shake256_prf(nb, (size_t)P.eta1 * 64, noise_seed, (uint8_t)i);
        cbd(key.se + 256 * i, nb, P.eta1);
    }
    int16_t shat[MAXK * 256], ehat[MAXK * 256];
    memcpy(shat, key.se, sizeof(int16_t) * 256 * K);
    memcpy(ehat, key.se + 256 * K, sizeof(int16_t) * 256 * K);
    for (int i = 0; i < K; i++) {
        poly_ntt(shat + 256 * i);
        poly_ntt(ehat + 256 * i);
    }
    for (int i = 0; i < K; i++) { // kosk.cpp:42-48
        int16_t *ti = key.t + 256 * i;
        matvec_row(ti, key.A + (size_t)i * K * 256, shat, K);
        for (int j = 0; j < 256; j++) ti[j] = (int16_t)barrett_reduce((int16_t)(ti[j] + ehat[256 * i + j]));
        poly_tobytes(pk + 384 * i, ti);
    }
    memcpy(pk + 384 * K, public_seed, 32);
    for (int i = 0; i < K; i++) poly_tobytes(sk + 384 * i, shat + 256 * i);
    memcpy(sk + 384 * K, pk, P.pk_bytes);
    sha3_256(sk + P.sk_bytes - 64, pk, P.pk_bytes);
    memcpy(sk + P.sk_bytes - 32, noise_seed, 32); // kosk.cpp:67-69: z is the noise seed
}

void host_decode_pk(const Params &P, const uint8_t *pk, HostKey &key)
{
    const int K = P.K;
    for (int i = 0; i < K; i++) // poly.c:151-158
        for (int j = 0; j < 128; j++) {
            const uint8_t *a = pk + 384 * i + 3 * j;
            key.t[256 * i + 2 * j] = (int16_t)(((a[0] >> 0) | ((uint16_t)a[1] << 8)) & 0xFFF);
            key.t[256 * i + 2 * j + 1] = (int16_t)(((a[1] >> 4) | ((uint16_t)a[2] << 4)) & 0xFFF);
        }
    gen_matrix(key.A, pk + 384 * K, K);
}

// -------------------------------------------------------------- Fiat-Shamir --
void fs_alpha(const Params &P, const uint8_t *tcomm_all, uint16_t *alpha)
{
    uint8_t h[32], a_[2 * MAXJ];
    sha3_256(h, tcomm_all, (size_t)NPARTY * 32);
    shake256_prf(a_, (size_t)2 * P.J, h, 1);
    for (int i = 0; i < P.J; i++) alpha[i] = (uint16_t)(((a_[2 * i] << 8) | a_[2 * i + 1]) % Q);
}

static void opened_from_ch(const uint8_t ch[32], uint16_t I[NOPEN], uint16_t rest[NREST]);
void fs_opened(const uint8_t *digests_all, uint16_t I[NOPEN], uint16_t rest[NREST])
{
    uint8_t ch[32];
    sha3_256(ch, digests_all, (size_t)NPARTY * 32);
    opened_from_ch(ch, I, rest);
}

// ------------------------------------------------- multi-buffer SHA3-256 (host) --
namespace {

template <int W>
struct VecT;
template <> struct VecT<4> { typedef uint64_t type __attribute__((vector_size(32))); };
template <> struct VecT<8> { typedef uint64_t type __attribute__((vector_size(64))); };

template <int W, int N>
__attribute__((always_inline)) static inline typename VecT<W>::type vrot(typename VecT<W>::type v)
{
    if constexpr (N == 0) return v;
    else return (v << N) | (v >> (64 - N));
}

template <int W, int X, int Y>
__attribute__((always_inline)) static inline void vrhopi(const typename VecT<W>::type *a, const typename VecT<W>::type *d, typename VecT<W>::type *b)
{
    b[Y + 5 * ((2 * X + 3 * Y) % 5)] = vrot<W, kRho[X + 5 * Y]>(a[X + 5 * Y] ^ d[X]);
}

template <int W>
__attribute__((always_inline)) static inline void vround(const typename VecT<W>::type *a, typename VecT<W>::type *o, uint64_t rc)
{
    typedef typename VecT<W>::type V;
    V c[5], d[5], b[25];
    for (int x = 0; x < 5; x++) c[x] = a[x] ^ a[x + 5] ^ a[x + 10] ^ a[x + 15] ^ a[x + 20];
    for (int x = 0; x < 5; x++) d[x] = c[(x + 4) % 5] ^ vrot<W, 1>(c[(x + 1) % 5]);
    vrhopi<W, 0, 0>(a, d, b); vrhopi<W, 1, 0>(a, d, b); vrhopi<W, 2, 0>(a, d, b); vrhopi<W, 3, 0>(a, d, b); vrhopi<W, 4, 0>(a, d, b);
    vrhopi<W, 0, 1>(a, d, b); vrhopi<W, 1, 1>(a, d, b); vrhopi<W, 2, 1>(a, d, b); vrhopi<W, 3, 1>(a, d, b); vrhopi<W, 4, 1>(a, d, b);
    vrhopi<W, 0, 2>(a, d, b); vrhopi<W, 1, 2>(a, d, b); vrhopi<W, 2, 2>(a, d, b); vrhopi<W, 3, 2>(a, d, b); vrhopi<W, 4, 2>(a, d, b);
    vrhopi<W, 0, 3>(a, d, b); vrhopi<W, 1, 3>(a, d, b); vrhopi<W, 2, 3>(a, d, b); vrhopi<W, 3, 3>(a, d, b); vrhopi<W, 4, 3>(a, d, b);
    vrhopi<W, 0, 4>(a, d, b); vrhopi<W, 1, 4>(a, d, b); vrhopi<W, 2, 4>(a, d, b); vrhopi<W, 3, 4>(a, d, b); vrhopi<W, 4, 4>(a, d, b);
    for (int y = 0; y < 25; y += 5)
        for (int x = 0; x < 5; x++) o[y + x] = b[y + x] ^ (~b[y + (x + 1) % 5] & b[y + (x + 2) % 5]);
    V r;
    for (int i = 0; i < W; i++) r[i] = rc;
    o[0] ^= r;
}

// W messages of `len` bytes, all of the same length, SHA3-256
template <int W>
__attribute__((always_inline)) static inline void sha3_256_xw(uint8_t *const *out, const uint8_t *const *in, size_t len)
{
    typedef typename VecT<W>::type V;
    V s[25], t[25];
    for (auto &v : s) v = V{};
    size_t off = 0;
    // (a macro, not a lambda: a lambda body would not inherit the caller's target("avx...") attribute)
#define KOSK_PERMUTE()                          \
    for (int r = 0; r < 24; r += 2) {           \
        vround<W>(s, t, kKeccak.rc[r]);         \
        vround<W>(t, s, kKeccak.rc[r + 1]);     \
    }
    while (len - off >= 136) {
        for (int w = 0; w < 17; w++) {
            V v;
            for (int i = 0; i < W; i++) v[i] = load64(in[i] + off + 8 * w);
            s[w] ^= v;
        }
        KOSK_PERMUTE();
        off += 136;
    }
    uint8_t last[W][136];
    for (int i = 0; i < W; i++) {
        memset(last[i], 0, 136);
        memcpy(last[i], in[i] + off, len - off);
        last[i][len - off] = 0x06;
        last[i][135] |= 0x80;
    }
    for (int w = 0; w < 17; w++) {
        V v;
        for (int i = 0; i < W; i++) v[i] = load64(last[i] + 8 * w);
        s[w] ^= v;
    }
    KOSK_PERMUTE();
#undef KOSK_PERMUTE
    for (int i = 0; i < W; i++)
        for (int w = 0; w < 4; w++) {
            const uint64_t x = s[w][i];
            memcpy(out[i] + 8 * w, &x, 8);
        }
}

// Eight messages of equal length, SHA3-256, AVX-512: the state's 25 lanes stay in 25 vector variables for the whole message and the
// 24 rounds are an IN-PLACE schedule (kosk_keccak_x8_rounds.inc, generated by tools/gen_keccak_x8.py): chi writes a row of the new
// state into the registers whose old lanes it has just consumed, so there is no second copy of the state to spill (the generic
// template above needs 50 vectors plus temporaries for the 32 registers: a third of its instructions were moves to and from the
// stack).  Three-way XOR and chi are one vpternlogq each.  The Fiat-Shamir rounds of a step hash 186 KB per proof on the host: this
// function is the largest single consumer of the host's CPU share.
#include <immintrin.h>
__attribute__((target("avx512f"))) void sha3_256_x8_avx512(uint8_t *const *out, const uint8_t *const *in, size_t len)
{
    const __m512i zero = _mm512_setzero_si512();
    __m512i s0 = zero, s1 = zero, s2 = zero, s3 = zero, s4 = zero, s5 = zero, s6 = zero, s7 = zero, s8 = zero, s9 = zero, s10 = zero, s11 = zero,
            s12 = zero, s13 = zero, s14 = zero, s15 = zero, s16 = zero, s17 = zero, s18 = zero, s19 = zero, s20 = zero, s21 = zero, s22 = zero,
            s23 = zero, s24 = zero;
    __m512i C0, C1, C2, C3, C4, D0, D1, D2, D3, D4;
#define KTHETA(x, a, b, c, d, e) C##x = _mm512_ternarylogic_epi64(_mm512_ternarylogic_epi64(a, b, c, 0x96), d, e, 0x96);
#define KD()                                            \
    D0 = _mm512_xor_si512(C4, _mm512_rol_epi64(C1, 1)); \
    D1 = _mm512_xor_si512(C0, _mm512_rol_epi64(C2, 1)); \
    D2 = _mm512_xor_si512(C1, _mm512_rol_epi64(C3, 1)); \
    D3 = _mm512_xor_si512(C2, _mm512_rol_epi64(C4, 1)); \
    D4 = _mm512_xor_si512(C3, _mm512_rol_epi64(C0, 1));
#define KROW(a0, d0, r0, a1, d1, r1, a2, d2, r2, a3, d3, r3, a4, d4, r4)        \
    {                                                                           \
        const __m512i t0 = _mm512_rol_epi64(_mm512_xor_si512(a0, D##d0), r0);    \
        const __m512i t1 = _mm512_rol_epi64(_mm512_xor_si512(a1, D##d1), r1);    \
        const __m512i t2 = _mm512_rol_epi64(_mm512_xor_si512(a2, D##d2), r2);    \
        const __m512i t3 = _mm512_rol_epi64(_mm512_xor_si512(a3, D##d3), r3);    \
        const __m512i t4 = _mm512_rol_epi64(_mm512_xor_si512(a4, D##d4), r4);    \
        a0 = _mm512_ternarylogic_epi64(t0, t1, t2, 0xD2); /* t0 ^ (~t1 & t2) */  \
        a1 = _mm512_ternarylogic_epi64(t1, t2, t3, 0xD2);                        \
        a2 = _mm512_ternarylogic_epi64(t2, t3, t4, 0xD2);                        \
        a3 = _mm512_ternarylogic_epi64(t3, t4, t0, 0xD2);                        \
        a4 = _mm512_ternarylogic_epi64(t4, t0, t1, 0xD2);                        \
    }
#define KIOTA(a, r) a = _mm512_xor_si512(a, _mm512_set1_epi64((long long)kKeccak.rc[r]));
#define KABSORB(w, src, off) s##w = _mm512_xor_si512(s##w, _mm512_set_epi64((long long)load64(src[7] + (off) + 8 * w), (long long)load64(src[6] + (off) + 8 * w), \
        (long long)load64(src[5] + (off) + 8 * w), (long long)load64(src[4] + (off) + 8 * w), (long long)load64(src[3] + (off) + 8 * w),                          \
        (long long)load64(src[2] + (off) + 8 * w), (long long)load64(src[1] + (off) + 8 * w), (long long)load64(src[0] + (off) + 8 * w)));
#define KABSORB_BLOCK(src, off)                                                                                                     \
    KABSORB(0, src, off) KABSORB(1, src, off) KABSORB(2, src, off) KABSORB(3, src, off) KABSORB(4, src, off) KABSORB(5, src, off)   \
    KABSORB(6, src, off) KABSORB(7, src, off) KABSORB(8, src, off) KABSORB(9, src, off) KABSORB(10, src, off) KABSORB(11, src, off) \
    KABSORB(12, src, off) KABSORB(13, src, off) KABSORB(14, src, off) KABSORB(15, src, off) KABSORB(16, src, off)
    size_t off = 0;
    while (len - off >= 136) {
        KABSORB_BLOCK(in, off)
#include "kosk_keccak_x8_rounds.inc"
        off += 136;
    }
    uint8_t last[8][136];
    const uint8_t *lp[8];
    for (int i = 0; i < 8; i++) {
        memset(last[i], 0, 136);
        memcpy(last[i], in[i] + off, len - off);
        last[i][len - off] = 0x06;
        last[i][135] |= 0x80;
        lp[i] = last[i];
    }
    KABSORB_BLOCK(lp, 0)
#include "kosk_keccak_x8_rounds.inc"
    alignas(64) uint64_t w[4][8];
    _mm512_store_si512(w[0], s0);
    _mm512_store_si512(w[1], s1);
    _mm512_store_si512(w[2], s2);
    _mm512_store_si512(w[3], s3);
    for (int i = 0; i < 8; i++)
        for (int q = 0; q < 4; q++) memcpy(out[i] + 8 * q, &w[q][i], 8);
#undef KTHETA
#undef KD
#undef KROW
#undef KIOTA
#undef KABSORB
#undef KABSORB_BLOCK
}
__attribute__((target("avx2"))) void sha3_256_x4_avx2(uint8_t *const *out, const uint8_t *const *in, size_t len) { sha3_256_xw<4>(out, in, len); }
__attribute__((target("avx512f,avx512vl"))) void sha3_256_x4_avx512vl(uint8_t *const *out, const uint8_t *const *in, size_t len) { sha3_256_xw<4>(out, in, len); }

struct CpuCaps {
    bool avx2, avx512f, avx512vl;
    CpuCaps()
    {
        __builtin_cpu_init();
        avx2 = __builtin_cpu_supports("avx2");
        avx512f = __builtin_cpu_supports("avx512f");
        avx512vl = __builtin_cpu_supports("avx512vl");
        if (getenv("KOSK_HOST_SCALAR")) avx2 = avx512f = avx512vl = false;
    }
};
const CpuCaps &caps() { static const CpuCaps c; return c; }

// hash group `g` of width w (messages g*w .. g*w+w-1, the tail group re-hashes its first message as padding)
void sha3_group(uint8_t *out, const uint8_t *const *in, size_t len, int count, int w, int g)
{
    const uint8_t *ip[8];
    uint8_t *op[8];
    uint8_t dump[8][32];
    for (int i = 0; i < w; i++) {
        const int m = g * w + i;
        ip[i] = in[m < count ? m : g * w];
        op[i] = m < count ? out + 32 * (size_t)m : dump[i];
    }
    if (w == 8) sha3_256_x8_avx512(op, ip, len);
    else if (w == 4 && caps().avx512vl) sha3_256_x4_avx512vl(op, ip, len);
    else if (w == 4) sha3_256_x4_avx2(op, ip, len);
    else sha3_256(op[0], ip[0], len);
}

} // namespace

int sha3_multi_width() { return caps().avx512f ? 8 : caps().avx2 ? 4 : 1; }

void sha3_256_multi(uint8_t *out, const uint8_t *const *in, size_t len, int count)
{
    const int w = sha3_multi_width();
    for (int g = 0; g * w < count; g++) sha3_group(out, in, len, count, w, g);
}

// the SIMD width is chosen so that the groups roughly fill the available threads: the 343-permutation
// chains are sequential per proof, so latency = one chain whatever the width
// `then(b)` (optional) runs for every proof b of a group on the worker that has just hashed the group: the per-proof
// derivations (a SHAKE PRF and a few hundred scalar steps each) ride with the hashing instead of forming a serial tail of
// n of them on the calling thread (r3: that tail was ~60 us of each of the four Fiat-Shamir rounds of a 46-proof step)
static void sha3_digest_tables(int n, const uint8_t *digs, size_t dig_stride, uint8_t *out, int nthreads, Pool *pool,
                               const std::function<void(int)> *then = nullptr, const std::function<void(int)> *prep = nullptr)
{
    std::vector<const uint8_t *> in(n);
    for (int b = 0; b < n; b++) in[b] = digs + (size_t)b * dig_stride;
    int w = sha3_multi_width();
    static const int forced = getenv("KOSK_FS_WIDTH") ? atoi(getenv("KOSK_FS_WIDTH")) : 0;
    if (forced == 8 || forced == 4 || forced == 1) w = forced <= w ? forced : w;
    else if (w == 8 && n <= 4 * nthreads && n < 16) w = 4; // few proofs: shorter chains per group beat fewer groups
    if (w > 1 && n <= nthreads && !caps().avx512f) w = 1;
    const int groups = (n + w - 1) / w;
    parallel_for(pool, groups, nthreads, [&](int g) {
        if (prep)
            for (int b = g * w; b < n && b < (g + 1) * w; b++) (*prep)(b);
        sha3_group(out, in.data(), (size_t)NPARTY * 32, n, w, g);
        if (then)
            for (int b = g * w; b < n && b < (g + 1) * w; b++) (*then)(b);
    });
}

void assemble_digest_table(uint8_t *table, const uint16_t *I, const uint8_t *unopened, const uint8_t *opened)
{
    bool used[NPARTY] = {false};
    for (int i = 0; i < NOPEN; i++) {
        const int p = I[i];
        if (p >= NPARTY) continue;
        used[p] = true;
        memcpy(table + (size_t)p * 32, opened + (size_t)i * 32, 32);
    }
    int j = 0;
    for (int p = 0; p < NPARTY && j < NREST; p++)
        if (!used[p]) memcpy(table + (size_t)p * 32, unopened + (size_t)(j++) * 32, 32);
}

void fs_alpha_batch(const Params &P, int n, const uint8_t *digs, size_t dig_stride, uint16_t *alpha, size_t alpha_stride, int nthreads, Pool *pool,
                    const std::function<void(int)> *prep)
{
    std::vector<uint8_t> h((size_t)n * 32);
    const std::function<void(int)> derive = [&](int b) {
        uint8_t a_[2 * MAXJ];
        shake256_prf(a_, (size_t)2 * P.J, &h[(size_t)b * 32], 1);
        uint16_t *al = alpha + (size_t)b * alpha_stride;
        for (int i = 0; i < P.J; i++) al[i] = (uint16_t)(((a_[2 * i] << 8) | a_[2 * i + 1]) % Q);
    };
    sha3_digest_tables(n, digs, dig_stride, h.data(), nthreads, pool, &derive, prep);
}

static void opened_from_ch(const uint8_t ch[32], uint16_t I[NOPEN], uint16_t rest[NREST])
{
    uint8_t I_[2 * NOPEN];
    shake256_prf(I_, sizeof I_, ch, 1);
    bool used[NPARTY] = {false};
    for (int i = 0; i < NOPEN; i++) {
        // first candidate, then the reference's "+inc, rescan" probing: the smallest
        // inc >= 0 such that (I[i] + inc) % N is not among I[0..i)
        int v = ((I_[2 * i] << 8) | I_[2 * i + 1]) % NPARTY;
        while (used[v]) v = (v + 1) % NPARTY;
        used[v] = true;
        I[i] = (uint16_t)v;
    }
    for (int p = 0, j = 0; p < NPARTY; p++)
        if (!used[p]) rest[j++] = (uint16_t)p;
}

void fs_opened_batch(int n, const uint8_t *digs, size_t dig_stride, uint16_t *I, uint16_t *rest, size_t sel_stride, int nthreads, Pool *pool,
                     bool windows, const std::function<void(int)> *prep)
{
    std::vector<uint8_t> h((size_t)n * 32);
    const std::function<void(int)> derive = [&](int b) {
        uint16_t *Ib = I + (size_t)b * sel_stride, *rb = rest + (size_t)b * sel_stride;
        opened_from_ch(&h[(size_t)b * 32], Ib, rb);
        if (windows) { // complement entries owned by each aligned 64-party window (k_assemble_fields), stored behind the list I
            uint16_t *win = Ib + SEL_WIN;
            for (int w = 0, j = 0; w <= NWIN; w++) {
                while (j < NREST && rb[j] < 64 * w) j++;
                win[w] = (uint16_t)j;
            }
            // the opened parties ascending and where each sits in I (the grouped image kernel writes an opened party's records
            // from the window its column lies in)
            uint16_t pos[NPARTY];
            for (int p = 0; p < NPARTY; p++) pos[p] = 0xFFFF;
            for (int i = 0; i < NOPEN; i++) pos[Ib[i]] = (uint16_t)i;
            for (int p = 0, k = 0; p < NPARTY; p++)
                if (pos[p] != 0xFFFF) { Ib[SEL_OSORT + k] = (uint16_t)p; Ib[SEL_OPOS + k] = pos[p]; k++; }
        }
    };
    sha3_digest_tables(n, digs, dig_stride, h.data(), nthreads, pool, &derive, prep);
}

// ------------------------------------------------------------------- tables --
namespace {
struct InvTable {
    uint16_t inv[Q];
    InvTable()
    {
        inv[0] = 0;
        for (uint32_t a = 1; a < (uint32_t)Q; a++) {
            uint32_t r = 1, b = a, e = Q - 2;
            while (e) {
                if (e & 1) r = r * b % Q;
                b = b * b % Q;
                e >>= 1;
            }
            inv[a] = (uint16_t)r;
        }
    }
};
const InvTable &inv_table()
{
    static const InvTable t;
    return t;
}
} // namespace

uint16_t gf_inv_host(uint16_t a) { return inv_table().inv[a % Q]; }

void lagrange_row(uint16_t *row, int n, int a, int t)
{
    std::vector<uint32_t> pre(n + 1), suf(n + 1), fact(n + 1);
    fact[0] = 1;
    for (int i = 1; i <= n; i++) fact[i] = fact[i - 1] * (uint32_t)i % Q;
    auto diff = [&](int m) { return (uint32_t)((((t - a - m) % Q) + Q) % Q); };
    pre[0] = 1;
    for (int m = 0; m < n; m++) pre[m + 1] = pre[m] * diff(m) % Q;
    suf[n] = 1;
    for (int m = n - 1; m >= 0; m--) suf[m] = suf[m + 1] * diff(m) % Q;
    for (int j = 0; j < n; j++) {
        const uint32_t num = pre[j] * suf[j + 1] % Q;
        uint32_t den = fact[j] * fact[n - 1 - j] % Q; // prod_{m != j} (j - m) up to sign
        if ((n - 1 - j) & 1) den = (Q - den) % Q;
        row[j] = (uint16_t)(num * gf_inv_host((uint16_t)den) % Q);
    }
}

void pack_limb_table(const std::vector<uint16_t> &A, int M, int Kdim, int Mpad, int KS, std::vector<uint8_t> &out)
{
    const int RT = Mpad / 16;
    out.assign((size_t)KS * RT * 2048, 0);
    for (int m = 0; m < M; m++)
        for (int k = 0; k < Kdim; k++) {
            const int32_t c = gf_center(A[(size_t)m * Kdim + k]);
            const int c0 = ((c + 32) & 63) - 32, c1 = (c - c0) >> 6;
            const int ks = k >> 6, kc = (k >> 4) & 3, rr = m & 15;
            const size_t base = ((size_t)(ks * RT + (m >> 4)) * 2) * 1024 + rr * 64 + ((kc ^ ((-(rr >> 2)) & 3)) << 4) + (k & 15);
            out[base] = (uint8_t)(int8_t)c0;
            out[base + 1024] = (uint8_t)(int8_t)c1;
        }
}

// The same limb split in "fragment-linear" order for kernels that load MFMA operands straight from global memory: inside a
// 1 KiB tile, lane l = 16 (k % 64 / 16) + (m % 16) of v_mfma_i32_16x16x64_i8 finds its 16 bytes at offset 16 l, so a wave
// reads the tile as one linear, fully coalesced 1 KiB load.
void pack_frag_table(const std::vector<uint16_t> &A, int M, int Kdim, int Mpad, int KS, std::vector<uint8_t> &out)
{
    const int RT = Mpad / 16;
    out.assign((size_t)KS * RT * 2048, 0);
    for (int m = 0; m < M; m++)
        for (int k = 0; k < Kdim; k++) {
            const int32_t c = gf_center(A[(size_t)m * Kdim + k]);
            const int c0 = ((c + 32) & 63) - 32, c1 = (c - c0) >> 6;
            const int ks = k >> 6, kc = (k >> 4) & 3, rr = m & 15;
            const size_t base = ((size_t)(ks * RT + (m >> 4)) * 2) * 1024 + (size_t)(kc * 16 + rr) * 16 + (k & 15);
            out[base] = (uint8_t)(int8_t)c0;
            out[base + 1024] = (uint8_t)(int8_t)c1;
        }
}

// -------------------------------------------------------------------- misc --
// Persistent worker pool: Fiat-Shamir rounds arrive in short bursts between GPU phases, so the
// workers spin briefly on a generation counter before they block.
//
// Every parallel_for publishes its own Job object (index counter, completion counter, bounds, function): nothing a
// worker can touch is ever reset for the next job, so a worker that wakes late for job N either finds no job
// (cur_ == nullptr) or helps with job N+1 -- both are valid.  A Job lives on run()'s stack; run() retires it
// (cur_ = nullptr) and then waits until no worker is inside the pick-up window (inside_ == 0).  inside_++ / load cur_
// on the worker and store cur_ / load inside_ in run() are sequentially consistent (Dekker pattern): either the worker
// sees the job retired or run() sees the worker inside.
//
// Sleeping workers wait on the generation word itself (futex), not on a condition variable: waking 17 workers for a Fiat-Shamir
// round through notify_all made each of them take the pool's mutex in turn before it could look at the job (a queue of 17
// futex hand-offs in front of a 67 us hash chain); FUTEX_WAKE releases them in one system call and they share nothing on the
// way to the job but its index counter.  Only as many sleepers are woken as the job wants; who works is decided by a ticket
// (Job::joined), not by a worker's number, because the kernel chooses which sleepers wake.
class Pool {
public:
    static Pool &get() { static Pool p; return p; }
    Pool() = default;
    ~Pool()
    {
        stop_.store(true);
        gen_.fetch_add(1);
        wake(INT32_MAX);
        for (auto &t : th_) t.join();
    }
    void run(int n, int nthreads, const std::function<void(int)> &fn)
    {
        std::lock_guard<std::mutex> job_lock(job_mu_);
        grow(nthreads - 1);
        Job job{&fn, n, std::min<int>((int)th_.size(), nthreads - 1)};
        cur_.store(&job);
        gen_.fetch_add(1); // seq_cst: ordered against the sleepers' count (a worker counts itself, THEN the kernel compares gen_)
        if (sleepers_.load() > 0) wake(job.want);
        work(job);
        while (job.done.load(std::memory_order_acquire) < n) __builtin_ia32_pause();
        cur_.store(nullptr);
        // a worker may still hold &job between its pick-up and its first (failing) index fetch
        while (inside_.load() != 0) __builtin_ia32_pause();
        if (job.failed.load()) throw std::runtime_error("exception in a host worker job");
    }

private:
    struct Job {
        const std::function<void(int)> *fn;
        int n, want;
        std::atomic<int> next{0}, done{0}, joined{0};
        std::atomic<bool> failed{false};
        Job(const std::function<void(int)> *f, int n_, int w) : fn(f), n(n_), want(w) {}
    };
    // Workers are created up front (pool_create) and on demand; a failed thread creation (pid / thread limit, no memory) is
    // not an error: the job's indices are claimed dynamically, so the pool simply runs on the threads it has (the caller
    // always works too).  Nothing here may throw across run(): the C ABI above must never terminate its caller.
    void grow(int want) noexcept
    {
        while ((int)th_.size() < want && th_.size() < 255 && !grow_failed_) {
            const uint32_t g = gen_.load(); // never pick up a job published before we existed
            try {
                th_.emplace_back([this, g] { loop(g); });
            } catch (...) {
                grow_failed_ = true;
            }
        }
    }
    static void work(Job &j)
    {
        for (;;) {
            const int i = j.next.fetch_add(1, std::memory_order_relaxed);
            if (i >= j.n) break;
            try {
                (*j.fn)(i);
            } catch (...) {
                j.failed.store(true, std::memory_order_relaxed); // rethrown by run() on the calling thread once the job is over
            }
            j.done.fetch_add(1, std::memory_order_release);
        }
    }
    static int spin_us()
    {
        static const int v = getenv("KOSK_POOL_SPIN_US") ? atoi(getenv("KOSK_POOL_SPIN_US")) : 0; // round 5: 0 (was 20): same throughput, one to two busy cores fewer (profiles/r05_sweep_host.txt)
        return v;
    }
    static_assert(sizeof(std::atomic<uint32_t>) == sizeof(uint32_t), "the generation word is handed to futex(2)");
    void wake(int count) noexcept { syscall(SYS_futex, reinterpret_cast<uint32_t *>(&gen_), FUTEX_WAKE_PRIVATE, count, nullptr, nullptr, 0); }
    void sleep_while(uint32_t seen) noexcept
    {
        sleepers_.fetch_add(1);
        // returns at once (EAGAIN) when gen_ has moved on already; spurious returns are fine, the caller re-checks
        syscall(SYS_futex, reinterpret_cast<uint32_t *>(&gen_), FUTEX_WAIT_PRIVATE, seen, nullptr, nullptr, 0);
        sleepers_.fetch_sub(1);
    }
    void loop(uint32_t seen0)
    {
        uint32_t seen = seen0;
        for (;;) {
            // spin briefly for the next job (pause, not yield: a yield can cost milliseconds in sandboxed
            // runtimes), then sleep: CPU time is usually under a cgroup quota shared with the other slots
            auto t0 = std::chrono::steady_clock::now();
            while (gen_.load(std::memory_order_acquire) == seen) {
                if (std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(spin_us())) {
                    sleep_while(seen);
                    continue; // (a sleeper that was not chosen for one job sleeps on; it compares with the newest generation)
                }
                __builtin_ia32_pause();
            }
            seen = gen_.load(std::memory_order_acquire);
            if (stop_.load()) return;
            inside_.fetch_add(1);
            if (Job *j = cur_.load())
                if (j->joined.fetch_add(1, std::memory_order_relaxed) < j->want) work(*j);
            inside_.fetch_sub(1);
        }
    }
    std::vector<std::thread> th_;
    std::mutex job_mu_;
    std::atomic<uint32_t> gen_{0};
    std::atomic<int> inside_{0}, sleepers_{0};
    std::atomic<Job *> cur_{nullptr};
    std::atomic<bool> stop_{false};
    bool grow_failed_ = false;

public:
    int reserve(int nthreads) noexcept
    {
        std::lock_guard<std::mutex> job_lock(job_mu_);
        grow(nthreads - 1);
        return (int)th_.size() + 1;
    }
};

Pool *pool_create() { return new Pool(); }
int pool_reserve(Pool *p, int nthreads) { return p ? p->reserve(nthreads) : 1; }

int host_cpu_count()
{
    cpu_set_t set;
    CPU_ZERO(&set);
    if (sched_getaffinity(0, sizeof(set), &set) == 0) {
        const int n = CPU_COUNT(&set);
        if (n > 0) return n;
    }
    const long n = sysconf(_SC_NPROCESSORS_ONLN);
    return n > 0 ? (int)n : 1;
}
void pool_destroy(Pool *p) { delete p; }

void parallel_for(Pool *pool, int n, int nthreads, const std::function<void(int)> &fn)
{
    if (n <= 0) return;
    if (nthreads > n) nthreads = n;
    if (nthreads <= 1) {
        for (int i = 0; i < n; i++) fn(i);
        return;
    }
    (pool ? *pool : Pool::get()).run(n, nthreads, fn);
}

void os_randombytes(uint8_t *out, size_t len)
{
    while (len) {
        const ssize_t r = getrandom(out, len > 256 ? 256 : len, 0);
        if (r < 0) abort(); // randombytes.c:49-52
        out += r;
        len -= (size_t)r;
    }
}

} // namespace kosk
