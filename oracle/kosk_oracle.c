/*
 * kosk_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see kosk_oracle.h).
 *
 * Plain-C, single-thread restatement of the reference algorithm.  Every
 * function names the reference file:line it follows (paths relative to the
 * reference root).  Loop structure and the '%'-based field arithmetic are
 * kept (ss.cpp:23-32, gf3329.c:274-284) so that timing this code is a fair
 * "port" CPU baseline.  Dead computations of the reference's prove()
 * (ntt_r1, ntt_Ar polyvec, e_cpy, recon_t) are not restated:
 * they consume no randomness and feed no output.
 */
#include "kosk_oracle.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* ======================================================================== */
/* parameters                                                               */
/* ======================================================================== */

/* params.hpp:12-36, kyber/params.h:29-53, mlwe_prover.hpp:57-75 */
int ko_get_params(int K, ko_params *p)
{
    if (K < 2 || K > 4) return -1;
    memset(p, 0, sizeof *p);
    p->K = K;
    p->eta1 = (K == 2) ? 3 : 2;
    p->V = 2 * K;
    p->M = KO_NCHK + p->V + 1;
    p->E = 2 * p->eta1 + 1;
    p->Z = 2 * p->eta1;
    p->pk_bytes = (size_t)384 * K + 32;
    p->sk_bytes = (size_t)384 * K + p->pk_bytes + 64;
    const size_t T = KO_OPENED, R = KO_REST, M = p->M, Kk = K, E = p->E, Z = p->Z;
    const size_t sz[KO_NFIELDS] = {
        T * M * 2, T * M * 2, R * KO_NCHK * 2, R * KO_NCHK * 2, R * 32, T * 2,
        T * Kk * 2, T * Kk * 2, R * Kk * 2, T * Kk * 2, T * Kk * 2, T * Kk * 2, T * Kk * 2,
        R * Kk * 2, R * Kk * 2, R * Kk * E * 2, R * Kk * E * 2, T * Kk * E * 2, T * Kk * E * 2,
        T * Kk * Z * 2, T * Kk * Z * 2, R * Kk * Z * 2, R * Kk * Z * 2, R * 32};
    size_t o = 0;
    for (int i = 0; i < KO_NFIELDS; i++) {
        p->off[i] = o;
        p->size[i] = sz[i];
        o += sz[i];
    }
    p->proof_bytes = o;
    /* randombytes call sequence */
    p->tape_calls = 1 + p->M + 2 * p->M + 2 * K * p->E + 2 * K + K + 2 * K * p->Z;
    p->tape_bytes = 64 + (size_t)32 * p->M +
                    (size_t)302 * (2 * p->M + 2 * K * p->E + 2 * K + K + 2 * K * p->Z);
    p->tcomm_msg_bytes = (size_t)2 * 2 * (K + p->M);
    p->view_msg_bytes = 32 + (size_t)2 * ((6 + 4 * p->Z) * K + 2 * p->M);
    return 0;
}

/* ======================================================================== */
/* randomness tape (replaces kyber/randombytes.c:44-57)                     */
/* ======================================================================== */

void ko_tape_init(ko_tape *t, const uint8_t *buf, size_t len)
{
    t->buf = buf;
    t->len = len;
    t->pos = 0;
    t->calls = 0;
    t->overrun = 0;
}

static void tape_bytes(ko_tape *t, uint8_t *out, size_t n)
{
    t->calls++;
    if (t->pos + n > t->len) {
        t->overrun = 1;
        memset(out, 0, n);
        return;
    }
    memcpy(out, t->buf + t->pos, n);
    t->pos += n;
}

/* ======================================================================== */
/* Keccak / SHA-3 / SHAKE  (kyber/fips202.c)                                */
/* ======================================================================== */

static const uint64_t KECCAK_RC[24] = {
    0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL,
    0x000000000000808bULL, 0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL,
    0x000000000000008aULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000aULL,
    0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL, 0x8000000000008003ULL,
    0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
    0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
static const int KECCAK_ROT[24] = {1, 3, 6, 10, 15, 21, 28, 36, 45, 55, 2, 14,
                                   27, 41, 56, 8, 25, 43, 62, 18, 39, 61, 20, 44};
static const int KECCAK_PI[24] = {10, 7, 11, 17, 18, 3, 5, 16, 8, 21, 24, 4,
                                  15, 23, 19, 13, 12, 2, 20, 14, 22, 9, 6, 1};

static inline uint64_t rol64(uint64_t v, int n) { return (v << n) | (v >> (64 - n)); }

/* fips202.c:82-344 (KeccakF1600_StatePermute), rolled form of FIPS 202 3.2 */
void ko_keccak_f1600(uint64_t a[25])
{
    for (int rnd = 0; rnd < 24; rnd++) {
        uint64_t c[5], t;
        for (int x = 0; x < 5; x++) c[x] = a[x] ^ a[x + 5] ^ a[x + 10] ^ a[x + 15] ^ a[x + 20];
        for (int x = 0; x < 5; x++) {
            t = c[(x + 4) % 5] ^ rol64(c[(x + 1) % 5], 1);
            for (int y = 0; y < 25; y += 5) a[y + x] ^= t;
        }
        t = a[1];
        for (int i = 0; i < 24; i++) {
            int j = KECCAK_PI[i];
            uint64_t b = a[j];
            a[j] = rol64(t, KECCAK_ROT[i]);
            t = b;
        }
        for (int y = 0; y < 25; y += 5) {
            for (int x = 0; x < 5; x++) c[x] = a[y + x];
            for (int x = 0; x < 5; x++) a[y + x] = c[x] ^ (~c[(x + 1) % 5] & c[(x + 2) % 5]);
        }
        a[0] ^= KECCAK_RC[rnd];
    }
}

static inline void xor_bytes_le(uint64_t *st, size_t pos, const uint8_t *in, size_t n)
{
    for (size_t i = 0; i < n; i++) st[(pos + i) >> 3] ^= (uint64_t)in[i] << (8 * ((pos + i) & 7));
}

/* fips202.c:461-485 keccak_absorb_once + :426-446 keccak_squeeze */
static void sponge(uint8_t *out, size_t outlen, const uint8_t *in, size_t inlen, size_t rate, uint8_t dom)
{
    uint64_t st[25];
    memset(st, 0, sizeof st);
    while (inlen >= rate) {
        xor_bytes_le(st, 0, in, rate);
        ko_keccak_f1600(st);
        in += rate;
        inlen -= rate;
    }
    xor_bytes_le(st, 0, in, inlen);
    xor_bytes_le(st, inlen, &dom, 1);
    st[(rate - 1) >> 3] ^= 1ULL << 63;
    while (outlen > 0) {
        ko_keccak_f1600(st);
        size_t n = outlen < rate ? outlen : rate;
        for (size_t i = 0; i < n; i++) out[i] = (uint8_t)(st[i >> 3] >> (8 * (i & 7)));
        out += n;
        outlen -= n;
    }
}

void ko_sha3_256(uint8_t out[32], const uint8_t *in, size_t inlen) { sponge(out, 32, in, inlen, 136, 0x06); } /* fips202.c:745-754 */
void ko_sha3_512(uint8_t out[64], const uint8_t *in, size_t inlen) { sponge(out, 64, in, inlen, 72, 0x06); }  /* fips202.c:765-774 */
void ko_shake128(uint8_t *out, size_t outlen, const uint8_t *in, size_t inlen) { sponge(out, outlen, in, inlen, 168, 0x1F); }
void ko_shake256(uint8_t *out, size_t outlen, const uint8_t *in, size_t inlen) { sponge(out, outlen, in, inlen, 136, 0x1F); } /* fips202.c:723-734 */

/* symmetric-shake.c:43-51 */
void ko_shake256_prf(uint8_t *out, size_t outlen, const uint8_t key[32], uint8_t nonce)
{
    uint8_t ext[33];
    memcpy(ext, key, 32);
    ext[32] = nonce;
    ko_shake256(out, outlen, ext, 33);
}

/* ======================================================================== */
/* GF(3329)  (utils/gf3329.c:274-323)                                       */
/* ======================================================================== */

uint16_t ko_gf_add(uint16_t a, uint16_t b) { return a + b < KO_Q ? a + b : a + b - KO_Q; }
uint16_t ko_gf_sub(uint16_t a, uint16_t b) { return a < b ? a + KO_Q - b : a - b; }
uint16_t ko_gf_mul(uint16_t a, uint16_t b) { return (uint16_t)((uint32_t)a * b % KO_Q); }
uint16_t ko_gf_encode(int16_t a) { return a < 0 ? (uint16_t)(KO_Q + a) : (uint16_t)a; }
int16_t ko_gf_decode(uint16_t a) { return a > KO_Q / 2 ? (int16_t)(a - KO_Q) : (int16_t)a; }

static uint16_t gf_pow(uint16_t a, unsigned e)
{
    uint16_t r = 1;
    while (e) {
        if (e & 1) r = ko_gf_mul(r, a);
        a = ko_gf_mul(a, a);
        e >>= 1;
    }
    return r;
}

static uint16_t g_inv[KO_Q];
static int g_inv_ready;
static void inv_init(void)
{
    if (g_inv_ready) return;
    g_inv[0] = 0; /* gf3329.c:286-292: inverse of 0 reported as 0 */
    for (unsigned a = 1; a < KO_Q; a++) g_inv[a] = gf_pow((uint16_t)a, KO_Q - 2);
    g_inv_ready = 1;
}
uint16_t ko_gf_inv(uint16_t a)
{
    inv_init();
    return g_inv[a % KO_Q];
}

/* ======================================================================== */
/* Kyber ring arithmetic (kyber/reduce.c, ntt.c, poly.c, polyvec.c)         */
/* ======================================================================== */

#define QINV (-3327) /* q^-1 mod 2^16, kyber/reduce.h */

/* reduce.c:16-23 */
int16_t ko_montgomery_reduce(int32_t a)
{
    int16_t t = (int16_t)((int16_t)a * (int16_t)QINV);
    return (int16_t)((a - (int32_t)t * KO_Q) >> 16);
}

/* reduce.c:35-42 */
int16_t ko_barrett_reduce(int16_t a)
{
    const int16_t v = ((1 << 26) + KO_Q / 2) / KO_Q;
    int16_t t = (int16_t)(((int32_t)v * a + (1 << 25)) >> 26);
    return (int16_t)(a - t * KO_Q);
}

static int16_t g_zetas[128];
static int g_zetas_ready;
/* ntt.c:7-37 describes how the table of ntt.c:39-56 is derived: powers of
 * the 256-th root of unity 17 in bit-reversed order, Montgomery form,
 * centred.  Regenerated here from that recipe. */
static void zetas_init(void)
{
    if (g_zetas_ready) return;
    const int32_t mont = 2285; /* 2^16 mod q */
    int32_t pw[128];
    pw[0] = 1;
    for (int i = 1; i < 128; i++) pw[i] = pw[i - 1] * 17 % KO_Q;
    for (int i = 0; i < 128; i++) {
        int br = 0;
        for (int b = 0; b < 7; b++) br |= ((i >> b) & 1) << (6 - b);
        int32_t z = pw[br] * mont % KO_Q;
        if (z > KO_Q / 2) z -= KO_Q;
        g_zetas[i] = (int16_t)z;
    }
    g_zetas_ready = 1;
}
const int16_t *ko_zetas(void)
{
    zetas_init();
    return g_zetas;
}

static inline int16_t fqmul(int16_t a, int16_t b) { return ko_montgomery_reduce((int32_t)a * b); }

/* ntt.c:80-95 */
void ko_ntt(int16_t r[256])
{
    zetas_init();
    int k = 1;
    for (int len = 128; len >= 2; len >>= 1) {
        for (int start = 0; start < 256; start += 2 * len) {
            int16_t z = g_zetas[k++];
            for (int j = start; j < start + len; j++) {
                int16_t t = fqmul(z, r[j + len]);
                r[j + len] = (int16_t)(r[j] - t);
                r[j] = (int16_t)(r[j] + t);
            }
        }
    }
}

/* poly.c:323-328 */
void ko_poly_reduce(int16_t r[256])
{
    for (int i = 0; i < 256; i++) r[i] = ko_barrett_reduce(r[i]);
}

/* poly.c:261-265 */
void ko_poly_ntt(int16_t r[256])
{
    ko_ntt(r);
    ko_poly_reduce(r);
}

/* poly.c:307-313 */
void ko_poly_tomont(int16_t r[256])
{
    const int16_t f = (int16_t)((1ULL << 32) % KO_Q);
    for (int i = 0; i < 256; i++) r[i] = ko_montgomery_reduce((int32_t)r[i] * f);
}

/* ntt.c:139-146 */
static void basemul2(int16_t r[2], const int16_t a[2], const int16_t b[2], int16_t zeta)
{
    r[0] = fqmul(a[1], b[1]);
    r[0] = fqmul(r[0], zeta);
    r[0] = (int16_t)(r[0] + fqmul(a[0], b[0]));
    r[1] = fqmul(a[0], b[1]);
    r[1] = (int16_t)(r[1] + fqmul(a[1], b[0]));
}

/* poly.c:290-297 */
static void poly_basemul(int16_t r[256], const int16_t a[256], const int16_t b[256])
{
    zetas_init();
    for (int i = 0; i < 64; i++) {
        basemul2(r + 4 * i, a + 4 * i, b + 4 * i, g_zetas[64 + i]);
        basemul2(r + 4 * i + 2, a + 4 * i + 2, b + 4 * i + 2, (int16_t)-g_zetas[64 + i]);
    }
}

/* polyvec.c:202-214; a, b are K consecutive polys */
void ko_polyvec_basemul_acc(int16_t r[256], const int16_t *a, const int16_t *b, int K)
{
    int16_t t[256];
    poly_basemul(r, a, b);
    for (int i = 1; i < K; i++) {
        poly_basemul(t, a + 256 * i, b + 256 * i);
        for (int j = 0; j < 256; j++) r[j] = (int16_t)(r[j] + t[j]);
    }
    ko_poly_reduce(r);
}

/* poly.c:124-139 */
void ko_poly_tobytes(uint8_t r[384], const int16_t a[256])
{
    for (int i = 0; i < 128; i++) {
        uint16_t t0 = (uint16_t)a[2 * i], t1 = (uint16_t)a[2 * i + 1];
        t0 = (uint16_t)(t0 + (((int16_t)t0 >> 15) & KO_Q));
        t1 = (uint16_t)(t1 + (((int16_t)t1 >> 15) & KO_Q));
        r[3 * i + 0] = (uint8_t)t0;
        r[3 * i + 1] = (uint8_t)((t0 >> 8) | (t1 << 4));
        r[3 * i + 2] = (uint8_t)(t1 >> 4);
    }
}

/* poly.c:151-158 */
void ko_poly_frombytes(int16_t r[256], const uint8_t a[384])
{
    for (int i = 0; i < 128; i++) {
        r[2 * i] = (int16_t)(((a[3 * i] >> 0) | ((uint16_t)a[3 * i + 1] << 8)) & 0xFFF);
        r[2 * i + 1] = (int16_t)(((a[3 * i + 1] >> 4) | ((uint16_t)a[3 * i + 2] << 4)) & 0xFFF);
    }
}

/* indcpa.c:124-145 (rej_uniform) + :168-193 (gen_matrix); the XOF stream is
 * SHAKE128(seed || x || y) (symmetric-shake.c:18-30).  Squeezing a long
 * prefix at once is the same byte stream the reference squeezes blockwise. */
void ko_gen_matrix(int16_t *A, const uint8_t seed[32], int transposed, int K)
{
    enum { XOFLEN = 168 * 8 };
    uint8_t ext[34], buf[XOFLEN];
    memcpy(ext, seed, 32);
    for (int i = 0; i < K; i++) {
        for (int j = 0; j < K; j++) {
            ext[32] = (uint8_t)(transposed ? i : j);
            ext[33] = (uint8_t)(transposed ? j : i);
            ko_shake128(buf, XOFLEN, ext, 34);
            int16_t *r = A + ((size_t)i * K + j) * 256;
            int ctr = 0;
            for (size_t pos = 0; ctr < 256 && pos + 3 <= XOFLEN; pos += 3) {
                uint16_t v0 = ((buf[pos] >> 0) | ((uint16_t)buf[pos + 1] << 8)) & 0xFFF;
                uint16_t v1 = ((buf[pos + 1] >> 4) | ((uint16_t)buf[pos + 2] << 4)) & 0xFFF;
                if (v0 < KO_Q) r[ctr++] = (int16_t)v0;
                if (ctr < 256 && v1 < KO_Q) r[ctr++] = (int16_t)v1;
            }
            if (ctr < 256) { /* probability < 2^-200 for 1344 bytes */
                fprintf(stderr, "ko_gen_matrix: XOF prefix exhausted\n");
                abort();
            }
        }
    }
}

/* cbd.c:58-107 */
void ko_cbd(int16_t r[256], const uint8_t *buf, int eta)
{
    if (eta == 2) {
        for (int i = 0; i < 32; i++) {
            uint32_t t = (uint32_t)buf[4 * i] | ((uint32_t)buf[4 * i + 1] << 8) |
                         ((uint32_t)buf[4 * i + 2] << 16) | ((uint32_t)buf[4 * i + 3] << 24);
            uint32_t d = (t & 0x55555555u) + ((t >> 1) & 0x55555555u);
            for (int j = 0; j < 8; j++) {
                int16_t a = (d >> (4 * j)) & 3, b = (d >> (4 * j + 2)) & 3;
                r[8 * i + j] = (int16_t)(a - b);
            }
        }
    } else {
        for (int i = 0; i < 64; i++) {
            uint32_t t = (uint32_t)buf[3 * i] | ((uint32_t)buf[3 * i + 1] << 8) | ((uint32_t)buf[3 * i + 2] << 16);
            uint32_t d = (t & 0x00249249u) + ((t >> 1) & 0x00249249u) + ((t >> 2) & 0x00249249u);
            for (int j = 0; j < 4; j++) {
                int16_t a = (d >> (6 * j)) & 7, b = (d >> (6 * j + 3)) & 7;
                r[4 * i + j] = (int16_t)(a - b);
            }
        }
    }
}

/* ======================================================================== */
/* Lagrange tables (utils/precomputed_kyber.h:10-13; the .c is not mounted, */
/* semantics from the call sites ss.cpp:26-27, :47, :66 -- SURVEY.md A8)    */
/* ======================================================================== */

#define NSHARE_ROWS (KO_PARTIES - KO_OPENED - 1) /* 1303 */
static uint16_t *g_tab_share; /* [1303][407]  L_j(407+x), nodes 0..406     */
static uint16_t *g_tab_rec_d; /* [256][407]   L_j(i),     nodes 256..662   */
static uint16_t *g_tab_rec_2d; /* [256][813]  L_j(i),     nodes 256..1068  */

/* row[j] = L_j(t) for the n consecutive integer nodes a..a+n-1 (t no node) */
static void lagrange_row(uint16_t *row, int n, int a, int t, const uint16_t *fact)
{
    uint16_t *pre = (uint16_t *)malloc(sizeof(uint16_t) * (size_t)(n + 1));
    uint16_t *suf = (uint16_t *)malloc(sizeof(uint16_t) * (size_t)(n + 1));
    pre[0] = 1;
    for (int m = 0; m < n; m++) {
        int d = ((t - a - m) % KO_Q + KO_Q) % KO_Q;
        pre[m + 1] = ko_gf_mul(pre[m], (uint16_t)d);
    }
    suf[n] = 1;
    for (int m = n - 1; m >= 0; m--) {
        int d = ((t - a - m) % KO_Q + KO_Q) % KO_Q;
        suf[m] = ko_gf_mul(suf[m + 1], (uint16_t)d);
    }
    for (int j = 0; j < n; j++) {
        uint16_t num = ko_gf_mul(pre[j], suf[j + 1]);
        uint16_t den = ko_gf_mul(fact[j], fact[n - 1 - j]);
        if ((n - 1 - j) & 1) den = ko_gf_sub(0, den);
        row[j] = ko_gf_mul(num, ko_gf_inv(den));
    }
    free(pre);
    free(suf);
}

static void tables_init(void)
{
    if (g_tab_share) return;
    inv_init();
    uint16_t fact[KO_DEG2 + 2];
    fact[0] = 1;
    for (int i = 1; i <= KO_DEG2 + 1; i++) fact[i] = ko_gf_mul(fact[i - 1], (uint16_t)i);
    uint16_t *ts = (uint16_t *)malloc(sizeof(uint16_t) * NSHARE_ROWS * (KO_DEG + 1));
    uint16_t *td = (uint16_t *)malloc(sizeof(uint16_t) * 256 * (KO_DEG + 1));
    uint16_t *t2 = (uint16_t *)malloc(sizeof(uint16_t) * 256 * (KO_DEG2 + 1));
    for (int x = 0; x < NSHARE_ROWS; x++) lagrange_row(ts + (size_t)x * (KO_DEG + 1), KO_DEG + 1, 0, KO_DEG + 1 + x, fact);
    for (int i = 0; i < 256; i++) lagrange_row(td + (size_t)i * (KO_DEG + 1), KO_DEG + 1, 256, i, fact);
    for (int i = 0; i < 256; i++) lagrange_row(t2 + (size_t)i * (KO_DEG2 + 1), KO_DEG2 + 1, 256, i, fact);
    g_tab_rec_d = td;
    g_tab_rec_2d = t2;
    g_tab_share = ts;
}

const uint16_t *ko_table_share_ddeg(void) { tables_init(); return g_tab_share; }
const uint16_t *ko_table_recon_ddeg(void) { tables_init(); return g_tab_rec_d; }
const uint16_t *ko_table_recon_2ddeg(void) { tables_init(); return g_tab_rec_2d; }

/* ======================================================================== */
/* packed Shamir sharing (ss.cpp)                                           */
/* ======================================================================== */

/* ss.cpp:23-32 == :88-97: parties 151..1453 from the 407 values at 0..406 */
static void expand_rest(uint16_t share_y[KO_PARTIES], const uint16_t sec[KO_DEG + 1])
{
    const uint16_t *tab = g_tab_share;
    for (int x = 0; x < NSHARE_ROWS; x++) {
        const uint16_t *row = tab + (size_t)x * (KO_DEG + 1);
        uint16_t acc = 0;
        for (int j = 0; j < KO_DEG + 1; j++) acc = ko_gf_add(acc, ko_gf_mul(sec[j], row[j]));
        share_y[KO_OPENED + 1 + x] = acc;
    }
}

/* ss.cpp:3-34 */
void ko_share_secrets_ddeg(uint16_t share_y[KO_PARTIES], const uint16_t secret[256], ko_tape *t)
{
    tables_init();
    uint8_t rb[(KO_OPENED + 1) * 2];
    uint16_t sec[KO_DEG + 1];
    tape_bytes(t, rb, sizeof rb);
    memcpy(sec, secret, 256 * sizeof(uint16_t));
    for (int i = 0; i < KO_OPENED + 1; i++) {
        uint16_t v = (uint16_t)(((rb[2 * i] << 8) | rb[2 * i + 1]) % KO_Q);
        share_y[i] = v;
        sec[256 + i] = v;
    }
    expand_rest(share_y, sec);
}

/* ss.cpp:76-99 */
void ko_recompute_share_secrets_ddeg(uint16_t share_y[KO_PARTIES], const uint16_t y[KO_DEG + 1])
{
    tables_init();
    for (int i = 0; i < KO_OPENED + 1; i++) share_y[i] = y[256 + i];
    expand_rest(share_y, y);
}

/* ss.cpp:37-54 */
void ko_recon_secrets_ddeg(uint16_t secret[256], const uint16_t share_y[KO_PARTIES])
{
    tables_init();
    for (int i = 0; i < 256; i++) {
        const uint16_t *row = g_tab_rec_d + (size_t)i * (KO_DEG + 1);
        uint16_t acc = 0;
        for (int j = 0; j < KO_DEG + 1; j++) acc = ko_gf_add(acc, ko_gf_mul(share_y[j], row[j]));
        secret[i] = acc;
    }
}

/* ss.cpp:56-73 */
void ko_recon_secrets_2ddeg(uint16_t secret[256], const uint16_t share_y[KO_PARTIES])
{
    tables_init();
    for (int i = 0; i < 256; i++) {
        const uint16_t *row = g_tab_rec_2d + (size_t)i * (KO_DEG2 + 1);
        uint16_t acc = 0;
        for (int j = 0; j < KO_DEG2 + 1; j++) acc = ko_gf_add(acc, ko_gf_mul(share_y[j], row[j]));
        secret[i] = acc;
    }
}

/* ---- interpolation through arbitrary nodes ------------------------------
 * Stands in for NTL 11.5.1 interpolate()/eval() (ZZ_pX.h:1053, :1040-1043;
 * call sites mlwe_verifier.cpp:201-219, :337-350, :410-440, :523-530).  The
 * interpolant over GF(3329) is unique, so any exact method gives the same
 * canonical values.  Barycentric form: p(k) = l(k) * sum_j w_j y_j /(k-x_j). */
typedef struct {
    int n;
    uint16_t *x;
    uint16_t *w;
} interp_nodes;

static void interp_nodes_init(interp_nodes *c, const uint16_t *x, int n)
{
    inv_init();
    c->n = n;
    c->x = (uint16_t *)malloc(sizeof(uint16_t) * (size_t)n);
    c->w = (uint16_t *)malloc(sizeof(uint16_t) * (size_t)n);
    for (int j = 0; j < n; j++) c->x[j] = (uint16_t)(x[j] % KO_Q);
    for (int j = 0; j < n; j++) {
        uint16_t d = 1;
        for (int m = 0; m < n; m++)
            if (m != j) d = ko_gf_mul(d, ko_gf_sub(c->x[j], c->x[m]));
        c->w[j] = g_inv[d];
    }
}
static void interp_nodes_free(interp_nodes *c)
{
    free(c->x);
    free(c->w);
}
static void interp_eval_range(const interp_nodes *c, const uint16_t *y, uint16_t *out, int k0, int k1)
{
    for (int k = k0; k < k1; k++) {
        uint16_t l = 1, acc = 0;
        int hit = -1;
        for (int j = 0; j < c->n; j++) {
            uint16_t d = ko_gf_sub((uint16_t)k, c->x[j]);
            if (d == 0) { hit = j; break; }
            l = ko_gf_mul(l, d);
            acc = ko_gf_add(acc, ko_gf_mul(ko_gf_mul(c->w[j], (uint16_t)(y[j] % KO_Q)), g_inv[d]));
        }
        out[k - k0] = hit >= 0 ? (uint16_t)(y[hit] % KO_Q) : ko_gf_mul(l, acc);
    }
}
void ko_interp_eval(uint16_t *out, int neval, const uint16_t *x, const uint16_t *y, int n)
{
    interp_nodes c;
    interp_nodes_init(&c, x, n);
    interp_eval_range(&c, y, out, 0, neval);
    interp_nodes_free(&c);
}

/* ======================================================================== */
/* preprocessing (mlwe_prover.cpp:4-59)                                     */
/* ======================================================================== */

#define MAXM 79
#define MAXE 7
#define MAXZ 6
struct ko_pre {
    uint16_t f[MAXM][256];
    uint16_t ntt_f[MAXM][256];
    uint16_t f_sh[MAXM][KO_PARTIES];
    uint16_t ntt_f_sh[MAXM][KO_PARTIES];
    uint16_t s_eta_sh[4][MAXE][KO_PARTIES];
    uint16_t e_eta_sh[4][MAXE][KO_PARTIES];
};
ko_pre *ko_pre_alloc(void) { return (ko_pre *)calloc(1, sizeof(ko_pre)); }
void ko_pre_free(ko_pre *p) { free(p); }
const uint16_t *ko_pre_f(const ko_pre *p, int i) { return p->f[i]; }
const uint16_t *ko_pre_ntt_f(const ko_pre *p, int i) { return p->ntt_f[i]; }
const uint16_t *ko_pre_f_shares(const ko_pre *p, int i) { return p->f_sh[i]; }
const uint16_t *ko_pre_ntt_f_shares(const ko_pre *p, int i) { return p->ntt_f_sh[i]; }
const uint16_t *ko_pre_s_eta_shares(const ko_pre *p, int i, int j) { return p->s_eta_sh[i][j]; }
const uint16_t *ko_pre_e_eta_shares(const ko_pre *p, int i, int j) { return p->e_eta_sh[i][j]; }

/* mlwe_prover.cpp:4-39 */
void ko_prepare_randomness(int K, ko_tape *t, ko_pre *pre)
{
    ko_params P;
    ko_get_params(K, &P);
    uint8_t seed[32], prand[512];
    for (int i = 0; i < P.M; i++) { /* :8-14 */
        tape_bytes(t, seed, 32);
        ko_shake256_prf(prand, 512, seed, (uint8_t)i);
        for (int j = 0; j < 256; j++) pre->f[i][j] = (uint16_t)(((prand[2 * j] << 8) | prand[2 * j + 1]) % KO_Q);
    }
    for (int i = 0; i < P.M; i++) { /* :17-26 */
        int16_t poly[256];
        for (int j = 0; j < 256; j++) poly[j] = ko_gf_decode(pre->f[i][j]);
        ko_poly_ntt(poly);
        for (int j = 0; j < 256; j++) pre->ntt_f[i][j] = ko_gf_encode(poly[j]);
    }
    for (int i = 0; i < P.M; i++) { /* :29-38, tape order f_i then NTT(f_i) */
        ko_share_secrets_ddeg(pre->f_sh[i], pre->f[i], t);
        ko_share_secrets_ddeg(pre->ntt_f_sh[i], pre->ntt_f[i], t);
    }
}

/* mlwe_prover.cpp:41-59 */
void ko_prepare_range_proof(int K, ko_tape *t, ko_pre *pre)
{
    ko_params P;
    ko_get_params(K, &P);
    uint16_t cst[MAXE][256];
    for (int c = -P.eta1; c <= P.eta1; c++) {
        uint16_t e = ko_gf_encode((int16_t)c);
        for (int j = 0; j < 256; j++) cst[c + P.eta1][j] = e;
    }
    for (int i = 0; i < K; i++)
        for (int j = 0; j < P.E; j++) { /* tape order: s then e */
            ko_share_secrets_ddeg(pre->s_eta_sh[i][j], cst[j], t);
            ko_share_secrets_ddeg(pre->e_eta_sh[i][j], cst[j], t);
        }
}

/* ======================================================================== */
/* keygen (kosk.cpp:4-70)                                                   */
/* ======================================================================== */

void ko_keygen(int K, ko_tape *t, uint8_t *pk, uint8_t *sk, ko_mlwe *raw)
{
    uint8_t buf[64];
    tape_bytes(t, buf, 64);
    buf[32] = (uint8_t)K;
    {
        uint8_t g[64];
        ko_sha3_512(g, buf, 33); /* kosk.cpp:12-14 */
        memcpy(buf, g, 64);
    }
    ko_keygen_from_seeds(K, buf, buf + 32, pk, sk, raw);
}

/* kosk.cpp:16-69: everything after hash_g, from the two seeds it yields.  (The reference hands both back: the public
 * seed is pk's last 32 bytes, kosk.cpp:58, and -- its quirk -- the noise seed is sk's z, :67-69; tests/test_oracle_vs_ref.py
 * runs the reference's own kyber_keygen on OS randomness and feeds those two seeds in here.) */
void ko_keygen_from_seeds(int K, const uint8_t public_seed[32], const uint8_t noise_seed[32], uint8_t *pk, uint8_t *sk, ko_mlwe *raw)
{
    ko_params P;
    ko_get_params(K, &P);
    int16_t A[4 * 4 * 256];
    ko_gen_matrix(A, public_seed, 0, K);
    uint8_t nb[3 * 256 / 4];
    int16_t shat[4 * 256], ehat[4 * 256], tt[4 * 256];
    uint8_t nonce = 0;
    for (int i = 0; i < K; i++) { /* kosk.cpp:17-18 */
        ko_shake256_prf(nb, (size_t)P.eta1 * 64, noise_seed, nonce++);
        ko_cbd(shat + 256 * i, nb, P.eta1);
    }
    for (int i = 0; i < K; i++) { /* kosk.cpp:19-20 */
        ko_shake256_prf(nb, (size_t)P.eta1 * 64, noise_seed, nonce++);
        ko_cbd(ehat + 256 * i, nb, P.eta1);
    }
    memset(raw, 0, sizeof *raw);
    for (int i = 0; i < K; i++)
        for (int j = 0; j < K; j++) memcpy(raw->A[i][j], A + ((size_t)i * K + j) * 256, 512);
    for (int i = 0; i < K; i++) {
        memcpy(raw->s[i], shat + 256 * i, 512);
        memcpy(raw->e[i], ehat + 256 * i, 512);
    }
    for (int i = 0; i < K; i++) { /* kosk.cpp:39-40 */
        ko_poly_ntt(shat + 256 * i);
        ko_poly_ntt(ehat + 256 * i);
    }
    for (int i = 0; i < K; i++) { /* kosk.cpp:42-48 */
        ko_polyvec_basemul_acc(tt + 256 * i, A + (size_t)i * K * 256, shat, K);
        ko_poly_tomont(tt + 256 * i);
        for (int j = 0; j < 256; j++) tt[256 * i + j] = (int16_t)(tt[256 * i + j] + ehat[256 * i + j]);
        ko_poly_reduce(tt + 256 * i);
        memcpy(raw->t[i], tt + 256 * i, 512);
    }
    for (int i = 0; i < K; i++) ko_poly_tobytes(pk + 384 * i, tt + 256 * i); /* :57-58 */
    memcpy(pk + 384 * K, public_seed, 32);
    for (int i = 0; i < K; i++) ko_poly_tobytes(sk + 384 * i, shat + 256 * i); /* :62-69 */
    memcpy(sk + 384 * K, pk, P.pk_bytes);
    ko_sha3_256(sk + P.sk_bytes - 64, pk, P.pk_bytes);
    memcpy(sk + P.sk_bytes - 32, noise_seed, 32); /* quirk: z == noise seed, kosk.cpp:67-69 */
}

/* ======================================================================== */
/* shared helpers of prove / verify                                         */
/* ======================================================================== */

static inline void put16(uint8_t *p, uint16_t v)
{
    p[0] = (uint8_t)v;
    p[1] = (uint8_t)(v >> 8);
}
static inline uint16_t get16(const uint8_t *p) { return (uint16_t)(p[0] | (p[1] << 8)); }

/* mlwe_prover.cpp:130-153 == mlwe_verifier.cpp:40-65 */
static void derive_alpha(const ko_params *P, const uint8_t *tcomm_all, uint8_t h1[32],
                         uint16_t *alpha, uint16_t pw[][MAXM])
{
    uint8_t a_[2 * (KO_NCHK + 8)];
    const int na = KO_NCHK + P->V;
    ko_sha3_256(h1, tcomm_all, (size_t)KO_PARTIES * 32);
    ko_shake256_prf(a_, (size_t)2 * na, h1, 1);
    for (int i = 0; i < na; i++) alpha[i] = (uint16_t)(((a_[2 * i] << 8) | a_[2 * i + 1]) % KO_Q);
    for (int i = 0; i < na; i++) {
        pw[i][0] = 1;
        pw[i][1] = alpha[i];
        for (int k = 2; k < P->M; k++) pw[i][k] = ko_gf_mul(pw[i][k - 1], alpha[i]);
    }
}

/* mlwe_prover.cpp:445-474 == mlwe_verifier.cpp:633-676 */
static void derive_opened(const uint8_t *digests_all, uint8_t ch[32], uint16_t I[KO_OPENED])
{
    uint8_t I_[2 * KO_OPENED];
    ko_sha3_256(ch, digests_all, (size_t)KO_PARTIES * 32);
    ko_shake256_prf(I_, sizeof I_, ch, 1);
    for (int i = 0; i < KO_OPENED; i++) I[i] = (uint16_t)(((I_[2 * i] << 8) | I_[2 * i + 1]) % KO_PARTIES);
    for (int i = 1; i < KO_OPENED; i++) {
        uint16_t inc = 0;
        int dup;
        do {
            dup = 0;
            for (int j = 0; j < i; j++)
                if ((I[i] + inc) % KO_PARTIES == I[j]) {
                    dup = 1;
                    inc++;
                    break;
                }
        } while (dup);
        I[i] = (uint16_t)((I[i] + inc) % KO_PARTIES);
    }
}

/* lin[j] = base + sum_{k=1}^{M-1} pw[j][k] * v[k]   (prover :159-203) */
static inline uint16_t lincomb(const uint16_t *pwj, const uint16_t *v, uint16_t base, int M)
{
    uint16_t acc = base;
    for (int k = 1; k < M; k++) acc = ko_gf_add(acc, ko_gf_mul(pwj[k], v[k]));
    return acc;
}

/* ======================================================================== */
/* prove (mlwe_prover.cpp:81-538)                                           */
/* ======================================================================== */

typedef uint16_t sharerow[KO_PARTIES];

/* ---- test hook: a CRAFTING prover ---------------------------------------
 * No counterpart in the reference.  The verifier's behaviour on u16 values
 * >= q (gf3329_add / gf3329_sub do not reduce, gf3329.c:274-280) can only be
 * exercised end to end by proofs whose Fiat-Shamir hashes were computed OVER
 * such values -- substituting them into a finished proof changes a hash and
 * is rejected for that reason alone.  ko_craft_add() queues "add mult * q to
 * share `idx` of party `party`" (kind 0: s_sh, 1: e_sh, 2: f_sh, 3:
 * ntt_f_sh); ko_prove applies the queue right after the witness sharing and
 * otherwise runs the reference's steps unchanged on what it finds.        */
typedef struct { int kind, idx, party, mult; } craft_item;
static craft_item g_craft[512];
static int g_ncraft;
void ko_craft_clear(void) { g_ncraft = 0; }
int ko_craft_add(int kind, int idx, int party, int mult)
{
    if (g_ncraft >= (int)(sizeof g_craft / sizeof g_craft[0]) || kind < 0 || kind > 3 || party < 0 || party >= KO_PARTIES || idx < 0) return -1;
    g_craft[g_ncraft++] = (craft_item){kind, idx, party, mult};
    return 0;
}

void ko_prove(int K, ko_tape *tp, uint8_t *pi, const ko_mlwe *mlwe, const ko_pre *pre, ko_trace *trace)
{
    ko_params P;
    ko_get_params(K, &P);
    tables_init();
    const int M = P.M, E = P.E, Z = P.Z, N = KO_PARTIES;

    /* heap workspace (the reference keeps ~2 MB of this on the stack) */
    sharerow *s_sh = calloc(4, sizeof(sharerow)), *e_sh = calloc(4, sizeof(sharerow));
    sharerow *sr_sh = calloc(4, sizeof(sharerow)), *er_sh = calloc(4, sizeof(sharerow));
    sharerow *r_sh = calloc(8, sizeof(sharerow)), *nttr_sh = calloc(8, sizeof(sharerow));
    sharerow *beta = calloc(KO_NCHK, sizeof(sharerow)), *gamma = calloc(KO_NCHK, sizeof(sharerow));
    sharerow *ntt_sr_sh = calloc(4, sizeof(sharerow)), *ntt_er_sh = calloc(4, sizeof(sharerow));
    sharerow *ntt_s_sh = calloc(4, sizeof(sharerow)), *ntt_e_sh = calloc(4, sizeof(sharerow));
    sharerow *ntt_Asr_sh = calloc(4, sizeof(sharerow)), *ntt_As_sh = calloc(4, sizeof(sharerow));
    sharerow *ntt_Ar_sh = calloc(4, sizeof(sharerow)), *ntt_t_sh = calloc(4, sizeof(sharerow));
    sharerow *s_sub = calloc(4 * MAXE, sizeof(sharerow)), *e_sub = calloc(4 * MAXE, sizeof(sharerow));
    sharerow *zs_d = calloc(4 * MAXZ, sizeof(sharerow)), *ze_d = calloc(4 * MAXZ, sizeof(sharerow));
    sharerow *zs_2d = calloc(4 * MAXZ, sizeof(sharerow)), *ze_2d = calloc(4 * MAXZ, sizeof(sharerow));
    sharerow *us = calloc(4 * MAXZ, sizeof(sharerow)), *ue = calloc(4 * MAXZ, sizeof(sharerow));
    uint8_t(*tcomm)[32] = calloc(N, 32), (*vdig)[32] = calloc(N, 32);

    /* P1 :89-101 share the witness, tape order s_i then e_i */
    uint16_t sec[256];
    for (int i = 0; i < K; i++) {
        for (int j = 0; j < 256; j++) sec[j] = ko_gf_encode(mlwe->s[i][j]);
        ko_share_secrets_ddeg(s_sh[i], sec, tp);
        for (int j = 0; j < 256; j++) sec[j] = ko_gf_encode(mlwe->e[i][j]);
        ko_share_secrets_ddeg(e_sh[i], sec, tp);
    }

    for (int c = 0; c < g_ncraft; c++) { /* test hook, see ko_craft_add */
        const craft_item *ci = &g_craft[c];
        ko_pre *wpre = (ko_pre *)pre;
        uint16_t *v = NULL;
        if (ci->kind == 0 && ci->idx < K) v = &s_sh[ci->idx][ci->party];
        else if (ci->kind == 1 && ci->idx < K) v = &e_sh[ci->idx][ci->party];
        else if (ci->kind == 2 && ci->idx < M) v = &wpre->f_sh[ci->idx][ci->party];
        else if (ci->kind == 3 && ci->idx < M) v = &wpre->ntt_f_sh[ci->idx][ci->party];
        if (v && (long)*v + (long)ci->mult * KO_Q <= 65535 && ci->mult >= 0) *v = (uint16_t)(*v + ci->mult * KO_Q);
    }

    /* P2+P3 :103-127 per-party commitment of (s, e, f, NTT f) shares */
    uint8_t msg[1024];
    for (int p = 0; p < N; p++) {
        uint8_t *m = msg;
        for (int j = 0; j < K; j++, m += 2) put16(m, s_sh[j][p]);
        for (int j = 0; j < K; j++, m += 2) put16(m, e_sh[j][p]);
        for (int j = 0; j < M; j++, m += 2) put16(m, pre->f_sh[j][p]);
        for (int j = 0; j < M; j++, m += 2) put16(m, pre->ntt_f_sh[j][p]);
        ko_sha3_256(tcomm[p], msg, (size_t)(m - msg));
    }

    /* P4 :130-153 */
    uint8_t h1[32];
    uint16_t alpha[KO_NCHK + 8];
    uint16_t pw[KO_NCHK + 8][MAXM];
    derive_alpha(&P, &tcomm[0][0], h1, alpha, pw);

    /* P5+P6 :159-214.  For r / NTT r the k == 0 term is f_sh[71], :187,:196 */
    uint16_t fv[MAXM], tv[MAXM];
    for (int p = 0; p < N; p++) {
        for (int k = 0; k < M; k++) {
            fv[k] = pre->f_sh[k][p];
            tv[k] = pre->ntt_f_sh[k][p];
        }
        for (int j = 0; j < KO_NCHK; j++) {
            beta[j][p] = lincomb(pw[j], fv, fv[0], M);
            gamma[j][p] = lincomb(pw[j], tv, tv[0], M);
        }
        for (int j = 0; j < P.V; j++) {
            r_sh[j][p] = lincomb(pw[KO_NCHK + j], fv, fv[KO_NCHK + 1], M);
            nttr_sh[j][p] = lincomb(pw[KO_NCHK + j], tv, tv[KO_NCHK + 1], M);
        }
    }

    /* P7 :222-245 open s+r, e+r */
    uint16_t sr_rec[4][256], er_rec[4][256];
    uint16_t sr_rnd[4][KO_DEG + 1], er_rnd[4][KO_DEG + 1];
    for (int i = 0; i < K; i++) {
        for (int p = 0; p < N; p++) {
            sr_sh[i][p] = ko_gf_add(s_sh[i][p], r_sh[i][p]);
            er_sh[i][p] = ko_gf_add(e_sh[i][p], r_sh[i + K][p]);
        }
        ko_recon_secrets_ddeg(sr_rec[i], sr_sh[i]);
        ko_recon_secrets_ddeg(er_rec[i], er_sh[i]);
        for (int j = 256; j < KO_DEG + 1; j++) {
            sr_rnd[i][j] = sr_sh[i][j - 256];
            er_rnd[i][j] = er_sh[i][j - 256];
        }
    }

    /* P8 (live part) :252,:256  NTT(s) ; P9 :260-277 NTT of opened values */
    int16_t shat[4 * 256], srhat[4 * 256], erhat[256];
    for (int i = 0; i < K; i++) {
        memcpy(shat + 256 * i, mlwe->s[i], 512);
        ko_poly_ntt(shat + 256 * i);
        for (int j = 0; j < 256; j++) srhat[256 * i + j] = ko_gf_decode(sr_rec[i][j]);
        ko_poly_ntt(srhat + 256 * i);
        for (int j = 0; j < 256; j++) erhat[j] = ko_gf_decode(er_rec[i][j]);
        ko_poly_ntt(erhat);
        for (int j = 0; j < 256; j++) {
            sr_rnd[i][j] = ko_gf_encode(srhat[256 * i + j]);
            er_rnd[i][j] = ko_gf_encode(erhat[j]);
        }
    }

    /* P10 :279-289 (live rows) */
    int16_t As[4][256], Asr[4][256];
    for (int i = 0; i < K; i++) {
        ko_polyvec_basemul_acc(As[i], &mlwe->A[i][0][0], shat, K);
        ko_poly_tomont(As[i]);
        ko_polyvec_basemul_acc(Asr[i], &mlwe->A[i][0][0], srhat, K);
        ko_poly_tomont(Asr[i]);
    }
    /* mlwe->A is [4][4][256]: rows of a K<4 matrix are not contiguous per
     * row-vector beyond K entries, but A[i][0..K) is contiguous, as needed. */

    /* P11 :292-318 */
    for (int i = 0; i < K; i++) {
        ko_recompute_share_secrets_ddeg(ntt_sr_sh[i], sr_rnd[i]);
        ko_recompute_share_secrets_ddeg(ntt_er_sh[i], er_rnd[i]);
        for (int p = 0; p < N; p++) {
            ntt_s_sh[i][p] = ko_gf_sub(ntt_sr_sh[i][p], nttr_sh[i][p]);
            ntt_e_sh[i][p] = ko_gf_sub(ntt_er_sh[i][p], nttr_sh[i + K][p]);
        }
    }
    for (int i = 0; i < K; i++) {
        uint16_t asr_rnd[KO_DEG + 1], as_sec[256];
        for (int j = 0; j < 256; j++) {
            as_sec[j] = ko_gf_encode(As[i][j]);
            asr_rnd[j] = ko_gf_encode(Asr[i][j]);
        }
        for (int j = 256; j < KO_DEG + 1; j++) asr_rnd[j] = sr_rnd[i][j];
        ko_recompute_share_secrets_ddeg(ntt_Asr_sh[i], asr_rnd);
        ko_share_secrets_ddeg(ntt_As_sh[i], as_sec, tp); /* tape, :316 */
        for (int p = 0; p < N; p++) ntt_Ar_sh[i][p] = ko_gf_sub(ntt_Asr_sh[i][p], ntt_As_sh[i][p]);
    }

    /* P12 :321-323 */
    for (int i = 0; i < K; i++)
        for (int p = 0; p < N; p++) ntt_t_sh[i][p] = ko_gf_add(ntt_As_sh[i][p], ntt_e_sh[i][p]);

    /* P13 :338-344 */
    for (int i = 0; i < K; i++)
        for (int j = 0; j < E; j++)
            for (int p = 0; p < N; p++) {
                s_sub[i * MAXE + j][p] = ko_gf_sub(s_sh[i][p], pre->s_eta_sh[i][j][p]);
                e_sub[i * MAXE + j][p] = ko_gf_sub(e_sh[i][p], pre->e_eta_sh[i][j][p]);
            }

    /* P14 :351-373 multiplication-gate chain, tape order s then e */
    for (int i = 0; i < K; i++)
        for (int j = 0; j < Z; j++) {
            const uint16_t *sa = j == 0 ? s_sub[i * MAXE] : zs_d[i * MAXZ + j - 1];
            const uint16_t *ea = j == 0 ? e_sub[i * MAXE] : ze_d[i * MAXZ + j - 1];
            const uint16_t *sb = s_sub[i * MAXE + j + 1], *eb = e_sub[i * MAXE + j + 1];
            for (int p = 0; p < N; p++) {
                zs_2d[i * MAXZ + j][p] = ko_gf_mul(sa[p], sb[p]);
                ze_2d[i * MAXZ + j][p] = ko_gf_mul(ea[p], eb[p]);
            }
            ko_recon_secrets_2ddeg(sec, zs_2d[i * MAXZ + j]);
            ko_share_secrets_ddeg(zs_d[i * MAXZ + j], sec, tp);
            ko_recon_secrets_2ddeg(sec, ze_2d[i * MAXZ + j]);
            ko_share_secrets_ddeg(ze_d[i * MAXZ + j], sec, tp);
        }
    /* P15 :374-381 */
    for (int i = 0; i < K; i++)
        for (int j = 0; j < Z; j++)
            for (int p = 0; p < N; p++) {
                us[i * MAXZ + j][p] = ko_gf_sub(zs_2d[i * MAXZ + j][p], zs_d[i * MAXZ + j][p]);
                ue[i * MAXZ + j][p] = ko_gf_sub(ze_2d[i * MAXZ + j][p], ze_d[i * MAXZ + j][p]);
            }

    /* P16 :395-444 view hash; only the first K of the 70 beta/gamma enter */
    for (int p = 0; p < N; p++) {
        uint8_t *m = msg;
        memcpy(m, tcomm[p], 32);
        m += 32;
        for (int j = 0; j < K; j++, m += 2) put16(m, s_sh[j][p]);
        for (int j = 0; j < K; j++, m += 2) put16(m, e_sh[j][p]);
        for (int j = 0; j < M; j++, m += 2) put16(m, pre->f_sh[j][p]);
        for (int j = 0; j < M; j++, m += 2) put16(m, pre->ntt_f_sh[j][p]);
        for (int j = 0; j < K; j++, m += 2) put16(m, beta[j][p]);
        for (int j = 0; j < K; j++, m += 2) put16(m, gamma[j][p]);
        for (int j = 0; j < K; j++, m += 2) put16(m, sr_sh[j][p]);
        for (int j = 0; j < K; j++, m += 2) put16(m, er_sh[j][p]);
        for (int j = 0; j < K; j++) {
            for (int k = 0; k < Z; k++, m += 2) put16(m, zs_d[j * MAXZ + k][p]);
            for (int k = 0; k < Z; k++, m += 2) put16(m, ze_d[j * MAXZ + k][p]);
            for (int k = 0; k < Z; k++, m += 2) put16(m, us[j * MAXZ + k][p]);
            for (int k = 0; k < Z; k++, m += 2) put16(m, ue[j * MAXZ + k][p]);
        }
        ko_sha3_256(vdig[p], msg, (size_t)(m - msg));
    }

    /* P17 :445-474 */
    uint8_t ch[32];
    uint16_t I[KO_OPENED], rest[KO_REST];
    derive_opened(&vdig[0][0], ch, I);
    uint8_t in_I[KO_PARTIES];
    memset(in_I, 0, sizeof in_I);
    for (int i = 0; i < KO_OPENED; i++) in_I[I[i]] = 1;
    for (int p = 0, j = 0; p < N; p++)
        if (!in_I[p]) rest[j++] = (uint16_t)p;

    /* P18 :480-537 wire image */
    memset(pi, 0, P.proof_bytes);
#define FLD(id) (pi + P.off[id])
    for (int i = 0; i < KO_OPENED; i++) {
        const int p = I[i];
        put16(FLD(KO_F_I) + 2 * i, I[i]);
        for (int j = 0; j < M; j++) {
            put16(FLD(KO_F_F) + 2 * ((size_t)i * M + j), pre->f_sh[j][p]);
            put16(FLD(KO_F_NTTF) + 2 * ((size_t)i * M + j), pre->ntt_f_sh[j][p]);
        }
        for (int j = 0; j < K; j++) {
            const size_t o = 2 * ((size_t)i * K + j);
            put16(FLD(KO_F_S) + o, s_sh[j][p]);
            put16(FLD(KO_F_E) + o, e_sh[j][p]);
            put16(FLD(KO_F_NTTS) + o, ntt_s_sh[j][p]);
            put16(FLD(KO_F_NTTE) + o, ntt_e_sh[j][p]);
            put16(FLD(KO_F_NTTAR) + o, ntt_Ar_sh[j][p]);
            put16(FLD(KO_F_NTTAS) + o, ntt_As_sh[j][p]);
            for (int k = 0; k < E; k++) {
                put16(FLD(KO_F_SSUB) + 2 * (((size_t)i * K + j) * E + k), s_sub[j * MAXE + k][p]);
                put16(FLD(KO_F_ESUB) + 2 * (((size_t)i * K + j) * E + k), e_sub[j * MAXE + k][p]);
            }
            for (int k = 0; k < Z; k++) {
                put16(FLD(KO_F_ZS) + 2 * (((size_t)i * K + j) * Z + k), zs_d[j * MAXZ + k][p]);
                put16(FLD(KO_F_ZE) + 2 * (((size_t)i * K + j) * Z + k), ze_d[j * MAXZ + k][p]);
            }
        }
    }
    for (int i = 0; i < KO_REST; i++) {
        const int p = rest[i];
        for (int j = 0; j < KO_NCHK; j++) {
            put16(FLD(KO_F_BETA) + 2 * ((size_t)i * KO_NCHK + j), beta[j][p]);
            put16(FLD(KO_F_GAMMA) + 2 * ((size_t)i * KO_NCHK + j), gamma[j][p]);
        }
        for (int j = 0; j < K; j++) {
            const size_t o = 2 * ((size_t)i * K + j);
            put16(FLD(KO_F_SR) + o, sr_sh[j][p]);
            put16(FLD(KO_F_ER) + o, er_sh[j][p]);
            put16(FLD(KO_F_T) + o, ntt_t_sh[j][p]);
            for (int k = 0; k < E; k++) {
                put16(FLD(KO_F_SETA) + 2 * (((size_t)i * K + j) * E + k), pre->s_eta_sh[j][k][p]);
                put16(FLD(KO_F_EETA) + 2 * (((size_t)i * K + j) * E + k), pre->e_eta_sh[j][k][p]);
            }
            for (int k = 0; k < Z; k++) {
                put16(FLD(KO_F_US) + 2 * (((size_t)i * K + j) * Z + k), us[j * MAXZ + k][p]);
                put16(FLD(KO_F_UE) + 2 * (((size_t)i * K + j) * Z + k), ue[j * MAXZ + k][p]);
            }
        }
        memcpy(FLD(KO_F_TCOMM) + 32 * (size_t)i, tcomm[p], 32);
        memcpy(FLD(KO_F_COMM) + 32 * (size_t)i, vdig[p], 32);
    }
#undef FLD

    if (trace) {
        memcpy(trace->tcomm, tcomm, (size_t)N * 32);
        memcpy(trace->h1, h1, 32);
        memcpy(trace->alpha, alpha, sizeof trace->alpha);
        memcpy(trace->view_digest, vdig, (size_t)N * 32);
        memcpy(trace->ch, ch, 32);
        memcpy(trace->sr_rec, sr_rec, sizeof sr_rec);
        memcpy(trace->er_rec, er_rec, sizeof er_rec);
    }
    free(s_sh); free(e_sh); free(sr_sh); free(er_sh); free(r_sh); free(nttr_sh);
    free(beta); free(gamma); free(ntt_sr_sh); free(ntt_er_sh); free(ntt_s_sh); free(ntt_e_sh);
    free(ntt_Asr_sh); free(ntt_As_sh); free(ntt_Ar_sh); free(ntt_t_sh); free(s_sub); free(e_sub);
    free(zs_d); free(ze_d); free(zs_2d); free(ze_2d); free(us); free(ue); free(tcomm); free(vdig);
}

/* kosk.cpp:72-86 */
void ko_verifiable_keygen(int K, ko_tape *t, uint8_t *pk, uint8_t *sk, uint8_t *pi, ko_trace *trace)
{
    ko_mlwe *raw = (ko_mlwe *)malloc(sizeof *raw);
    ko_pre *pre = ko_pre_alloc();
    ko_keygen(K, t, pk, sk, raw);
    ko_prepare_randomness(K, t, pre);
    ko_prepare_range_proof(K, t, pre);
    ko_prove(K, t, pi, raw, pre, trace);
    ko_pre_free(pre);
    free(raw);
}

/* ======================================================================== */
/* verify (mlwe_verifier.cpp:4-686)                                         */
/* ======================================================================== */

#define FAIL(...)                                          \
    do {                                                   \
        if (why) snprintf(why, whylen, __VA_ARGS__);       \
        ok = 0;                                            \
        goto done;                                         \
    } while (0)

int ko_verify(int K, const uint8_t *pi, const ko_mlwe *mlwe, char *why, size_t whylen)
{
    ko_params P;
    ko_get_params(K, &P);
    tables_init();
    const int M = P.M, E = P.E, Z = P.Z, N = KO_PARTIES, T = KO_OPENED, R = KO_REST;
    int ok = 1;
    if (why && whylen) why[0] = 0;

#define G16(id, idx) get16(pi + P.off[id] + 2 * (size_t)(idx))
    sharerow *beta = calloc(KO_NCHK, sizeof(sharerow)), *gamma = calloc(KO_NCHK, sizeof(sharerow));
    sharerow *sr_sh = calloc(4, sizeof(sharerow)), *er_sh = calloc(4, sizeof(sharerow));
    sharerow *ntt_sr_sh = calloc(4, sizeof(sharerow)), *ntt_er_sh = calloc(4, sizeof(sharerow));
    sharerow *ntt_Asr_sh = calloc(4, sizeof(sharerow)), *t_sh = calloc(4, sizeof(sharerow));
    sharerow *seta_sh = calloc(4 * MAXE, sizeof(sharerow)), *eeta_sh = calloc(4 * MAXE, sizeof(sharerow));
    sharerow *us_sh = calloc(4 * MAXZ, sizeof(sharerow)), *ue_sh = calloc(4 * MAXZ, sizeof(sharerow));
    uint8_t(*tcomm)[32] = calloc(N, 32), (*vdig)[32] = calloc(N, 32);
    uint16_t(*r_op)[8] = calloc(T, sizeof *r_op), (*nttr_op)[8] = calloc(T, sizeof *nttr_op);
    interp_nodes nd = {0, NULL, NULL}, nd2 = {0, NULL, NULL};

    /* V0 :8-19.  The reference indexes in_I[] with unchecked proof data; an
     * index >= N or a repeated index is undefined behaviour there and can
     * never reproduce I in V10, so it is rejected here up front. */
    uint16_t I[KO_OPENED], rest[KO_REST], pos_of[KO_PARTIES];
    uint8_t in_I[KO_PARTIES];
    memset(in_I, 0, sizeof in_I);
    for (int i = 0; i < T; i++) {
        I[i] = G16(KO_F_I, i);
        if (I[i] >= N || in_I[I[i]]) FAIL("malformed opened-party list at %d", i);
        in_I[I[i]] = 1;
        pos_of[I[i]] = (uint16_t)i;
    }
    for (int p = 0, j = 0; p < N; p++)
        if (!in_I[p]) rest[j++] = (uint16_t)p;

    /* V1 :22-65 */
    uint8_t msg[1024];
    for (int i = 0; i < T; i++) {
        uint8_t *m = msg;
        for (int j = 0; j < K; j++, m += 2) put16(m, G16(KO_F_S, i * K + j));
        for (int j = 0; j < K; j++, m += 2) put16(m, G16(KO_F_E, i * K + j));
        for (int j = 0; j < M; j++, m += 2) put16(m, G16(KO_F_F, i * M + j));
        for (int j = 0; j < M; j++, m += 2) put16(m, G16(KO_F_NTTF, i * M + j));
        ko_sha3_256(tcomm[I[i]], msg, (size_t)(m - msg));
    }
    for (int i = 0; i < R; i++) memcpy(tcomm[rest[i]], pi + P.off[KO_F_TCOMM] + 32 * (size_t)i, 32);
    uint8_t h1[32];
    uint16_t alpha[KO_NCHK + 8];
    uint16_t pw[KO_NCHK + 8][MAXM];
    derive_alpha(&P, &tcomm[0][0], h1, alpha, pw);

    /* V2 :67-124 */
    uint16_t fv[MAXM], tv[MAXM];
    for (int i = 0; i < T; i++) {
        for (int k = 0; k < M; k++) {
            fv[k] = G16(KO_F_F, i * M + k);
            tv[k] = G16(KO_F_NTTF, i * M + k);
        }
        for (int j = 0; j < KO_NCHK; j++) {
            beta[j][I[i]] = lincomb(pw[j], fv, fv[0], M);
            gamma[j][I[i]] = lincomb(pw[j], tv, tv[0], M);
        }
        for (int j = 0; j < P.V; j++) { /* V3 :148-170 */
            r_op[i][j] = lincomb(pw[KO_NCHK + j], fv, fv[KO_NCHK + 1], M);
            nttr_op[i][j] = lincomb(pw[KO_NCHK + j], tv, tv[KO_NCHK + 1], M);
        }
    }
    for (int i = 0; i < R; i++)
        for (int j = 0; j < KO_NCHK; j++) {
            beta[j][rest[i]] = G16(KO_F_BETA, i * KO_NCHK + j);
            gamma[j][rest[i]] = G16(KO_F_GAMMA, i * KO_NCHK + j);
        }
    for (int j = 0; j < KO_NCHK; j++) {
        uint16_t bs[256], gs[256];
        int16_t bp[256];
        ko_recon_secrets_ddeg(bs, beta[j]);
        ko_recon_secrets_ddeg(gs, gamma[j]);
        for (int k = 0; k < 256; k++) bp[k] = ko_gf_decode(bs[k]);
        ko_poly_ntt(bp);
        for (int k = 0; k < 256; k++)
            if (bp[k] != ko_gf_decode(gs[k])) FAIL("Check failed for beta[%d] and gamma[%d] relation.", j, j);
    }

    /* V4 :173-247 */
    uint16_t xs[KO_DEG2 + 1], ys[KO_DEG2 + 1], yval[KO_DEG + 1];
    for (int j = 0; j < KO_DEG2 + 1; j++) xs[j] = (uint16_t)(rest[j] + 256);
    interp_nodes_init(&nd, xs, KO_DEG + 1);
    interp_nodes_init(&nd2, xs, KO_DEG2 + 1);
    int16_t srp[4 * 256], erp[4 * 256];
    uint16_t sr_rnd[4][KO_DEG + 1], er_rnd[4][KO_DEG + 1];
    for (int i = 0; i < K; i++) {
        for (int j = 0; j < KO_DEG + 1; j++) ys[j] = G16(KO_F_SR, j * K + i);
        interp_eval_range(&nd, ys, yval, 0, KO_DEG + 1);
        for (int j = 0; j < 256; j++) srp[256 * i + j] = (int16_t)yval[j]; /* gf3329_int_to_int16, :211 */
        ko_recompute_share_secrets_ddeg(sr_sh[i], yval);
        for (int j = 256; j < KO_DEG + 1; j++) sr_rnd[i][j] = sr_sh[i][j - 256];
        for (int j = 0; j < KO_DEG + 1; j++) ys[j] = G16(KO_F_ER, j * K + i);
        interp_eval_range(&nd, ys, yval, 0, KO_DEG + 1);
        for (int j = 0; j < 256; j++) erp[256 * i + j] = (int16_t)yval[j];
        ko_recompute_share_secrets_ddeg(er_sh[i], yval);
        for (int j = 256; j < KO_DEG + 1; j++) er_rnd[i][j] = er_sh[i][j - 256];
        for (int k = 0; k < R; k++) {
            if (sr_sh[i][rest[k]] != G16(KO_F_SR, k * K + i)) FAIL("s + r share error at %d.", rest[k] + 256);
            if (er_sh[i][rest[k]] != G16(KO_F_ER, k * K + i)) FAIL("e + r share error at %d.", rest[k] + 256);
        }
    }

    /* V5 :257-284 */
    for (int i = 0; i < K; i++) {
        ko_poly_ntt(srp + 256 * i);
        ko_poly_ntt(erp + 256 * i);
        for (int j = 0; j < 256; j++) {
            sr_rnd[i][j] = ko_gf_encode(srp[256 * i + j]);
            er_rnd[i][j] = ko_gf_encode(erp[256 * i + j]);
        }
        ko_recompute_share_secrets_ddeg(ntt_sr_sh[i], sr_rnd[i]);
        ko_recompute_share_secrets_ddeg(ntt_er_sh[i], er_rnd[i]);
    }
    for (int i = 0; i < K; i++)
        for (int j = 0; j < T; j++) {
            if (G16(KO_F_NTTS, j * K + i) != ko_gf_sub(ntt_sr_sh[i][I[j]], nttr_op[j][i]))
                FAIL("Check failed for NTT(s[%d]) at view %d.", i, I[j]);
            if (G16(KO_F_NTTE, j * K + i) != ko_gf_sub(ntt_er_sh[i][I[j]], nttr_op[j][i + K]))
                FAIL("Check failed for NTT(e[%d]) at view %d.", i, I[j]);
        }

    /* V6 :287-312 */
    for (int i = 0; i < K; i++) {
        int16_t asr[256];
        uint16_t rnd[KO_DEG + 1];
        ko_polyvec_basemul_acc(asr, &mlwe->A[i][0][0], srp, K);
        ko_poly_tomont(asr);
        for (int j = 0; j < 256; j++) rnd[j] = ko_gf_encode(asr[j]);
        for (int j = 256; j < KO_DEG + 1; j++) rnd[j] = sr_rnd[i][j];
        ko_recompute_share_secrets_ddeg(ntt_Asr_sh[i], rnd);
    }
    for (int i = 0; i < K; i++)
        for (int j = 0; j < T; j++)
            if (ntt_Asr_sh[i][I[j]] != ko_gf_add(G16(KO_F_NTTAS, j * K + i), G16(KO_F_NTTAR, j * K + i)))
                FAIL("Check failed for NTT(A*(s[%d]+r)) at view %d.", i, I[j]);

    /* V7 :316-376 */
    for (int i = 0; i < K; i++) {
        int16_t tp[256];
        for (int j = 0; j < KO_DEG + 1; j++) ys[j] = G16(KO_F_T, j * K + i);
        interp_eval_range(&nd, ys, yval, 0, KO_DEG + 1);
        for (int j = 0; j < 256; j++) tp[j] = (int16_t)yval[j];
        ko_recompute_share_secrets_ddeg(t_sh[i], yval);
        ko_poly_reduce(tp); /* :354 */
        for (int j = 0; j < 256; j++)
            if (ko_gf_encode(tp[j]) != ko_gf_encode(mlwe->t[i][j])) FAIL("Check failed for t[%d] at %d.", i, j);
    }
    for (int i = 0; i < K; i++)
        for (int j = 0; j < T; j++)
            if (t_sh[i][I[j]] != ko_gf_add(G16(KO_F_NTTAS, j * K + i), G16(KO_F_NTTE, j * K + i)))
                FAIL("Check failed for t = A*s + e at view %d.", I[j]);

    /* V8 :382-466 */
    for (int i = 0; i < K; i++)
        for (int j = 0; j < E; j++) {
            const uint16_t cur = ko_gf_sub((uint16_t)j, (uint16_t)P.eta1);
            for (int who = 0; who < 2; who++) {
                const int fid = who ? KO_F_EETA : KO_F_SETA;
                for (int k = 0; k < KO_DEG + 1; k++) ys[k] = G16(fid, (k * K + i) * E + j);
                interp_eval_range(&nd, ys, yval, 0, KO_DEG + 1);
                for (int k = 0; k < 256; k++)
                    if (yval[k] != cur) FAIL("Check failed for %c_eta[%d] at %d (%d != %d).", who ? 'e' : 's', i, k, cur, yval[k]);
                ko_recompute_share_secrets_ddeg(who ? eeta_sh[i * MAXE + j] : seta_sh[i * MAXE + j], yval);
            }
        }
    for (int i = 0; i < K; i++)
        for (int j = 0; j < T; j++)
            for (int k = 0; k < E; k++) {
                if (G16(KO_F_SSUB, (j * K + i) * E + k) != ko_gf_sub(G16(KO_F_S, j * K + i), seta_sh[i * MAXE + k][I[j]]))
                    FAIL("Check failed for s - eta at view %d, eta[%d][%d][%d].", I[j], i, j, k);
                if (G16(KO_F_ESUB, (j * K + i) * E + k) != ko_gf_sub(G16(KO_F_E, j * K + i), eeta_sh[i * MAXE + k][I[j]]))
                    FAIL("Check failed for e - eta at view %d.", I[j]);
            }

    /* V9 :469-581 */
    for (int i = 0; i < K; i++)
        for (int j = 0; j < Z; j++) {
            for (int k = 0; k < T; k++) {
                const size_t b = ((size_t)k * K + i);
                uint16_t sa = j == 0 ? G16(KO_F_SSUB, b * E) : G16(KO_F_ZS, b * Z + j - 1);
                uint16_t ea = j == 0 ? G16(KO_F_ESUB, b * E) : G16(KO_F_ZE, b * Z + j - 1);
                uint16_t s2 = ko_gf_mul(sa, G16(KO_F_SSUB, b * E + j + 1));
                uint16_t e2 = ko_gf_mul(ea, G16(KO_F_ESUB, b * E + j + 1));
                us_sh[i * MAXZ + j][I[k]] = ko_gf_sub(s2, G16(KO_F_ZS, b * Z + j));
                ue_sh[i * MAXZ + j][I[k]] = ko_gf_sub(e2, G16(KO_F_ZE, b * Z + j));
            }
            for (int who = 0; who < 2; who++) {
                const int fid = who ? KO_F_UE : KO_F_US;
                uint16_t z256[256];
                for (int k = 0; k < KO_DEG2 + 1; k++) ys[k] = G16(fid, (k * K + i) * Z + j);
                interp_eval_range(&nd2, ys, z256, 0, 256);
                for (int k = 0; k < 256; k++)
                    if (z256[k] != 0) FAIL("Check failed for %c.u[%d] at %d (%d != 0).", who ? 'e' : 's', i, k, z256[k]);
                uint16_t *row = who ? ue_sh[i * MAXZ + j] : us_sh[i * MAXZ + j];
                for (int k = 0; k < R; k++) row[rest[k]] = G16(fid, (k * K + i) * Z + j);
                ko_recon_secrets_2ddeg(z256, row);
                for (int k = 0; k < 256; k++)
                    if (z256[k] != 0) FAIL("Check failed for inconsistency of %c.u2d[%d] shares at %d.", who ? 'e' : 's', i, k);
            }
        }

    /* V10 :584-683 */
    for (int i = 0; i < T; i++) {
        uint8_t *m = msg;
        const int p = I[i];
        memcpy(m, tcomm[p], 32);
        m += 32;
        for (int j = 0; j < K; j++, m += 2) put16(m, G16(KO_F_S, i * K + j));
        for (int j = 0; j < K; j++, m += 2) put16(m, G16(KO_F_E, i * K + j));
        for (int j = 0; j < M; j++, m += 2) put16(m, G16(KO_F_F, i * M + j));
        for (int j = 0; j < M; j++, m += 2) put16(m, G16(KO_F_NTTF, i * M + j));
        for (int j = 0; j < K; j++, m += 2) put16(m, beta[j][p]);
        for (int j = 0; j < K; j++, m += 2) put16(m, gamma[j][p]);
        for (int j = 0; j < K; j++, m += 2) put16(m, sr_sh[j][p]);
        for (int j = 0; j < K; j++, m += 2) put16(m, er_sh[j][p]);
        for (int j = 0; j < K; j++) {
            for (int k = 0; k < Z; k++, m += 2) put16(m, G16(KO_F_ZS, (i * K + j) * Z + k));
            for (int k = 0; k < Z; k++, m += 2) put16(m, G16(KO_F_ZE, (i * K + j) * Z + k));
            for (int k = 0; k < Z; k++, m += 2) put16(m, us_sh[j * MAXZ + k][p]);
            for (int k = 0; k < Z; k++, m += 2) put16(m, ue_sh[j * MAXZ + k][p]);
        }
        ko_sha3_256(vdig[p], msg, (size_t)(m - msg));
    }
    for (int i = 0; i < R; i++) memcpy(vdig[rest[i]], pi + P.off[KO_F_COMM] + 32 * (size_t)i, 32);
    {
        uint8_t ch[32];
        uint16_t I2[KO_OPENED];
        derive_opened(&vdig[0][0], ch, I2);
        for (int i = 0; i < T; i++)
            if (I2[i] != I[i]) FAIL("Check failed for reom_I[%d]=%d (pi.I[%d]=%d).", i, I2[i], i, I[i]);
    }
    (void)pos_of;

done:
    if (nd.x) interp_nodes_free(&nd);
    if (nd2.x) interp_nodes_free(&nd2);
    free(beta); free(gamma); free(sr_sh); free(er_sh); free(ntt_sr_sh); free(ntt_er_sh);
    free(ntt_Asr_sh); free(t_sh); free(seta_sh); free(eeta_sh); free(us_sh); free(ue_sh);
    free(tcomm); free(vdig); free(r_op); free(nttr_op);
#undef G16
    return ok;
}

/* kosk.cpp:88-117 */
int ko_kosk_verify(int K, const uint8_t *pi, const uint8_t *pk, char *why, size_t whylen)
{
    ko_mlwe *raw = (ko_mlwe *)calloc(1, sizeof *raw);
    int16_t A[4 * 4 * 256];
    for (int i = 0; i < K; i++) ko_poly_frombytes(raw->t[i], pk + 384 * i);
    ko_gen_matrix(A, pk + 384 * K, 0, K);
    for (int i = 0; i < K; i++)
        for (int j = 0; j < K; j++) memcpy(raw->A[i][j], A + ((size_t)i * K + j) * 256, 512);
    int ok = ko_verify(K, pi, raw, why, whylen);
    free(raw);
    return ok;
}

/* main.cpp:18-94 style timing, for bench.py's cpu_baseline leg */
int ko_bench(int K, int nproofs, const uint8_t *tapes, size_t tape_stride,
             double *sec_keygen_prove, double *sec_verify)
{
    ko_params P;
    if (ko_get_params(K, &P)) return -1;
    uint8_t *pk = malloc(P.pk_bytes), *sk = malloc(P.sk_bytes), *pi = malloc(P.proof_bytes);
    double tp = 0, tv = 0;
    int good = 0;
    for (int b = 0; b < nproofs; b++) {
        ko_tape t;
        ko_tape_init(&t, tapes + (size_t)b * tape_stride, P.tape_bytes);
        clock_t c0 = clock();
        ko_verifiable_keygen(K, &t, pk, sk, pi, NULL);
        clock_t c1 = clock();
        good += ko_kosk_verify(K, pi, pk, NULL, 0);
        clock_t c2 = clock();
        tp += (double)(c1 - c0) / CLOCKS_PER_SEC;
        tv += (double)(c2 - c1) / CLOCKS_PER_SEC;
    }
    if (sec_keygen_prove) *sec_keygen_prove = tp;
    if (sec_verify) *sec_verify = tv;
    free(pk); free(sk); free(pi);
    return good;
}
