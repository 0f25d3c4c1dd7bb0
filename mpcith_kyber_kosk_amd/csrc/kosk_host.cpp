// Host-side crypto and helpers of the product path (see kosk_host.hpp).
#include "kosk_host.hpp"

#include <sys/random.h>
#include <unistd.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
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
    uint8_t ext[34], buf[168 * 8];
    memcpy(ext, seed, 32);
    for (int i = 0; i < K; i++)
        for (int j = 0; j < K; j++) {
            ext[32] = (uint8_t)j;
            ext[33] = (uint8_t)i;
            size_t have = 168 * 4;
            shake128(buf, have, ext, 34);
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
                if (have == sizeof buf) { fprintf(stderr, "kosk: gen_matrix XOF prefix exhausted\n"); abort(); }
                have = sizeof buf; // the XOF stream is a prefix-consistent byte stream: re-squeeze longer
                shake128(buf, have, ext, 34);
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

void fs_opened(const uint8_t *digests_all, uint16_t I[NOPEN], uint16_t rest[NREST])
{
    uint8_t ch[32], I_[2 * NOPEN];
    sha3_256(ch, digests_all, (size_t)NPARTY * 32);
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

void pack_gemm_table(const std::vector<uint16_t> &A, int M, int Kdim, int Mpad, int KP, std::vector<uint32_t> &out)
{
    out.assign((size_t)KP * Mpad, 0);
    for (int m = 0; m < M; m++)
        for (int kp = 0; kp < KP; kp++) {
            const int k0 = 2 * kp, k1 = 2 * kp + 1;
            const int32_t v0 = k0 < Kdim ? gf_center(A[(size_t)m * Kdim + k0]) : 0;
            const int32_t v1 = k1 < Kdim ? gf_center(A[(size_t)m * Kdim + k1]) : 0;
            out[(size_t)kp * Mpad + m] = ((uint32_t)v0 & 0xFFFFu) | ((uint32_t)v1 << 16);
        }
}

// -------------------------------------------------------------------- misc --
void parallel_for(int n, int nthreads, const std::function<void(int)> &fn)
{
    if (n <= 0) return;
    if (nthreads > n) nthreads = n;
    if (nthreads <= 1) {
        for (int i = 0; i < n; i++) fn(i);
        return;
    }
    std::atomic<int> next{0};
    std::vector<std::thread> th;
    th.reserve(nthreads);
    for (int t = 0; t < nthreads; t++)
        th.emplace_back([&] {
            for (;;) {
                const int i = next.fetch_add(1);
                if (i >= n) break;
                fn(i);
            }
        });
    for (auto &t : th) t.join();
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
