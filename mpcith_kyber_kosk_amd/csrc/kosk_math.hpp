// Field / ring / Keccak primitives usable from both host and gfx950 device code.
//
// Reference semantics restated: utils/gf3329.c:274-323 (canonical GF(3329)),
// kyber/reduce.c:16-42 (Montgomery / Barrett), kyber/ntt.c:39-56,:139-146
// (zetas, basemul), kyber/fips202.c:82-344 (Keccak-f[1600]).
#pragma once
#include "kosk_params.hpp"

namespace kosk {

// ---------------------------------------------------------------- GF(3329) --
KOSK_HD inline uint32_t gf_add(uint32_t a, uint32_t b) { uint32_t s = a + b; return s >= (uint32_t)Q ? s - Q : s; }
KOSK_HD inline uint32_t gf_sub(uint32_t a, uint32_t b) { return a >= b ? a - b : a + Q - b; }
KOSK_HD inline uint32_t gf_mul(uint32_t a, uint32_t b) { return a * b % (uint32_t)Q; }
// canonical representative of any int32
KOSK_HD inline uint32_t gf_from_i32(int32_t a) { int32_t r = a % Q; return (uint32_t)(r < 0 ? r + Q : r); }
// centred representative in [-1664, 1664] of a canonical value (decode_from_gf3329)
KOSK_HD inline int32_t gf_center(uint32_t a) { return a > (uint32_t)(Q / 2) ? (int32_t)a - Q : (int32_t)a; }
// encode_to_gf3329 for |a| < q
KOSK_HD inline uint32_t gf_encode(int32_t a) { return (uint32_t)(a < 0 ? a + Q : a); }

// ------------------------------------------------------------ Kyber reduce --
constexpr int32_t QINV = -3327; // q^-1 mod 2^16

// reduce.c:16-23; a in (-q*2^15, q*2^15), result in (-q, q)
KOSK_HD inline int32_t montgomery_reduce(int32_t a)
{
    int32_t t = (int16_t)((int16_t)a * (int16_t)QINV);
    return (a - t * Q) >> 16;
}
// reduce.c:35-42; centred representative of an int16-range value
KOSK_HD inline int32_t barrett_reduce(int32_t a)
{
    constexpr int32_t v = ((1 << 26) + Q / 2) / Q;
    int32_t t = (v * a + (1 << 25)) >> 26;
    return a - t * Q;
}
KOSK_HD inline int32_t fqmul(int32_t a, int32_t b) { return montgomery_reduce(a * b); }

// zetas[k] = 17^bitrev7(k) * 2^16 mod q, centred (recipe of ntt.c:7-37)
struct ZetaTable {
    int16_t z[128];
    constexpr ZetaTable() : z()
    {
        int32_t pw[128] = {};
        pw[0] = 1;
        for (int i = 1; i < 128; i++) pw[i] = pw[i - 1] * 17 % Q;
        for (int i = 0; i < 128; i++) {
            int br = 0;
            for (int b = 0; b < 7; b++) br |= ((i >> b) & 1) << (6 - b);
            int32_t v = pw[br] * 2285 % Q;
            if (v > Q / 2) v -= Q;
            z[i] = (int16_t)v;
        }
    }
};
static constexpr ZetaTable kZetas{};

// The same zetas as the two operands of the NTT kernel's four-instruction butterfly (kosk_kernels.hip, ntt_bfly):
//   zq[k] = z[k] * q^-1 mod 2^16 -- m = lo16(a * zq) is the Montgomery factor (int16)(a z QINV) of reduce.c:19 --
//   zz[k] = { lo16: z[k], hi16: -q } -- one v_dot2_i32_i16 of {a, m} with it is a z - m q, whose high half is fqmul(a, z).
struct ZetaTableDot {
    struct alignas(8) Pair { uint32_t zq, zz; }; // one 8-byte load per zeta
    Pair e[128];
    constexpr ZetaTableDot() : e()
    {
        const ZetaTable t{};
        for (int i = 0; i < 128; i++) {
            e[i].zq = (uint32_t)((int32_t)t.z[i] * QINV) & 0xFFFFu;
            e[i].zz = ((uint32_t)(uint16_t)t.z[i]) | ((uint32_t)(uint16_t)(-Q) << 16);
        }
    }
};

// plain (non-Montgomery) roots 17^bitrev7(k) mod q, centred, as floats: the packed-fp32 NTT variant
struct ZetaTableF {
    float z[128];
    constexpr ZetaTableF() : z()
    {
        int32_t pw[128] = {};
        pw[0] = 1;
        for (int i = 1; i < 128; i++) pw[i] = pw[i - 1] * 17 % Q;
        for (int i = 0; i < 128; i++) {
            int br = 0;
            for (int b = 0; b < 7; b++) br |= ((i >> b) & 1) << (6 - b);
            int32_t v = pw[br];
            if (v > Q / 2) v -= Q;
            z[i] = (float)v;
        }
    }
};

// ------------------------------------------------------------------ Keccak --
struct KeccakConsts {
    uint64_t rc[24];
    constexpr KeccakConsts() : rc()
    {
        // FIPS 202 3.2.5: rc bits from the LFSR x^8+x^6+x^5+x^4+1
        uint8_t lfsr = 1;
        for (int r = 0; r < 24; r++) {
            uint64_t c = 0;
            for (int j = 0; j < 7; j++) {
                if (lfsr & 1) c |= 1ULL << ((1 << j) - 1);
                lfsr = (uint8_t)((lfsr << 1) ^ ((lfsr & 0x80) ? 0x71 : 0));
            }
            rc[r] = c;
        }
    }
};
static constexpr KeccakConsts kKeccak{};

// rho offsets indexed by lane x + 5y (FIPS 202 table 2)
static constexpr int kRho[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43,
                                 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};

template <int N>
KOSK_HD inline uint64_t rotl64c(uint64_t v)
{
    if constexpr (N == 0) return v;
    else return (v << N) | (v >> (64 - N));
}

// one round, reading a[] and writing o[] (distinct arrays: two rounds per loop
// trip need no register moves for the pi permutation)
template <int X, int Y>
KOSK_HD inline void keccak_rhopi(const uint64_t (&a)[25], const uint64_t (&d)[5], uint64_t (&b)[25])
{
    b[Y + 5 * ((2 * X + 3 * Y) % 5)] = rotl64c<kRho[X + 5 * Y]>(a[X + 5 * Y] ^ d[X]);
}

KOSK_HD inline void keccak_round(const uint64_t (&a)[25], uint64_t (&o)[25], uint64_t rc)
{
    uint64_t c[5], d[5], b[25];
#pragma unroll
    for (int x = 0; x < 5; x++) c[x] = a[x] ^ a[x + 5] ^ a[x + 10] ^ a[x + 15] ^ a[x + 20];
#pragma unroll
    for (int x = 0; x < 5; x++) d[x] = c[(x + 4) % 5] ^ rotl64c<1>(c[(x + 1) % 5]);
    keccak_rhopi<0, 0>(a, d, b); keccak_rhopi<1, 0>(a, d, b); keccak_rhopi<2, 0>(a, d, b); keccak_rhopi<3, 0>(a, d, b); keccak_rhopi<4, 0>(a, d, b);
    keccak_rhopi<0, 1>(a, d, b); keccak_rhopi<1, 1>(a, d, b); keccak_rhopi<2, 1>(a, d, b); keccak_rhopi<3, 1>(a, d, b); keccak_rhopi<4, 1>(a, d, b);
    keccak_rhopi<0, 2>(a, d, b); keccak_rhopi<1, 2>(a, d, b); keccak_rhopi<2, 2>(a, d, b); keccak_rhopi<3, 2>(a, d, b); keccak_rhopi<4, 2>(a, d, b);
    keccak_rhopi<0, 3>(a, d, b); keccak_rhopi<1, 3>(a, d, b); keccak_rhopi<2, 3>(a, d, b); keccak_rhopi<3, 3>(a, d, b); keccak_rhopi<4, 3>(a, d, b);
    keccak_rhopi<0, 4>(a, d, b); keccak_rhopi<1, 4>(a, d, b); keccak_rhopi<2, 4>(a, d, b); keccak_rhopi<3, 4>(a, d, b); keccak_rhopi<4, 4>(a, d, b);
#pragma unroll
    for (int y = 0; y < 25; y += 5) {
#pragma unroll
        for (int x = 0; x < 5; x++) o[y + x] = b[y + x] ^ (~b[y + (x + 1) % 5] & b[y + (x + 2) % 5]);
    }
    o[0] ^= rc;
}

#if defined(__HIPCC__)
__constant__ static const uint64_t kKeccakRcDev[24] = {
    0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL,
    0x000000000000808bULL, 0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL,
    0x000000000000008aULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000aULL,
    0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL, 0x8000000000008003ULL,
    0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
    0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
#endif
#if defined(__HIP_DEVICE_COMPILE__)
#define KOSK_RC(i) kKeccakRcDev[i]
#else
#define KOSK_RC(i) kKeccak.rc[i]
#endif

KOSK_HD inline void keccak_f1600(uint64_t (&s)[25])
{
    uint64_t t[25];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
    for (int r = 0; r < 24; r += 2) {
        keccak_round(s, t, KOSK_RC(r));
        keccak_round(t, s, KOSK_RC(r + 1));
    }
}

} // namespace kosk
