// Keccak-f[1600] with one state spread over TWO adjacent lanes of a wave ("lane-pair sponge"):
// the even lane holds the low 32 bits of each of the 25 words, the odd lane the high 32 bits.
//
// Why: a gfx950 SIMD issues at most one instruction per ~5 cycles from any ONE wave but one per 1.5-2.7
// cycles overall (profiles/r02_probe_keccak.txt), so a launch with about one wave per SIMD -- the 46-proof
// batch is 1046 waves on 1024 SIMDs -- is issue-latency bound and quantised to whole waves.  Splitting a
// state over a lane pair gives twice the waves with 2/3 of the instructions each: theta's column sums,
// theta's application, chi and iota act on the two halves independently; only the 64-bit rotations
// (theta's rot-1 and rho) need the partner's half, one v_mov_b32_dpp quad_perm:[1,0,3,2] each.
// Per lane and round: 60 v_bitop3 + 29 v_alignbit + 29 DPP moves + 2 = 120 instructions (one-lane form: 180).
// Semantics: kyber/fips202.c:82-344 (KeccakF1600_StatePermute).
#pragma once
#include <hip/hip_runtime.h>

#include "kosk_keccak_dev.hpp"

namespace kosk {

// the partner lane's value (lane ^ 1)
__device__ __forceinline__ uint32_t kpartner(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1 /* quad_perm:[1,0,3,2] */, 0xF, 0xF, true);
}

// my half of rotl64(word, N), given my half `m` and the partner's half `p` of the word.
// low half of rotl by N<32: (lo << N) | (hi >> (32-N)); high half: (hi << N) | (lo >> (32-N)) -- the same
// expression in (mine, partner); for N>32 the roles swap.
template <int N>
__device__ __forceinline__ uint32_t krot_half(uint32_t m, uint32_t p)
{
    if constexpr (N == 0) return m;
    else if constexpr (N == 32) return p;
    else if constexpr (N < 32) return __builtin_amdgcn_alignbit(m, p, 32 - N);
    else return __builtin_amdgcn_alignbit(p, m, 64 - N);
}

struct KHalf {
    uint32_t w[25];
};

template <int X, int Y>
__device__ __forceinline__ void ksplit_lane(const KHalf &a, const uint32_t (&c)[5], const uint32_t (&r)[5], KHalf &b)
{
    constexpr int i = X + 5 * Y, o = Y + 5 * ((2 * X + 3 * Y) % 5);
    const uint32_t t = kx3(a.w[i], c[(X + 4) % 5], r[(X + 1) % 5]);
    if constexpr (kRho[i] == 0) b.w[o] = t;
    else b.w[o] = krot_half<kRho[i]>(t, kpartner(t));
}

// rc = this lane's half of the round constant
__device__ __forceinline__ void ksplit_round(const KHalf &a, KHalf &o, uint32_t rc)
{
    uint32_t c[5], r[5];
#pragma unroll
    for (int x = 0; x < 5; x++) c[x] = kx3(kx3(a.w[x], a.w[x + 5], a.w[x + 10]), a.w[x + 15], a.w[x + 20]);
#pragma unroll
    for (int x = 0; x < 5; x++) r[x] = krot_half<1>(c[x], kpartner(c[x]));
    KHalf b;
    ksplit_lane<0, 0>(a, c, r, b); ksplit_lane<1, 0>(a, c, r, b); ksplit_lane<2, 0>(a, c, r, b); ksplit_lane<3, 0>(a, c, r, b); ksplit_lane<4, 0>(a, c, r, b);
    ksplit_lane<0, 1>(a, c, r, b); ksplit_lane<1, 1>(a, c, r, b); ksplit_lane<2, 1>(a, c, r, b); ksplit_lane<3, 1>(a, c, r, b); ksplit_lane<4, 1>(a, c, r, b);
    ksplit_lane<0, 2>(a, c, r, b); ksplit_lane<1, 2>(a, c, r, b); ksplit_lane<2, 2>(a, c, r, b); ksplit_lane<3, 2>(a, c, r, b); ksplit_lane<4, 2>(a, c, r, b);
    ksplit_lane<0, 3>(a, c, r, b); ksplit_lane<1, 3>(a, c, r, b); ksplit_lane<2, 3>(a, c, r, b); ksplit_lane<3, 3>(a, c, r, b); ksplit_lane<4, 3>(a, c, r, b);
    ksplit_lane<0, 4>(a, c, r, b); ksplit_lane<1, 4>(a, c, r, b); ksplit_lane<2, 4>(a, c, r, b); ksplit_lane<3, 4>(a, c, r, b); ksplit_lane<4, 4>(a, c, r, b);
#pragma unroll
    for (int y = 0; y < 25; y += 5) {
#pragma unroll
        for (int x = 0; x < 5; x++) o.w[y + x] = kchi(b.w[y + x], b.w[y + (x + 1) % 5], b.w[y + (x + 2) % 5]);
    }
    o.w[0] ^= rc;
}

// `hi` is true on the lane that holds the high halves (odd lane).  All 64 lanes of the wave must be active
// (the DPP exchange reads the partner lane's register).
__device__ __forceinline__ void keccak_f1600_split(KHalf &s, bool hi)
{
    KHalf t;
#pragma unroll 1
    for (int r = 0; r < 24; r += 2) {
        const uint64_t c0 = kKeccakRcDev[r], c1 = kKeccakRcDev[r + 1];
        ksplit_round(s, t, hi ? (uint32_t)(c0 >> 32) : (uint32_t)c0);
        ksplit_round(t, s, hi ? (uint32_t)(c1 >> 32) : (uint32_t)c1);
    }
}

} // namespace kosk
