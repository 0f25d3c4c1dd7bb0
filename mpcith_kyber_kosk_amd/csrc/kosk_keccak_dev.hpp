// Keccak-f[1600] for gfx950: the 25 lanes live as 2 x 25 32-bit VGPRs so that
// every 64-bit rotation is two v_alignbit_b32 (or a free register swap) and
// theta / chi are 3-input v_bitop3_b32 -- 180 VALU per round instead of the
// ~290 hipcc emits for the portable u64 form in kosk_math.hpp.
// Semantics: kyber/fips202.c:82-344 (KeccakF1600_StatePermute).
#pragma once
#include <hip/hip_runtime.h>

#include "kosk_math.hpp"

namespace kosk {

__device__ __forceinline__ uint32_t kx3(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96); }  // a^b^c
__device__ __forceinline__ uint32_t kchi(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0xD2); } // a^(~b&c)

// rotate the 64-bit value (hi:lo) left by N
template <int N>
__device__ __forceinline__ void krot(uint32_t lo, uint32_t hi, uint32_t &olo, uint32_t &ohi)
{
    if constexpr (N == 0) { olo = lo; ohi = hi; }
    else if constexpr (N == 32) { olo = hi; ohi = lo; }
    else if constexpr (N < 32) {
        ohi = __builtin_amdgcn_alignbit(hi, lo, 32 - N);
        olo = __builtin_amdgcn_alignbit(lo, hi, 32 - N);
    } else {
        ohi = __builtin_amdgcn_alignbit(lo, hi, 64 - N);
        olo = __builtin_amdgcn_alignbit(hi, lo, 64 - N);
    }
}

struct KState {
    uint32_t lo[25], hi[25];
};

template <int X, int Y>
__device__ __forceinline__ void kround_lane(const KState &a, const uint32_t (&clo)[5], const uint32_t (&chi_)[5],
                                            const uint32_t (&rlo)[5], const uint32_t (&rhi)[5], KState &b)
{
    constexpr int i = X + 5 * Y, o = Y + 5 * ((2 * X + 3 * Y) % 5);
    const uint32_t tlo = kx3(a.lo[i], clo[(X + 4) % 5], rlo[(X + 1) % 5]);
    const uint32_t thi = kx3(a.hi[i], chi_[(X + 4) % 5], rhi[(X + 1) % 5]);
    krot<kRho[i]>(tlo, thi, b.lo[o], b.hi[o]);
}

__device__ __forceinline__ void kround(const KState &a, KState &o, uint32_t rclo, uint32_t rchi)
{
    uint32_t clo[5], chi_[5], rlo[5], rhi[5];
#pragma unroll
    for (int x = 0; x < 5; x++) {
        clo[x] = kx3(kx3(a.lo[x], a.lo[x + 5], a.lo[x + 10]), a.lo[x + 15], a.lo[x + 20]);
        chi_[x] = kx3(kx3(a.hi[x], a.hi[x + 5], a.hi[x + 10]), a.hi[x + 15], a.hi[x + 20]);
    }
#pragma unroll
    for (int x = 0; x < 5; x++) krot<1>(clo[x], chi_[x], rlo[x], rhi[x]);
    KState b;
    kround_lane<0, 0>(a, clo, chi_, rlo, rhi, b); kround_lane<1, 0>(a, clo, chi_, rlo, rhi, b); kround_lane<2, 0>(a, clo, chi_, rlo, rhi, b); kround_lane<3, 0>(a, clo, chi_, rlo, rhi, b); kround_lane<4, 0>(a, clo, chi_, rlo, rhi, b);
    kround_lane<0, 1>(a, clo, chi_, rlo, rhi, b); kround_lane<1, 1>(a, clo, chi_, rlo, rhi, b); kround_lane<2, 1>(a, clo, chi_, rlo, rhi, b); kround_lane<3, 1>(a, clo, chi_, rlo, rhi, b); kround_lane<4, 1>(a, clo, chi_, rlo, rhi, b);
    kround_lane<0, 2>(a, clo, chi_, rlo, rhi, b); kround_lane<1, 2>(a, clo, chi_, rlo, rhi, b); kround_lane<2, 2>(a, clo, chi_, rlo, rhi, b); kround_lane<3, 2>(a, clo, chi_, rlo, rhi, b); kround_lane<4, 2>(a, clo, chi_, rlo, rhi, b);
    kround_lane<0, 3>(a, clo, chi_, rlo, rhi, b); kround_lane<1, 3>(a, clo, chi_, rlo, rhi, b); kround_lane<2, 3>(a, clo, chi_, rlo, rhi, b); kround_lane<3, 3>(a, clo, chi_, rlo, rhi, b); kround_lane<4, 3>(a, clo, chi_, rlo, rhi, b);
    kround_lane<0, 4>(a, clo, chi_, rlo, rhi, b); kround_lane<1, 4>(a, clo, chi_, rlo, rhi, b); kround_lane<2, 4>(a, clo, chi_, rlo, rhi, b); kround_lane<3, 4>(a, clo, chi_, rlo, rhi, b); kround_lane<4, 4>(a, clo, chi_, rlo, rhi, b);
#pragma unroll
    for (int y = 0; y < 25; y += 5) {
#pragma unroll
        for (int x = 0; x < 5; x++) {
            o.lo[y + x] = kchi(b.lo[y + x], b.lo[y + (x + 1) % 5], b.lo[y + (x + 2) % 5]);
            o.hi[y + x] = kchi(b.hi[y + x], b.hi[y + (x + 1) % 5], b.hi[y + (x + 2) % 5]);
        }
    }
    o.lo[0] ^= rclo;
    o.hi[0] ^= rchi;
}

__device__ __forceinline__ void keccak_f1600_dev(KState &s)
{
    KState t;
#pragma unroll 1
    for (int r = 0; r < 24; r += 2) {
        const uint64_t c0 = kKeccakRcDev[r], c1 = kKeccakRcDev[r + 1];
        kround(s, t, (uint32_t)c0, (uint32_t)(c0 >> 32));
        kround(t, s, (uint32_t)c1, (uint32_t)(c1 >> 32));
    }
}

__device__ __forceinline__ void kstate_zero(KState &s)
{
#pragma unroll
    for (int i = 0; i < 25; i++) { s.lo[i] = 0; s.hi[i] = 0; }
}

} // namespace kosk
