// Device functions of Kyber key generation (kosk.cpp:4-70), one sponge per thread, usable from any kernel so that the
// prover front can run them as extra block ranges of its first launch (kosk_kernels.hip: k_prover_pre).
// The squeezed blocks are parsed straight from the state registers (static lane indices after unrolling): no per-thread
// LDS byte buffer, no dependent LDS reads inside the rejection loop.
//   kg_seed_hash     sha3_512(d || K) -> public seed || noise seed               kosk.cpp:12-14
//   (gen_matrix)     SHAKE128(seed || j || i) + rej_uniform: kosk_keygen_wave_dev.hpp     indcpa.c:124-145, :168-193
//   kg_noise         SHAKE256(noise seed || nonce) + cbd2 / cbd3                  poly.c:225-230, cbd.c:58-107
#pragma once
#include <hip/hip_runtime.h>

#include "kosk_keccak_dev.hpp"
#include "kosk_keccak_split_dev.hpp"
#include "kosk_math.hpp"

namespace kosk {

// dword w of the sponge state seen as a little-endian byte string (w = 2 * lane + half)
template <int WIDX>
__device__ __forceinline__ uint32_t kdword(const KState &s)
{
    return (WIDX & 1) ? s.hi[WIDX >> 1] : s.lo[WIDX >> 1];
}

// fresh state = pad(seed32 || extra[0..nextra) || dom) for a sponge of `rate` bytes; seed32 as 8 dwords
__device__ __forceinline__ void kg_absorb(KState &s, const uint32_t (&seed)[8], uint32_t extra, int nextra, int rate, uint32_t dom)
{
    kstate_zero(s);
#pragma unroll
    for (int l = 0; l < 4; l++) { s.lo[l] = seed[2 * l]; s.hi[l] = seed[2 * l + 1]; }
    s.lo[4] = (extra & ((1u << (8 * nextra)) - 1u)) | (dom << (8 * nextra)); // nextra <= 2
    const int last = rate / 8 - 1;
#pragma unroll
    for (int l = 0; l < 25; l++)
        if (l == last) s.hi[l] ^= 0x80000000u;
}

__device__ __forceinline__ void kg_load_seed(uint32_t (&seed)[8], const uint8_t *p)
{
    if ((reinterpret_cast<uintptr_t>(p) & 3) == 0) {
#pragma unroll
        for (int i = 0; i < 8; i++) seed[i] = reinterpret_cast<const uint32_t *>(p)[i];
    } else {
#pragma unroll
        for (int i = 0; i < 8; i++) seed[i] = (uint32_t)p[4 * i] | ((uint32_t)p[4 * i + 1] << 8) | ((uint32_t)p[4 * i + 2] << 16) | ((uint32_t)p[4 * i + 3] << 24);
    }
}

// the 24 bits at bit offset 24 * T of the squeezed block
template <int T>
__device__ __forceinline__ uint32_t kg_triple(const KState &s)
{
    constexpr int k = (3 * T) / 4, sh = (24 * T) % 32;
    if constexpr (sh == 0) return kdword<k>(s) & 0xFFFFFFu;
    else if constexpr (sh == 8) return kdword<k>(s) >> 8;
    else return __builtin_amdgcn_alignbit(kdword<k + 1>(s), kdword<k>(s), sh) & 0xFFFFFFu;
}

// 8 cbd2 coefficients from 32 bits / 4 cbd3 coefficients from 24 bits (cbd.c:58-107)
__device__ __forceinline__ uint4 kg_cbd2_word(uint32_t x)
{
    const uint32_t d = (x & 0x55555555u) + ((x >> 1) & 0x55555555u);
    uint32_t o[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int32_t c0 = (int32_t)((d >> (8 * q)) & 3) - (int32_t)((d >> (8 * q + 2)) & 3);
        const int32_t c1 = (int32_t)((d >> (8 * q + 4)) & 3) - (int32_t)((d >> (8 * q + 6)) & 3);
        o[q] = ((uint32_t)c0 & 0xFFFFu) | ((uint32_t)c1 << 16);
    }
    return make_uint4(o[0], o[1], o[2], o[3]);
}
__device__ __forceinline__ uint2 kg_cbd3_triple(uint32_t x)
{
    const uint32_t d = (x & 0x00249249u) + ((x >> 1) & 0x00249249u) + ((x >> 2) & 0x00249249u);
    uint32_t o[2];
#pragma unroll
    for (int q = 0; q < 2; q++) {
        const int32_t c0 = (int32_t)((d >> (12 * q)) & 7) - (int32_t)((d >> (12 * q + 3)) & 7);
        const int32_t c1 = (int32_t)((d >> (12 * q + 6)) & 7) - (int32_t)((d >> (12 * q + 9)) & 7);
        o[q] = ((uint32_t)c0 & 0xFFFFu) | ((uint32_t)c1 << 16);
    }
    return make_uint2(o[0], o[1]);
}

// ---- the same three samplers on the LANE-PAIR sponge (kosk_keccak_split_dev.hpp; round 5) -----------------------------------------
// One state on two adjacent lanes (even lane: low halves of the 25 words, odd lane: high halves): 120 instead of 180 vector
// instructions per lane and round.  These samplers are a few waves per launch, each alone on its SIMD, so their time is the
// dependent chain of permutations at ONE wave's issue rate; two thirds of the instructions are two thirds of that time.  All 64
// lanes of a wave must stay active (the DPP exchange reads the partner's registers): callers clamp indices and predicate stores.

// both halves of word l of the state on both lanes of the pair: lo = dword 2 l, hi = dword 2 l + 1 of the squeezed block
__device__ __forceinline__ void kp_full(const KHalf &s, bool hi, int nwords, KState &f)
{
#pragma unroll
    for (int l = 0; l < 25; l++) {
        if (l < nwords) {
            const uint32_t p = kpartner(s.w[l]);
            f.lo[l] = hi ? p : s.w[l];
            f.hi[l] = hi ? s.w[l] : p;
        }
    }
}
// my half of pad(seed32 || extra[0..nextra) || dom) for a sponge of `rate` bytes (kg_absorb)
__device__ __forceinline__ void kp_absorb(KHalf &s, bool hi, const uint32_t (&seed)[8], uint32_t extra, int nextra, int rate, uint32_t dom)
{
#pragma unroll
    for (int l = 0; l < 25; l++) s.w[l] = 0;
#pragma unroll
    for (int l = 0; l < 4; l++) s.w[l] = hi ? seed[2 * l + 1] : seed[2 * l];
    if (!hi) s.w[4] = (extra & ((1u << (8 * nextra)) - 1u)) | (dom << (8 * nextra));
    const int last = rate / 8 - 1;
#pragma unroll
    for (int l = 0; l < 25; l++)
        if (l == last && hi) s.w[l] ^= 0x80000000u;
}
// sha3_512(d[32] || K) on the pair; both lanes end up with both seeds
__device__ __forceinline__ void kp_seed_hash(const uint8_t *d32, int K, bool hi, uint32_t (&pub)[8], uint32_t (&noise)[8])
{
    uint32_t d[8];
    kg_load_seed(d, d32);
    KHalf s;
    kp_absorb(s, hi, d, (uint32_t)K, 1, 72, 0x06);
    keccak_f1600_split(s, hi);
    KState f;
    kp_full(s, hi, 8, f);
#pragma unroll
    for (int l = 0; l < 4; l++) { pub[2 * l] = f.lo[l]; pub[2 * l + 1] = f.hi[l]; noise[2 * l] = f.lo[4 + l]; noise[2 * l + 1] = f.hi[4 + l]; }
}
// (gen_matrix runs on the wave sponge since round 6: kosk_keygen_wave_dev.hpp)
// poly_getnoise_eta1: eta1 == 2: every lane turns its own dwords of the block into coefficients (dword W = 2 l + half -> eight
// coefficients, no exchange); eta1 == 3: 3-byte groups straddle dwords, so both lanes rebuild the block and the even lane runs the
// one-lane parse
__device__ __forceinline__ void kp_noise(const uint32_t (&noise)[8], int nonce, int eta1, bool hi, bool store, int16_t *__restrict__ r)
{
    KHalf s;
    kp_absorb(s, hi, noise, (uint32_t)nonce, 1, 136, 0x1F);
    keccak_f1600_split(s, hi);
    if (eta1 == 2) {
        if (store) {
#pragma unroll
            for (int l = 0; l < 16; l++) reinterpret_cast<uint4 *>(r)[2 * l + (hi ? 1 : 0)] = kg_cbd2_word(s.w[l]);
        }
        return;
    }
    const bool st = store && !hi;
    KState f;
    kp_full(s, hi, 17, f);
    auto go1 = [&]<int... Ts>(std::integer_sequence<int, Ts...>) {
        ((st ? (void)(reinterpret_cast<uint2 *>(r)[Ts] = kg_cbd3_triple(kg_triple<Ts>(f))) : (void)0), ...);
    };
    go1(std::make_integer_sequence<int, 45>{});
    const uint32_t carry = kdword<33>(f) >> 24; // byte 135
    keccak_f1600_split(s, hi);
    kp_full(s, hi, 8, f); // bytes 0 .. 55 of block 2 (triples 45 .. 63 end at byte 56)
    if (st) reinterpret_cast<uint2 *>(r)[45] = kg_cbd3_triple(carry | ((kdword<0>(f) & 0xFFFFu) << 8));
    auto go2 = [&]<int... Us>(std::integer_sequence<int, Us...>) {
        (([&] {
             constexpr int bit = 16 + 24 * Us, k = bit / 32, sh = bit % 32;
             uint32_t x;
             if constexpr (sh == 0) x = kdword<k>(f) & 0xFFFFFFu;
             else if constexpr (sh == 8) x = kdword<k>(f) >> 8;
             else x = __builtin_amdgcn_alignbit(kdword<k + 1>(f), kdword<k>(f), sh) & 0xFFFFFFu;
             if (st) reinterpret_cast<uint2 *>(r)[46 + Us] = kg_cbd3_triple(x);
         }()),
         ...);
    };
    go2(std::make_integer_sequence<int, 18>{});
}

} // namespace kosk
