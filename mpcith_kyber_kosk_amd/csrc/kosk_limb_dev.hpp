// Device helpers shared by the mod-q MFMA kernels (kosk_kernels.hip, kosk_verify_kernels.hip): the int8 limb form of
// field values (kosk_device.hpp: "limb matrix") and the reduction of the recombined accumulators.
#pragma once
#include <hip/hip_runtime.h>

#include "kosk_device.hpp"
#include "kosk_math.hpp"

namespace kosk {

typedef int v4i __attribute__((ext_vector_type(4)));

// XCD-aware workgroup order for one-dimensional grids whose size is a multiple of 8: consecutive workgroup ids go round the
// 8 XCDs (each with its own L2), so workgroups that read the same data are given CONSECUTIVE virtual ids, which this maps to
// ids of one XCD: virtual id = position in this XCD's sequence.
__device__ __forceinline__ int xcd_virtual_id()
{
    constexpr int NXCD = 8;
    const int per_xcd = (int)gridDim.x / NXCD;
    return ((int)blockIdx.x % NXCD) * per_xcd + (int)blockIdx.x / NXCD;
}

// 16 canonical u16 (two uint4) -> 16 low-limb bytes + 16 high-limb bytes of the centred representatives
__device__ __forceinline__ void gm_split16(const uint4 &x0, const uint4 &x1, uint4 &lo, uint4 &hi)
{
    const uint32_t w[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
    uint32_t l[4] = {0, 0, 0, 0}, h[4] = {0, 0, 0, 0};
#pragma unroll
    for (int q = 0; q < 16; q++) {
        uint32_t v = (q & 1) ? (w[q >> 1] >> 16) : (w[q >> 1] & 0xFFFFu);
        if (v >= (uint32_t)Q) v %= Q; // never for honest data; keeps arbitrary input bounded
        int c0, c1;
        limb_split(gf_center(v), c0, c1);
        l[q >> 2] |= ((uint32_t)c0 & 0xFFu) << (8 * (q & 3));
        h[q >> 2] |= ((uint32_t)c1 & 0xFFu) << (8 * (q & 3));
    }
    lo = make_uint4(l[0], l[1], l[2], l[3]);
    hi = make_uint4(h[0], h[1], h[2], h[3]);
}


// ---- packed 16-bit conversion of CANONICAL values (< q) to limbs: both halves of a dword at once -------------------------
typedef unsigned short us2 __attribute__((ext_vector_type(2)));
typedef short ss2 __attribute__((ext_vector_type(2)));

// two canonical values (the halves of v) -> their low limbs (bytes 0 and 2 of lo) and high limbs (bytes 0 and 2 of hi)
__device__ __forceinline__ void limb_split_pk(uint32_t v, uint32_t &lo, uint32_t &hi)
{
    const us2 x = __builtin_bit_cast(us2, v);
    const us2 u = x + (us2){1664, 1664}, t = x - (us2){1665, 1665};          // t = u - q (wraps above u when u < q)
    const us2 m = __builtin_elementwise_min(u, t);                             // (v + 1664) mod q
    const ss2 w = __builtin_bit_cast(ss2, m) - (ss2){1632, 1632};              // centred value + 32, in [-1632, 1696]
    hi = __builtin_bit_cast(uint32_t, w >> (ss2){6, 6});                       // floor(w / 64) = c1
    lo = __builtin_bit_cast(uint32_t, __builtin_bit_cast(ss2, __builtin_bit_cast(uint32_t, w) & 0x003F003Fu) - (ss2){32, 32}); // c0
}
// rows q .. q+3 of one limb, two points: r[i] holds the limb of point 0 in byte 0 and of point 1 in byte 2
__device__ __forceinline__ void limb_pack4(const uint32_t (&r)[4], uint32_t &p0, uint32_t &p1)
{
    const uint32_t t01 = __builtin_amdgcn_perm(r[1], r[0], 0x06020400u), t23 = __builtin_amdgcn_perm(r[3], r[2], 0x06020400u);
    p0 = __builtin_amdgcn_perm(t23, t01, 0x05040100u);
    p1 = __builtin_amdgcn_perm(t23, t01, 0x07060302u);
}


// 16 canonical u16 (two uint4, consecutive k of one row) -> 16 low-limb bytes + 16 high-limb bytes, as gm_split16 but without its
// fold of values >= q (callers guarantee canonical input, or input whose products are discarded): 7 packed instructions per value
// pair and one byte-permute per output dword and limb, about a third of gm_split16's instruction count
__device__ __forceinline__ void gm_split16_pk(const uint4 &x0, const uint4 &x1, uint4 &lo, uint4 &hi)
{
    const uint32_t w[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
    uint32_t l[8], h[8];
#pragma unroll
    for (int q = 0; q < 8; q++) limb_split_pk(w[q], l[q], h[q]); // limbs of values 2q (byte 0) and 2q + 1 (byte 2)
    uint32_t lo4[4], hi4[4];
#pragma unroll
    for (int d = 0; d < 4; d++) {
        lo4[d] = __builtin_amdgcn_perm(l[2 * d + 1], l[2 * d], 0x06040200u);
        hi4[d] = __builtin_amdgcn_perm(h[2 * d + 1], h[2 * d], 0x06040200u);
    }
    lo = make_uint4(lo4[0], lo4[1], lo4[2], lo4[3]);
    hi = make_uint4(hi4[0], hi4[1], hi4[2], hi4[3]);
}

// x mod q for ANY 32-bit x.  t = trunc(float(x) * c), c = (1 - 2^-22) / q rounded to fp32, is floor(x / q) or one less for every
// x < 2^32 (checked exhaustively on the host: tools/float_reduce_check.c, largest x - t q = 4 185 < 2 q), so one unsigned minimum
// finishes (r - q wraps above r when r < q).  Six full-rate vector instructions (convert, multiply, convert, 24-bit multiply-add,
// add, minimum); the integer form (multiply-high + multiply-low, both quarter rate) costs the issue slots of eleven.
__device__ __forceinline__ uint32_t gf_reduce_u32(uint32_t x)
{
    const uint32_t t = (uint32_t)((float)x * 0x1.3afb72p-12f);
    const uint32_t r = (uint32_t)(__mul24((int)t, -Q) + (int)x); // t < 2^21: v_mad_i32_i24, low 32 bits
    return min(r, r - (uint32_t)Q);
}

// Recombination of the three limb products of an int8-limb MFMA accumulation (value = c0 + 64 c1, c0 in [-32, 31], c1 in
// [-26, 26]; 4096 = 767 mod q):  S0 + 64 S1 + 767 S2 mod q.  |S0 + 64 S1 + 767 S2| <= k (1 024 + 64 * 1 664 + 767 * 676) =
// 626 012 k <= 5.21e8 for k <= 832 whatever the operands are (the verifier multiplies values a prover chose), so a bias of
// 200 000 q = 6.66e8 makes the sum a positive number below 2^31; |S2| <= 562 432 fits the 24-bit multiplier.
constexpr int32_t LIMB_BIAS = 200000 * Q;
static_assert(832LL * 626012 < LIMB_BIAS && 832LL * 626012 + LIMB_BIAS < (1LL << 31), "bias covers 13 k-steps of 64");
__device__ __forceinline__ uint32_t gf_reduce_limbs(int32_t s0, int32_t s1, int32_t s2)
{
    return gf_reduce_u32((uint32_t)(__mul24(s2, 767) + s1 * 64 + s0 + LIMB_BIAS));
}

// The same reduction for TWO outputs at once, with the last step in packed 16-bit arithmetic (round 5: k_table_gemm_p, k_lincomb_stream).
// s0 already holds the bias (the first MFMA of an output block accumulates into {LIMB_BIAS, ...} instead of zero), so the
// recombination is two instructions (24-bit multiply-add, shift-add); t = trunc(float(x) c) is floor(x / q) or one less, so
// r = x - t q lies in [0, 2 q) and fits 16 bits: the two r are packed into one register and brought into [0, q) by one packed
// subtract and one packed unsigned minimum -- 7 instead of 9.5 vector instructions per output.
__device__ __forceinline__ uint32_t gf_reduce_limbs_biased_lazy(int32_t s0_biased, int32_t s1, int32_t s2) // -> [0, 2 q)
{
    int32_t y;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(y) : "v"(s2), "s"(767), "v"(s0_biased)); // 767 s2 + (s0 + bias): |s2| < 2^23 for k <= 832 (gfx9 VOP3 takes no literal: the constant sits in a scalar register)
    const uint32_t x = (uint32_t)y + ((uint32_t)s1 << 6);                        // v_lshl_add_u32
    const uint32_t t = (uint32_t)((float)x * 0x1.3afb72p-12f);
    return (uint32_t)(__mul24((int)t, -Q) + (int)x);
}
__device__ __forceinline__ uint32_t gf_reduce_limbs_lazy(int32_t s0, int32_t s1, int32_t s2) // the bias added here (kernels short of registers)
{
    return gf_reduce_limbs_biased_lazy(s0 + LIMB_BIAS, s1, s2);
}
__device__ __forceinline__ uint32_t gf_canon_pair(uint32_t r_lo, uint32_t r_hi) // two values in [0, 2 q) -> packed canonical pair
{
    typedef unsigned short us2_ __attribute__((ext_vector_type(2)));
    const us2_ c = __builtin_bit_cast(us2_, r_lo | (r_hi << 16));
    const us2_ m = c - (us2_){(unsigned short)Q, (unsigned short)Q}; // wraps above c when c < q
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(c, m));
}
// packed a + b mod q of two canonical pairs
__device__ __forceinline__ uint32_t gf_add_pair(uint32_t a, uint32_t b)
{
    typedef unsigned short us2_ __attribute__((ext_vector_type(2)));
    const us2_ c = __builtin_bit_cast(us2_, a) + __builtin_bit_cast(us2_, b); // < 2 q: no carry between the halves
    const us2_ m = c - (us2_){(unsigned short)Q, (unsigned short)Q};
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(c, m));
}

} // namespace kosk
