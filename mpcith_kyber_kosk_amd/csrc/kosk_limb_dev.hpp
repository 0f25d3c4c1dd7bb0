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

__device__ __forceinline__ uint32_t gf_reduce_pos(uint32_t x) // x < 2^32 - q
{
    // t = floor(x * floor(2^32 / q) / 2^32) is floor(x / q) or one less, so x - t q < 2 q: one conditional subtraction,
    // done as an unsigned minimum (r - q wraps above r when r < q)
    const uint32_t t = __umulhi(x, 1290167u);
    const uint32_t r = x - t * (uint32_t)Q;
    return min(r, r - (uint32_t)Q);
}

} // namespace kosk
