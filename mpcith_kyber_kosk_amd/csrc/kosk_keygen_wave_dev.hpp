// gen_matrix on the WAVE sponge (round 6): one matrix entry A[i][j] per 64-lane wave, the Keccak state spread over the wave as in
// kosk_fs_dev.hpp (variant B: DPP column sums + ds_bpermute exchanges, ~2.1 us per permutation).
//
// Why: an entry is a chain of one seed hash (prover) + three or more SHAKE128 permutations, each followed by a rejection parse whose
// running count is sequential; on the lane-pair sponge (kosk_keygen_dev.hpp: kp_gen_matrix) that chain is ~9 us per permutation plus
// ~600 parse instructions per block at ONE wave's issue rate -- 56 us for the verifier's launch and the longest role of the prover's
// first launch (k_prover_pre), with 276 x K x K sponges on a machine of 1 024 SIMDs: the chain's length, not the machine, sets the
// time.  Here every vector instruction of the permutation acts on the whole state, and the parse is one step for the whole block: 56
// lanes take one 3-byte group each, two ballots give every accepted candidate its place (rej_uniform's order: group by group, low
// 12 bits first).
//   kosk.cpp:12-14 (sha3_512(d || K) -> rho), kyber/indcpa.c:124-145 (rej_uniform), :168-193 (gen_matrix, not transposed: xof_absorb(rho, j, i)),
//   kyber/symmetric-shake.c:18-29, kyber/fips202.c:461-485.
#pragma once
#include <hip/hip_runtime.h>

#include "kosk_device.hpp"
#include "kosk_fs_dev.hpp"

namespace kosk {

constexpr int KW_ST_WORDS = 64, KW_SQ_BYTES = 176; // per-wave LDS: the state's words for re-interleaving; one squeezed block (168 bytes)

// seed32: 32 bytes at any alignment.  hash_d: they are the key generation's d and rho = sha3_512(d || K)[0..32) comes first; else they are rho.
// r: the entry's 256 coefficients (canonical int16).  st / sq: this wave's LDS scratch.  All 64 lanes must be active.
__device__ __forceinline__ void kw_gen_matrix(const uint8_t *seed32, bool hash_d, int K, int i, int j, int16_t *__restrict__ r, const XofGuard &xof,
                                              uint32_t *st, uint8_t *sq)
{
    const int lane = threadIdx.x & 63;
    FsSpongeBperm sp;
    sp.setup(lane, nullptr);
    const int word = sp.word();
    const uint32_t half = sp.half();
    // words 0..3 of the first block: the seed
    uint32_t lo = 0, hi = 0;
    if (word < 4) {
        const uint8_t *p = seed32 + 8 * word;
        lo = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
        hi = (uint32_t)p[4] | ((uint32_t)p[5] << 8) | ((uint32_t)p[6] << 16) | ((uint32_t)p[7] << 24);
    }
    uint32_t a;
    if (hash_d) { // sha3_512: rate 72 bytes; d || K || 0x06 ... 0x80
        if (word == 4) lo = (uint32_t)K | (0x06u << 8);
        if (word == 8) hi = 0x80000000u;
        a = fs_deinterleave_half(lo, hi, half);
        sp.permute(a);
        // rho = words 0..3 of the digest: the next sponge's first words as they stand (still interleaved)
        lo = 0; hi = 0;
        if (word == 4) lo = (uint32_t)j | ((uint32_t)i << 8) | (0x1Fu << 16);
        if (word == 20) hi = 0x80000000u; // byte 167 of SHAKE128's rate
        a = (word < 4 ? a : 0u) ^ fs_deinterleave_half(lo, hi, half);
    } else {
        if (word == 4) lo = (uint32_t)j | ((uint32_t)i << 8) | (0x1Fu << 16);
        if (word == 20) hi = 0x80000000u;
        a = fs_deinterleave_half(lo, hi, half);
    }
    int ctr = 0;
#pragma unroll 1
    for (int blk = 0; blk < xof.max_blocks && ctr < 256; blk++) { // (uniform)
        sp.permute(a);
        st[lane] = a;
        __builtin_amdgcn_wave_barrier();
        if (lane < 21) {
            const int x = lane % 5, y = lane / 5;
            uint32_t l2, h2;
            fs_interleave(st[FsSpongeBperm::lane_of(x, y, 0)], st[FsSpongeBperm::lane_of(x, y, 1)], l2, h2);
            *reinterpret_cast<uint2 *>(sq + 8 * lane) = make_uint2(l2, h2);
        }
        __builtin_amdgcn_wave_barrier();
        // rej_uniform over the block's 56 three-byte groups: lane t takes group t
        uint32_t v0 = 0xFFFFu, v1 = 0xFFFFu;
        if (lane < 56) {
            const uint32_t x = (uint32_t)sq[3 * lane] | ((uint32_t)sq[3 * lane + 1] << 8) | ((uint32_t)sq[3 * lane + 2] << 16);
            v0 = x & 0xFFFu;
            v1 = x >> 12;
        }
        const bool ok0 = v0 < (uint32_t)Q, ok1 = v1 < (uint32_t)Q;
        const uint64_t m0 = __builtin_amdgcn_ballot_w64(ok0), m1 = __builtin_amdgcn_ballot_w64(ok1);
        const uint64_t below = ((uint64_t)1 << lane) - 1;
        const int at0 = ctr + __popcll(m0 & below) + __popcll(m1 & below), at1 = at0 + (ok0 ? 1 : 0);
        if (ok0 && at0 < 256) r[at0] = (int16_t)v0;
        if (ok1 && at1 < 256) r[at1] = (int16_t)v1;
        ctr += __popcll(m0) + __popcll(m1);
        __builtin_amdgcn_wave_barrier(); // the next block's bytes overwrite sq
    }
    if (ctr < 256) { // block limit reached (XofGuard): what the caller's error check reports, with a defined result
        for (int c = ctr + lane; c < 256; c += 64) r[c] = 0;
        if (lane == 0 && xof.err) *reinterpret_cast<volatile uint32_t *>(xof.err) = DEVERR_XOF_BLOCKS;
    }
}

} // namespace kosk
