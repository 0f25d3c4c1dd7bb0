// Hand-written gfx950 (CDNA4) kernels for the KOSK party-view simulation.
//
// Every kernel works on the row matrix described in kosk_params.hpp:
//   P[proof][row][x]   canonical u16, x = evaluation point (0..1709), RS = 1728.
// Kernel <-> reference loop map:
//   k_commit_hash   K4   mlwe_prover.cpp:116-127, :397-444 ; mlwe_verifier.cpp:23-35, :585-632
//   k_sha3_msgs          kyber/fips202.c:745-754 on message-major input (tests, generic ABI)
//   k_prover_pre         mlwe_prover.cpp:8-14 (SHAKE256 PRF, BE16 % q), ss.cpp:5-11 (tape randoms), witness secrets
//   k_ntt256        K5   kyber/ntt.c:80-95 + poly.c:261-265
//   k_matvec_ntt    K6   polyvec.c:202-214 + poly.c:307-313
//   k_rows_to_limbs, k_gemm_modq  K1/K2 ss.cpp:23-32, :44-51, :63-70, :88-97 (+ verifier interpolation apply)
//   k_lincomb       K3   mlwe_prover.cpp:159-203 ; mlwe_verifier.cpp:68-89, :149-170
//   k_post_*        K7   ss.cpp:101-136 call sites in prove()
//   k_assemble_*    K8   mlwe_prover.cpp:480-537
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <utility>

#include "kosk_device.hpp"
#include "kosk_keccak_dev.hpp"
#include "kosk_keccak_split_dev.hpp"
#include "kosk_keygen_dev.hpp"
#include "kosk_keygen_wave_dev.hpp"
#include "kosk_math.hpp"
#include "kosk_limb_dev.hpp"

namespace kosk {

// =========================================================================
// K4  SHA3-256 view commitments: one party lane per thread, 25 x u64 state
// in VGPRs (v_bitop3_b32 / v_alignbit_b32), message words gathered from the
// row matrix with one coalesced 128-byte line per wave per row.
// =========================================================================
// PIPE: software-pipelined loads (more registers, for launches with few waves per SIMD)
template <int PREFIX_WORDS, int NROWS, bool PIPE>
__global__ __launch_bounds__(64) void k_commit_hash(HashArgs a)
{
    const int lane = blockIdx.x * 64 + threadIdx.x;
    const int g = blockIdx.y;
    if (lane >= a.lanes_per_group) return;
    const int col = a.lane_map ? (int)a.lane_map[(size_t)g * a.lane_map_stride + lane] : lane;
    const uint16_t *__restrict__ base = a.rows + (size_t)g * a.group_stride + a.col_off + col;
    const size_t dig = ((size_t)g * a.out_lanes_per_group + col) * 32;

    constexpr int W = PREFIX_WORDS + NROWS; // u16 words in the message
    constexpr int RATE_W = 68;              // 136-byte rate
    constexpr int NBLK = W / RATE_W + 1;    // pad10*1 always adds to the last (possibly empty) block

    KState s;
    kstate_zero(s);

    if constexpr (PIPE) {
        // Software pipeline over the 136-byte blocks: the 68 row loads of block i+1 are issued BEFORE the
        // permutation of block i, so their latency hides under ~4300 VALU instructions (with one wave per
        // SIMD nothing else would hide it).  34 dwords of look-ahead.
        uint32_t cur[RATE_W / 2], nxt[RATE_W / 2];
        auto load_block = [&](auto blkc, uint32_t (&dst)[RATE_W / 2]) {
            constexpr int blk = decltype(blkc)::value;
    #pragma unroll
            for (int w = 0; w < RATE_W; w += 2) {
                const int gw = blk * RATE_W + w; // first of two u16 words forming half a lane
                uint32_t v = 0;
                if (gw < W) {
                    if (gw + 1 < PREFIX_WORDS) {
                        v = *reinterpret_cast<const uint32_t *>(a.prefix + dig + 2 * gw);
                    } else {
                        v = base[(size_t)(gw - PREFIX_WORDS) * a.row_stride];
                        if (gw + 1 < W) v |= (uint32_t)base[(size_t)(gw + 1 - PREFIX_WORDS) * a.row_stride] << 16;
                    }
                }
                dst[w / 2] = v;
            }
        };
        auto run = [&]<int... Bs>(std::integer_sequence<int, Bs...>) {
            load_block(std::integral_constant<int, 0>{}, cur);
            (([&] {
                 if constexpr (Bs + 1 < NBLK) load_block(std::integral_constant<int, Bs + 1>{}, nxt);
    #pragma unroll
                 for (int q = 0; q < RATE_W / 2; q++) {
                     if ((q & 1) == 0) s.lo[q / 2] ^= cur[q];
                     else s.hi[q / 2] ^= cur[q];
                 }
                 if constexpr (Bs == NBLK - 1) {
                     constexpr int padbyte = 2 * W - (NBLK - 1) * 136;
                     constexpr uint32_t padv = 0x06u << (8 * (padbyte % 4));
                     if constexpr ((padbyte % 8) < 4) s.lo[padbyte / 8] ^= padv;
                     else s.hi[padbyte / 8] ^= padv;
                     s.hi[16] ^= 0x80000000u;
                 }
                 // keep the look-ahead at ONE block: without these fences the compiler hoists every block's loads to
                 // the top (187 VGPRs, 2 waves/SIMD), which costs the saturated rate at large lane counts
                 asm volatile("" ::: "memory");
                 keccak_f1600_dev(s);
                 asm volatile("" ::: "memory");
                 if constexpr (Bs + 1 < NBLK) {
    #pragma unroll
                     for (int q = 0; q < RATE_W / 2; q++) cur[q] = nxt[q];
                 }
             }()),
             ...);
        };
        run(std::make_integer_sequence<int, NBLK>{});
    } else {
#pragma unroll
        for (int blk = 0; blk < NBLK; blk++) {
#pragma unroll
            for (int w = 0; w < RATE_W; w += 2) {
                const int gw = blk * RATE_W + w; // first of two u16 words forming half a lane
                if (gw >= W) continue;
                uint32_t v = 0;
                if (gw + 1 < PREFIX_WORDS) {
                    v = *reinterpret_cast<const uint32_t *>(a.prefix + dig + 2 * gw);
                } else {
                    v = base[(size_t)(gw - PREFIX_WORDS) * a.row_stride];
                    if (gw + 1 < W) v |= (uint32_t)base[(size_t)(gw + 1 - PREFIX_WORDS) * a.row_stride] << 16;
                }
                if ((w & 2) == 0) s.lo[w / 4] ^= v;
                else s.hi[w / 4] ^= v;
            }
            if (blk == NBLK - 1) {
                constexpr int padbyte = 2 * W - (NBLK - 1) * 136;
                constexpr uint32_t padv = 0x06u << (8 * (padbyte % 4));
                if ((padbyte % 8) < 4) s.lo[padbyte / 8] ^= padv;
                else s.hi[padbyte / 8] ^= padv;
                s.hi[16] ^= 0x80000000u;
            }
            keccak_f1600_dev(s);
        }
    }
    uint4 *o = reinterpret_cast<uint4 *>(a.out + dig);
    o[0] = make_uint4(s.lo[0], s.hi[0], s.lo[1], s.hi[1]);
    o[1] = make_uint4(s.lo[2], s.hi[2], s.lo[3], s.hi[3]);
}

// The same hash with the message rows staged through LDS by LDS-DMA (global_load_lds_dwordx4): one wave-instruction
// brings 8 rows x 128 bytes = the 64 lanes' words of 8 message rows as whole cache lines, with no VGPR destination and
// no per-row address arithmetic on the VALU (the one-lane form issues 220 two-byte gathers and as many 64-bit adds per
// lane).  A 136-byte block of the 64 lanes is [<=72 rows][64 lanes] u16 = 9 KiB; with NBUF = 2 the next block lands
// in the other buffer while Keccak-f runs on this one (launches with few waves per SIMD), with NBUF = 1 occupancy
// hides the landing (many waves per SIMD).  Needs 16-byte aligned row segments and readable padding up to the wave's
// 64th lane (launch_hash_t checks); lanes beyond lanes_per_group take part in the loads and store nothing.
template <int PREFIX_WORDS, int NROWS, int NBUF>
__global__ __launch_bounds__(64) void k_commit_hash_dma(HashArgs a)
{
    constexpr int W = PREFIX_WORDS + NROWS; // u16 words in the message
    constexpr int RATE_W = 68;              // 136-byte rate
    constexpr int NBLK = W / RATE_W + 1;    // pad10*1 always adds to the last (possibly empty) block
    constexpr int STAGE_ROWS = 72;          // 9 DMA pieces of 8 rows
    __shared__ __attribute__((aligned(16))) uint16_t stage[NBUF][STAGE_ROWS * 64];

    // the commitments gate the host's Fiat-Shamir round: when kernels of other pipeline slots share the SIMD, these waves
    // issue first
    __builtin_amdgcn_s_setprio(3);
    const int tl = threadIdx.x;
    // one-dimensional grid, group-major (the dispatcher deals consecutive workgroups evenly over XCDs and SIMDs)
    const int wpg = (a.lanes_per_group + 63) >> 6;
    const int g = (int)blockIdx.x / wpg, bx = (int)blockIdx.x - g * wpg;
    const int lane = bx * 64 + tl;
    const uint16_t *__restrict__ rowbase = a.rows + (size_t)g * a.group_stride + a.col_off + bx * 64; // wave-uniform
    const size_t dig = ((size_t)g * a.out_lanes_per_group + lane) * 32;
    const bool live = lane < a.lanes_per_group;

    // rows [rlo, rhi) of block b sit in stage rows 0 .. rhi - rlo - 1
    auto issue = [&](auto blkc, int buf) {
        constexpr int blk = decltype(blkc)::value;
        constexpr int rlo = blk * RATE_W > PREFIX_WORDS ? blk * RATE_W - PREFIX_WORDS : 0;
        constexpr int rhi = (blk + 1) * RATE_W - PREFIX_WORDS < NROWS ? (blk + 1) * RATE_W - PREFIX_WORDS : NROWS;
        constexpr int cnt = rhi - rlo;
        if constexpr (cnt > 0) {
            constexpr int pieces = (cnt + 7) / 8;
#pragma unroll
            for (int p = 0; p < pieces; p++) {
                int r = rlo + p * 8 + (tl >> 3);
                if (p == pieces - 1 && (cnt & 7)) r = r < rhi ? r : rhi - 1; // the tail piece re-reads the last row
                const uint16_t *src = rowbase + (size_t)r * a.row_stride + (tl & 7) * 8;
                __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void *)(stage[buf] + p * 512), 16, 0, 0);
            }
        }
    };

    KState s;
    kstate_zero(s);
    issue(std::integral_constant<int, 0>{}, 0);
    if constexpr (PREFIX_WORDS > 0) {
        static_assert(PREFIX_WORDS == 16, "the prefix is one 32-byte digest");
        if (live) {
            const uint4 *pp = reinterpret_cast<const uint4 *>(a.prefix + dig);
            const uint4 p0 = pp[0], p1 = pp[1];
            s.lo[0] = p0.x; s.hi[0] = p0.y; s.lo[1] = p0.z; s.hi[1] = p0.w;
            s.lo[2] = p1.x; s.hi[2] = p1.y; s.lo[3] = p1.z; s.hi[3] = p1.w;
        }
    }
    auto run = [&]<int... Bs>(std::integer_sequence<int, Bs...>) {
        (([&] {
             constexpr int blk = Bs;
             constexpr int buf = NBUF == 2 ? (blk & 1) : 0;
             constexpr int rlo = blk * RATE_W > PREFIX_WORDS ? blk * RATE_W - PREFIX_WORDS : 0;
             if constexpr (NBUF == 2 && blk + 1 < NBLK) {
                 // the other buffer was consumed one block ago: start the next block's DMA before waiting for this one
                 issue(std::integral_constant<int, blk + 1>{}, buf ^ 1);
                 constexpr int nrlo = (blk + 1) * RATE_W - PREFIX_WORDS;
                 constexpr int nrhi = (blk + 2) * RATE_W - PREFIX_WORDS < NROWS ? (blk + 2) * RATE_W - PREFIX_WORDS : NROWS;
                 constexpr int npieces = nrhi > nrlo ? (nrhi - nrlo + 7) / 8 : 0;
                 __builtin_amdgcn_s_waitcnt(0x0F70 | (npieces & 15) | ((npieces >> 4) << 14)); // vmcnt(npieces): this block has landed
             } else {
                 __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0)
             }
             asm volatile("" ::: "memory");
             const uint16_t *st = stage[buf];
#pragma unroll
             for (int w = 0; w < RATE_W; w += 2) {
                 const int gw = blk * RATE_W + w; // first of two u16 words forming half a 64-bit lane
                 if (gw < PREFIX_WORDS || gw >= W) continue;
                 uint32_t v = st[(gw - PREFIX_WORDS - rlo) * 64 + tl];
                 if (gw + 1 < W) v |= (uint32_t)st[(gw + 1 - PREFIX_WORDS - rlo) * 64 + tl] << 16;
                 if ((w & 2) == 0) s.lo[w / 4] ^= v;
                 else s.hi[w / 4] ^= v;
             }
             if constexpr (blk == NBLK - 1) {
                 constexpr int padbyte = 2 * W - (NBLK - 1) * 136;
                 constexpr uint32_t padv = 0x06u << (8 * (padbyte % 4));
                 if constexpr ((padbyte % 8) < 4) s.lo[padbyte / 8] ^= padv;
                 else s.hi[padbyte / 8] ^= padv;
                 s.hi[16] ^= 0x80000000u;
             }
             if constexpr (NBUF == 1 && blk + 1 < NBLK) {
                 // single buffer: every LDS read of this block must have returned before the next DMA overwrites it
                 asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                 issue(std::integral_constant<int, blk + 1>{}, 0);
             }
             asm volatile("" ::: "memory");
             keccak_f1600_dev(s);
         }()),
         ...);
    };
    run(std::make_integer_sequence<int, NBLK>{});
    // The wave's 64 digests are 2 KiB of consecutive table bytes.  Through LDS (the staging buffer is free now) so that every
    // store instruction writes 1 KiB of CONSECUTIVE bytes, 16 per lane -- whole lines for HBM (lane l's own 32 bytes would be two
    // half-dense instructions)
    uint4 *sd = reinterpret_cast<uint4 *>(&stage[0][0]);
    sd[2 * tl] = make_uint4(s.lo[0], s.hi[0], s.lo[1], s.hi[1]);
    sd[2 * tl + 1] = make_uint4(s.lo[2], s.hi[2], s.lo[3], s.hi[3]);
    __syncthreads(); // one wave per workgroup: orders the LDS writes above against the reads below
    const size_t base16 = ((size_t)g * a.out_lanes_per_group + (size_t)bx * 64) * 2; // 16-byte units
    uint4 *o = reinterpret_cast<uint4 *>(a.out) + base16;
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const int c = j * 64 + tl; // chunk c = half (c & 1) of the digest of lane c >> 1
        if (bx * 64 + (c >> 1) < a.lanes_per_group) {
            const uint4 v = sd[c];
            o[c] = v;
        }
    }
    (void)live;
    (void)dig;
}

// SHA3-256 / SHAKE256 of n byte messages of equal length stored message-major.
__global__ __launch_bounds__(64) void k_sha3_msgs(const uint8_t *__restrict__ in, size_t in_stride, int len,
                                                  uint8_t *__restrict__ out, size_t out_stride, int outlen,
                                                  int n, int domain)
{
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    const uint8_t *m = in + (size_t)i * in_stride;
    uint64_t s[25];
#pragma unroll
    for (int k = 0; k < 25; k++) s[k] = 0;
    int pos = 0;
    for (int b = 0; b < len; b++) {
        const uint64_t v = (uint64_t)m[b] << (8 * (pos & 7));
        const int w = pos >> 3;
#pragma unroll
        for (int k = 0; k < 17; k++)
            if (k == w) s[k] ^= v;
        if (++pos == 136) {
            keccak_f1600(s);
            pos = 0;
        }
    }
    {
        const uint64_t v = (uint64_t)domain << (8 * (pos & 7));
        const int w = pos >> 3;
#pragma unroll
        for (int k = 0; k < 17; k++)
            if (k == w) s[k] ^= v;
        s[16] ^= 0x8000000000000000ULL;
    }
    uint8_t *o = out + (size_t)i * out_stride;
    int produced = 0;
    while (produced < outlen) {
        keccak_f1600(s);
        const int take = min(136, outlen - produced);
        for (int b = 0; b < take; b++) {
            uint64_t lanev = 0;
            const int w = b >> 3;
#pragma unroll
            for (int k = 0; k < 17; k++)
                if (k == w) lanev = s[k];
            o[produced + b] = (uint8_t)(lanev >> (8 * (b & 7)));
        }
        produced += take;
    }
}

// The "warp-cooperative" sponge of BASELINE.json's north_star in its cheapest form (kosk_keccak_split_dev.hpp): one state spread
// over a LANE PAIR -- the even lane holds the low halves of the 25 words, the odd lane the high halves, the 64-bit rotations
// exchange halves by DPP.  32 messages per wave.  SHA3-256 / SHAKE of message-major input like k_sha3_msgs; kernel-level entry
// kosk_sha3_256_batch_pair.  The pipeline keeps the one-lane layout (this one has 1.28x less single-wave latency but fewer
// states per issue slot: DESIGN.md 8).  All 64 lanes stay active: the DPP exchange reads the partner's registers.
__global__ __launch_bounds__(64) void k_sha3_msgs_pair(const uint8_t *__restrict__ in, size_t in_stride, int len,
                                                       uint8_t *__restrict__ out, size_t out_stride, int outlen, int n, int domain)
{
    const int msg = blockIdx.x * 32 + (threadIdx.x >> 1);
    const bool hi = threadIdx.x & 1;
    const bool live = msg < n;
    const uint8_t *m = in + (size_t)(live ? msg : n - 1) * in_stride; // idle pairs hash the last message and store nothing
    KHalf s;
#pragma unroll
    for (int k = 0; k < 25; k++) s.w[k] = 0;
    // byte `pos` of the 136-byte rate block belongs to word pos >> 3, half (pos >> 2) & 1, bits 8 (pos & 3)
    auto absorb_byte = [&](int pos, uint32_t byte) {
        if ((((pos >> 2) & 1) != 0) != hi) return;
        const uint32_t v = byte << (8 * (pos & 3));
        const int w = pos >> 3;
#pragma unroll
        for (int k = 0; k < 17; k++)
            if (k == w) s.w[k] ^= v;
    };
    int pos = 0;
    for (int b = 0; b < len; b++) {
        absorb_byte(pos, m[b]);
        if (++pos == 136) {
            keccak_f1600_split(s, hi);
            pos = 0;
        }
    }
    absorb_byte(pos, (uint32_t)domain);
    if (hi) s.w[16] ^= 0x80000000u;
    uint8_t *o = out + (size_t)(live ? msg : 0) * out_stride;
    int produced = 0;
    while (produced < outlen) {
        keccak_f1600_split(s, hi);
        const int take = min(136, outlen - produced);
        for (int b = 0; b < take; b++) {
            if ((((b >> 2) & 1) != 0) != hi) continue;
            uint32_t v = 0;
            const int w = b >> 3;
#pragma unroll
            for (int k = 0; k < 17; k++)
                if (k == w) v = s.w[k];
            if (live) o[produced + b] = (uint8_t)(v >> (8 * (b & 3)));
        }
        produced += take;
    }
}

// =========================================================================
// preprocessing expansions from the randomness tape
// =========================================================================

// The three tape / witness expansions that open the prover, in ONE launch of 64-thread blocks (role by block range):
//  A  f_i = BE16(SHAKE256(seed_i || i)[2j..2j+1]) % q                          mlwe_prover.cpp:8-14
//  B  shares of parties 0..150 of every fresh sharing = BE16(tape) % q         ss.cpp:5-11
//  C  packed secrets that depend only on the witness: s, e, the range constants and the multiplication-gate
//     chain prod_{m<=k+1} (s - eta_m) (what the reference obtains through recon_secrets_2ddeg, :351-373)
struct PreArgs {
    const uint8_t *tape;
    size_t tape_stride;
    TapeSegs segs; // count > 0: proof b's tape is segs.ptr[b / segs.per] + (b % segs.per) * tape_stride
    uint16_t *P;
    size_t proof_stride;
    int row_f, M, nproofs;
    int slice0_off, slice_begin, slice_end; // fresh sharings [slice_begin, slice_end) of the tape order are drawn
    int witness_mode;                       // 0 none, 1 everything, 2 only the range constants (prepare_range_proof)
    const int16_t *fresh_rows;
    const int16_t *se;
    size_t se_stride;
    RowMap rm;
    int eta1;
    int nbA, nbB; // blocks of role A, B (role C: 4 per proof)
    // key generation as two more roles of the same launch (kosk.cpp:12-20): G one thread per matrix entry A[i][j], N one per
    // noise polynomial; each hashes d || K itself (one extra permutation) so that no role waits for another
    int nbG, nbN, K;
    uint8_t *kg_seeds;   // public seed || noise seed of proof b at kg_seeds + b * kg_seed_stride (written by the nonce-0 thread of role N)
    size_t kg_seed_stride;
    int16_t *kg_A;       // [proof][K][K][256]
    size_t kg_A_stride;
    int16_t *kg_se;      // [proof][2K][256] s then e
    size_t kg_se_stride;
    XofGuard xof;
};
constexpr int PRE_SLICES = 8; // fresh sharings per role-B block
__device__ __forceinline__ const uint8_t *pre_tape(const PreArgs &a, int b)
{
    if (a.segs.count == 0) return a.tape + (size_t)b * a.tape_stride;
    const int j = b / a.segs.per;
    // static indices only: a run-time index into the by-value argument block would make the compiler copy all of it to scratch
    const uint8_t *base = a.segs.ptr[0];
#pragma unroll
    for (int k = 1; k < 16; k++) base = j == k ? a.segs.ptr[k] : base;
    return base + (size_t)(b - j * a.segs.per) * a.tape_stride;
}

__device__ __forceinline__ void pre_tape_randoms(const PreArgs &a, int idx, int lane)
{
    const int ngrp = (a.slice_end - a.slice_begin + PRE_SLICES - 1) / PRE_SLICES;
    const int part = idx % 3, grp = (idx / 3) % ngrp, b = idx / (3 * ngrp);
    const int t = part * 64 + lane;
    if (t > NOPEN) return;
#pragma unroll
    for (int q = 0; q < PRE_SLICES; q++) {
        const int slice = a.slice_begin + grp * PRE_SLICES + q;
        if (slice >= a.slice_end) break;
        const uint8_t *src = pre_tape(a, b) + a.slice0_off + 302 * slice + 2 * t;
        const uint32_t v = (((uint32_t)src[0] << 8) | src[1]) % (uint32_t)Q;
        a.P[(size_t)b * a.proof_stride + (size_t)a.fresh_rows[slice] * RS + NSEC + t] = (uint16_t)v;
    }
}

__device__ __forceinline__ void pre_witness_secrets(const PreArgs &a, int idx, int lane)
{
    const int j = (idx & 3) * 64 + lane, b = idx >> 2;
    const RowMap &rm = a.rm;
    uint16_t *Pb = a.P + (size_t)b * a.proof_stride;
    const int16_t *sb = a.se + (size_t)b * a.se_stride;
    for (int who = 0; who < 2; who++) {
        for (int i = 0; i < rm.K; i++) {
            if (a.witness_mode == 2) { // mlwe_prover.cpp:41-59: the constants -eta1..eta1, replicated over the 256 positions
                for (int m = 0; m < rm.E; m++)
                    Pb[(size_t)((who ? rm.eeta : rm.seta) + i * rm.E + m) * RS + j] = (uint16_t)gf_encode(m - a.eta1);
                continue;
            }
            const uint32_t v = gf_encode(sb[(who * rm.K + i) * 256 + j]);
            Pb[(size_t)((who ? rm.e : rm.s) + i) * RS + j] = (uint16_t)v;
            uint32_t z = 0;
            for (int m = 0; m < rm.E; m++) {
                const uint32_t c = gf_encode(m - a.eta1);
                Pb[(size_t)((who ? rm.eeta : rm.seta) + i * rm.E + m) * RS + j] = (uint16_t)c;
                const uint32_t x = gf_sub(v, c);
                z = (m == 0) ? x : gf_mul(z, x);
                if (m >= 1) Pb[(size_t)(who ? rm.ze(i, m - 1) : rm.zs(i, m - 1)) * RS + j] = (uint16_t)z;
            }
        }
    }
}

// ---- roles A, G, N on the lane-pair sponge: pair pr = thread >> 1 of the role's range, hi = the lane that holds the high halves ----
__device__ __forceinline__ void pre_expand_f_pair(const PreArgs &a, int pr, bool hi)
{
    const int n = a.M * a.nproofs;
    const bool live = pr < n;
    const int t = live ? pr : n - 1;
    const int b = t / a.M, i = t % a.M;
    const uint32_t *seed = reinterpret_cast<const uint32_t *>(pre_tape(a, b) + 64 + 32 * i) + (hi ? 1 : 0); // my half of each of the four seed words
    KHalf s;
#pragma unroll
    for (int k = 0; k < 25; k++) s.w[k] = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) s.w[k] = seed[2 * k];
    if (!hi) s.w[4] = (uint32_t)(uint8_t)i | (0x1Fu << 8);
    if (hi) s.w[16] = 0x80000000u;
    uint16_t *dst = a.P + (size_t)b * a.proof_stride + (size_t)(a.row_f + i) * RS + (hi ? 2 : 0);
#pragma unroll 1
    for (int blk = 0; blk < 4; blk++) {
        keccak_f1600_split(s, hi);
        if (live) {
#pragma unroll
            for (int w = 0; w < 17; w++) { // my half of word w = elements 4 w (+ 2), 4 w + 1 (+ 2) of the block's 68 (52 in the last)
                if (blk == 3 && w >= 13) break;
                const uint32_t v = s.w[w];
                const uint32_t h0 = v & 0xFFFFu, h1 = v >> 16;
                const uint32_t be0 = ((h0 & 0xFF) << 8) | (h0 >> 8), be1 = ((h1 & 0xFF) << 8) | (h1 >> 8);
                *reinterpret_cast<uint32_t *>(dst + blk * 68 + w * 4) = (be0 % (uint32_t)Q) | ((be1 % (uint32_t)Q) << 16);
            }
        }
    }
}
// role G on the WAVE sponge (kosk_keygen_wave_dev.hpp): block t of the role = matrix entry t, the whole wave on its chain
// (sha3_512(d || K), then SHAKE128(rho || j || i) block by block with a one-step rejection parse)
__device__ __forceinline__ void pre_gen_matrix_wave(const PreArgs &a, int t, uint32_t *st, uint8_t *sq)
{
    const int KK = a.K * a.K;
    const int b = t / KK, ij = t - b * KK, i = ij / a.K, j = ij - i * a.K;
    __builtin_amdgcn_s_setprio(3);
    kw_gen_matrix(pre_tape(a, b), true, a.K, i, j, a.kg_A + (size_t)b * a.kg_A_stride + (size_t)ij * 256, a.xof, st, sq);
}
__device__ __forceinline__ void pre_noise_pair(const PreArgs &a, int pr, bool hi)
{
    const int n = a.nproofs * 2 * a.K;
    const bool live = pr < n;
    const int t = live ? pr : n - 1;
    const int b = t / (2 * a.K), nonce = t - b * 2 * a.K;
    uint32_t pub[8], noise[8];
    kp_seed_hash(pre_tape(a, b), a.K, hi, pub, noise);
    if (nonce == 0 && live && !hi) {
        uint32_t *o = reinterpret_cast<uint32_t *>(a.kg_seeds + (size_t)b * a.kg_seed_stride);
#pragma unroll
        for (int q = 0; q < 8; q++) { o[q] = pub[q]; o[8 + q] = noise[q]; }
    }
    kp_noise(noise, nonce, a.eta1, hi, live, a.kg_se + (size_t)b * a.kg_se_stride + (size_t)nonce * 256);
}

__global__ __launch_bounds__(64) void k_prover_pre(PreArgs a)
{
    // Workgroups are dispatched in index order, so the roles come chains first: G (a matrix entry: four or more dependent Keccak
    // permutations; a wave per entry since round 6 -- on one lane pair the chain was ~60 us and set the launch's time), N, A (four
    // permutations per lane pair) start at once and the thousands of short role-B workgroups stream through beside them.  With B in
    // front of G (rounds 1-3) the launch took B's streaming time PLUS G's chain: 100 us at 138 proofs, 73 us at 46.
    int blk = blockIdx.x;
    // role G: one matrix entry per block on the wave sponge (round 6); roles N, A on the lane-pair sponge (kosk_keygen_dev.hpp: kp_*): 32 sponges per 64-thread block
    __shared__ __attribute__((aligned(16))) uint32_t kw_st[KW_ST_WORDS];
    __shared__ __attribute__((aligned(16))) uint8_t kw_sq[KW_SQ_BYTES];
    const bool hi = threadIdx.x & 1;
    if (blk < a.nbG) return pre_gen_matrix_wave(a, blk, kw_st, kw_sq);
    blk -= a.nbG;
    if (blk < a.nbN) return pre_noise_pair(a, blk * 32 + ((int)threadIdx.x >> 1), hi);
    blk -= a.nbN;
    if (blk < a.nbA) return pre_expand_f_pair(a, blk * 32 + ((int)threadIdx.x >> 1), hi);
    blk -= a.nbA;
    if (blk < a.nbB) return pre_tape_randoms(a, blk, threadIdx.x);
    pre_witness_secrets(a, blk - a.nbB, threadIdx.x);
}

// =========================================================================
// K5  NTT-256: 16 lanes x 16 coefficients per polynomial, 4 polynomials per
// wave, 16 per workgroup.  Layers len=128..16 run on the stride-16 register
// layout, one LDS transpose, layers len=8..2 on the contiguous layout, then
// Barrett.  Global traffic is 16-byte coalesced both ways.
// =========================================================================
__constant__ static const ZetaTable kZetasDev = ZetaTable();
__constant__ static const ZetaTableDot kZetasDot = ZetaTableDot();

// Cooley-Tukey butterfly (ntt.c:86-91) in four full-rate instructions.  A coefficient lives in the low half of its register.
//   hi.h16 = lo16(hi.l16 * zq)                 m = (int16)(a z QINV), the Montgomery factor (reduce.c:19), into the spare half
//   t      = hi.l16 * z + hi.h16 * (-q)        one dot product of packed int16 pairs: a z - m q (reduce.c:20 before the shift)
//   hi     = lo.l16 - t.h16 ;  lo = lo.l16 + t.h16     fqmul(z, a) is t's high half: the SDWA operand selects sign-extend it
// The three multiplies of the plain form (a z, (a z) QINV, m q) are half-rate 24/32-bit multiplies on gfx950; these four are not.
__device__ __forceinline__ void ntt_bfly(int32_t &lo, int32_t &hi, int32_t zq, int32_t zz)
{
    int32_t t, nlo, nhi;
    asm("v_mad_u16 %0, %0, %1, 0 op_sel:[0,0,0,1]" : "+v"(hi) : "v"(zq));
    asm("v_dot2_i32_i16 %0, %1, %2, 0" : "=v"(t) : "v"(hi), "v"(zz));
    asm("v_sub_u32_sdwa %0, sext(%1), sext(%2) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1" : "=v"(nhi) : "v"(lo), "v"(t));
    asm("v_add_u32_sdwa %0, sext(%1), sext(%2) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1" : "=v"(nlo) : "v"(lo), "v"(t));
    lo = nlo;
    hi = nhi;
}
// the same with the zeta operands in scalar registers (layers whose zeta is the same for every lane)
__device__ __forceinline__ void ntt_bfly_s(int32_t &lo, int32_t &hi, int32_t zq, int32_t zz)
{
    int32_t t, nlo, nhi;
    asm("v_mad_u16 %0, %0, %1, 0 op_sel:[0,0,0,1]" : "+v"(hi) : "s"(zq));
    asm("v_dot2_i32_i16 %0, %1, %2, 0" : "=v"(t) : "v"(hi), "s"(zz));
    asm("v_sub_u32_sdwa %0, sext(%1), sext(%2) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1" : "=v"(nhi) : "v"(lo), "v"(t));
    asm("v_add_u32_sdwa %0, sext(%1), sext(%2) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1" : "=v"(nlo) : "v"(lo), "v"(t));
    lo = nlo;
    hi = nhi;
}

constexpr int NTT_PPB = 16;
constexpr int NTT_LSTRIDE = 256 + 16; // int16 per polynomial in LDS (32-byte pad: conflict-free stride reads)

#define KOSK_BFLY(lo, hi, z)                 \
    {                                        \
        const int32_t t_ = fqmul((z), (hi)); \
        (hi) = (lo) - t_;                    \
        (lo) = (lo) + t_;                    \
    }

// Workgroup barrier that orders LDS traffic only.  __syncthreads() is a fence as well: hipcc waits for EVERY outstanding memory
// operation (s_waitcnt vmcnt(0)) in front of it, which would make the workgroup wait for the next tile's prefetched global loads
// and for the previous tile's stores at each of the transform's barriers.
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// polynomial p -> (group g, index i inside the group): p = g npg + i.  A 32-bit division costs ~20 vector instructions and the
// kernel needs three per thread; with floor(2^32 / npg) from the host it is a multiply-high and one correction step.
__device__ __forceinline__ void ntt_split(const NttArgs &a, int p, int &g, int &i)
{
    uint32_t gg = __umulhi((uint32_t)p, a.npg_magic);
    int r = p - (int)gg * a.npg;
    if (r >= a.npg) { gg++; r -= a.npg; }
    g = (int)gg;
    i = r;
}

// A tile = up to 16 polynomials p0 .. p0+15, one 256-thread workgroup.  Its input as two 16-byte pieces per thread: loaded into
// registers (ntt256_fetch), later written to the LDS staging image (ntt256_stage) -- apart, so that a workgroup that walks
// several tiles has the NEXT tile's loads in flight while it transforms the current one.
// PLAIN: the polynomials lie back to back in both buffers (no groups, no offset tables, no compare mode): polynomial p at p * 256
template <bool PLAIN>
__device__ __forceinline__ void ntt256_fetch(const NttArgs &a, const int p0, uint4 (&x)[2])
{
    if constexpr (PLAIN) {
        const uint4 *src = reinterpret_cast<const uint4 *>(a.in) + (size_t)p0 * 32 + threadIdx.x;
        const int left = (a.npoly - p0) * 32; // 16-byte pieces of this tile that exist
#pragma unroll
        for (int q = 0; q < 2; q++) {
            x[q] = make_uint4(0, 0, 0, 0);
            if ((int)threadIdx.x + q * 256 < left) x[q] = src[q * 256];
        }
        return;
    }
    int g0, i0;
    ntt_split(a, p0 + ((int)threadIdx.x >> 5), g0, i0);
#pragma unroll
    for (int q = 0; q < 2; q++) {
        const int c = (int)threadIdx.x + q * 256;
        const int pl = c >> 5, ch = c & 31, p = p0 + pl;
        x[q] = make_uint4(0, 0, 0, 0);
        if (p < a.npoly) {
            int g = g0, i = i0;
            if (q) { // the thread's second polynomial is 8 further on
                i += 8;
                if (a.npg >= 8) { if (i >= a.npg) { i -= a.npg; g++; } }
                else ntt_split(a, p, g, i);
            }
            const size_t off = (size_t)g * a.in_gstride + (a.src_off ? (size_t)a.src_off[i] : (size_t)i * 256);
            x[q] = *reinterpret_cast<const uint4 *>(a.in + off + ch * 8);
        }
    }
}
__device__ __forceinline__ void ntt256_stage(const uint4 (&x)[2], int16_t *__restrict__ lds)
{
#pragma unroll
    for (int q = 0; q < 2; q++) {
        const int c = (int)threadIdx.x + q * 256;
        *reinterpret_cast<uint4 *>(lds + (c >> 5) * NTT_LSTRIDE + (c & 31) * 8) = x[q];
    }
}
template <bool PLAIN>
__device__ __forceinline__ void ntt256_transform(const NttArgs &a, const int p0, int16_t *__restrict__ lds);

// one tile by a 256-thread workgroup (all threads must call it)
__device__ __forceinline__ void ntt256_tile(const NttArgs &a, const int p0, int16_t *__restrict__ lds)
{
    uint4 x[2];
    ntt256_fetch<false>(a, p0, x);
    ntt256_stage(x, lds);
    __syncthreads();
    ntt256_transform<false>(a, p0, lds);
}

// the staged tile (every thread's ntt256_stage done, barrier passed) -> transformed polynomials in global memory
template <bool PLAIN>
__device__ __forceinline__ void ntt256_transform(const NttArgs &a, const int p0, int16_t *__restrict__ lds)
{
    const int tid = threadIdx.x;
    const int pl = tid >> 4, l = tid & 15;
    int16_t *mine = lds + pl * NTT_LSTRIDE;
    int32_t r[16]; // a coefficient is the LOW half of its register; the high half is scratch (ntt_bfly)
#pragma unroll
    for (int i = 0; i < 16; i++) r[i] = mine[l + 16 * i];

#define KOSK_BFLYZ(lo, hi, k) ntt_bfly_s((lo), (hi), (int32_t)kZetasDot.e[(k)].zq, (int32_t)kZetasDot.e[(k)].zz)
    // coefficient index j = l + 16 i : len = 128, 64, 32, 16 <-> register distance 8, 4, 2, 1
#pragma unroll
    for (int i = 0; i < 8; i++) KOSK_BFLYZ(r[i], r[i + 8], 1);
#pragma unroll
    for (int blk = 0; blk < 2; blk++)
#pragma unroll
        for (int i = 0; i < 4; i++) KOSK_BFLYZ(r[8 * blk + i], r[8 * blk + i + 4], 2 + blk);
#pragma unroll
    for (int blk = 0; blk < 4; blk++)
#pragma unroll
        for (int i = 0; i < 2; i++) KOSK_BFLYZ(r[4 * blk + i], r[4 * blk + i + 2], 4 + blk);
#pragma unroll
    for (int blk = 0; blk < 8; blk++) KOSK_BFLYZ(r[2 * blk], r[2 * blk + 1], 8 + blk);

    lds_barrier();
#pragma unroll
    for (int i = 0; i < 16; i++) mine[l + 16 * i] = (int16_t)r[i];
    lds_barrier();
    {
        const uint4 v0 = *reinterpret_cast<const uint4 *>(mine + 16 * l);
        const uint4 v1 = *reinterpret_cast<const uint4 *>(mine + 16 * l + 8);
        const uint32_t w[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
        for (int q = 0; q < 8; q++) {
            r[2 * q] = (int32_t)w[q];         // low half = coefficient 2q (the high half is ignored)
            r[2 * q + 1] = (int32_t)(w[q] >> 16);
        }
    }
    // coefficient index j = 16 l + c : zeta index = 128/len + j/(2 len)  (per lane from here on: vector operands)
#undef KOSK_BFLYZ
#define KOSK_BFLYZ(lo, hi, zq_, zz_) ntt_bfly((lo), (hi), (zq_), (zz_))
    {
        const ZetaTableDot::Pair e8 = kZetasDot.e[16 + l];
        const int32_t q8 = (int32_t)e8.zq, z8 = (int32_t)e8.zz;
#pragma unroll
        for (int c = 0; c < 8; c++) KOSK_BFLYZ(r[c], r[c + 8], q8, z8);
        const ZetaTableDot::Pair e4a = kZetasDot.e[32 + 2 * l], e4b = kZetasDot.e[33 + 2 * l]; // neighbours: one 16-byte load
        const int32_t q4a = (int32_t)e4a.zq, z4a = (int32_t)e4a.zz, q4b = (int32_t)e4b.zq, z4b = (int32_t)e4b.zz;
#pragma unroll
        for (int c = 0; c < 4; c++) {
            KOSK_BFLYZ(r[c], r[c + 4], q4a, z4a);
            KOSK_BFLYZ(r[8 + c], r[12 + c], q4b, z4b);
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const ZetaTableDot::Pair e2 = kZetasDot.e[64 + 4 * l + q];
            const int32_t q2 = (int32_t)e2.zq, z2 = (int32_t)e2.zz;
            KOSK_BFLYZ(r[4 * q], r[4 * q + 2], q2, z2);
            KOSK_BFLYZ(r[4 * q + 1], r[4 * q + 3], q2, z2);
        }
    }
#undef KOSK_BFLYZ
    const int p = p0 + pl;
    if (p < a.npoly) {
        // poly_reduce (poly.c:261-265) / encode_to_gf3329: x -> the representative the consumer wants.  One Montgomery step with
        // 2^16 mod q (the butterfly's first two instructions) leaves x mod q in (-q, q) in the high half of a register; pairs of
        // those are packed and finished with packed 16-bit arithmetic -- 5 to 6 instructions per coefficient where Barrett
        // (two 24-bit multiplies) + select + pack took 9.
        constexpr int32_t CQ = (int32_t)(((uint32_t)(2285 * QINV)) & 0xFFFFu);                       // (2^16 mod q) q^-1 mod 2^16
        constexpr int32_t CZ = (int32_t)(2285u | ((uint32_t)(uint16_t)(-Q) << 16));
        constexpr int32_t S15 = 0x000F000F, Q2 = (int32_t)((uint32_t)Q | ((uint32_t)Q << 16)), H2 = (int32_t)((uint32_t)(Q / 2) | ((uint32_t)(Q / 2) << 16));
        uint32_t w[8];
#pragma unroll
        for (int q = 0; q < 8; q++) {
            int32_t t0, t1;
            asm("v_mad_u16 %0, %0, %1, 0 op_sel:[0,0,0,1]" : "+v"(r[2 * q]) : "s"(CQ));
            asm("v_dot2_i32_i16 %0, %1, %2, 0" : "=v"(t0) : "v"(r[2 * q]), "s"(CZ));
            asm("v_mad_u16 %0, %0, %1, 0 op_sel:[0,0,0,1]" : "+v"(r[2 * q + 1]) : "s"(CQ));
            asm("v_dot2_i32_i16 %0, %1, %2, 0" : "=v"(t1) : "v"(r[2 * q + 1]), "s"(CZ));
            uint32_t c = __builtin_amdgcn_perm((uint32_t)t1, (uint32_t)t0, 0x07060302u), m; // {t0.h16, t1.h16}: both in (-q, q)
            // c += (c >> 15) & q : [0, q), encode_to_gf3329 -- packed 16-bit, constants in scalar registers
            asm("v_pk_ashrrev_i16 %0, %1, %2" : "=v"(m) : "s"(S15), "v"(c));
            asm("v_and_b32 %0, %1, %0" : "+v"(m) : "s"(Q2));
            asm("v_pk_add_u16 %0, %0, %1" : "+v"(c) : "v"(m));
            if (!a.out_canonical) { // c -= ((q/2 - c) >> 15) & q : the centred representative [-(q-1)/2, (q-1)/2] of poly_reduce
                asm("v_pk_sub_i16 %0, %1, %2" : "=v"(m) : "s"(H2), "v"(c));
                asm("v_pk_ashrrev_i16 %0, %1, %0" : "+v"(m) : "s"(S15));
                asm("v_and_b32 %0, %1, %0" : "+v"(m) : "s"(Q2));
                asm("v_pk_sub_i16 %0, %0, %1" : "+v"(c) : "v"(m));
            }
            w[q] = c;
        }
        if constexpr (PLAIN) {
            uint4 *o = reinterpret_cast<uint4 *>(a.out) + (size_t)p * 32 + 2 * l;
            o[0] = make_uint4(w[0], w[1], w[2], w[3]);
            o[1] = make_uint4(w[4], w[5], w[6], w[7]);
            return;
        }
        int g, i;
        ntt_split(a, p, g, i);
        const size_t off = (size_t)g * a.out_gstride + (a.dst_off ? (size_t)a.dst_off[i] : (size_t)i * 256);
        uint4 *o = reinterpret_cast<uint4 *>(a.out + off + 16 * l);
        if (a.cmp_fail) {
            const uint4 c0 = o[a.cmp_delta / 8], c1 = o[a.cmp_delta / 8 + 1];
            if (c0.x != w[0] || c0.y != w[1] || c0.z != w[2] || c0.w != w[3] || c1.x != w[4] || c1.y != w[5] || c1.z != w[6] || c1.w != w[7])
                atomicOr(&a.cmp_fail[g], 1u << a.cmp_bit);
        } else {
            o[0] = make_uint4(w[0], w[1], w[2], w[3]);
            o[1] = make_uint4(w[4], w[5], w[6], w[7]);
        }
    }
}

// One tile per workgroup.  (Measured in round 4, tools/ntt_time.py: a workgroup that walks several tiles with the next tile's
// loads in flight during the transform is no faster -- 22.5 against 21.9 us at 65 536 polynomials -- because the kernel is bound by
// instruction issue, not by exposed memory latency: without its butterflies it takes 14.4 us, a plain copy of the same bytes 10.4 us.)
template <bool PLAIN>
__global__ __launch_bounds__(256) void k_ntt256(NttArgs a)
{
    __shared__ __attribute__((aligned(16))) int16_t lds[NTT_PPB * NTT_LSTRIDE];
    uint4 x[2];
    ntt256_fetch<PLAIN>(a, blockIdx.x * NTT_PPB, x);
    ntt256_stage(x, lds);
    lds_barrier();
    ntt256_transform<PLAIN>(a, blockIdx.x * NTT_PPB, lds);
}

// K6  r_i = tomont(Barrett(sum_l basemul(A[i][l], v[l])))      polyvec.c:202-214, poly.c:307-313
// one thread per degree-1 factor (pair of coefficients); output canonical u16
__device__ __forceinline__ void matvec_ntt_item(const int16_t *__restrict__ A, size_t A_stride, uint16_t *P, size_t proof_stride,
                                                int v_row0, int row0, int K, const int t, const int i, const int b)
{
    const int32_t zeta = (t & 1) ? -(int32_t)kZetasDev.z[64 + (t >> 1)] : (int32_t)kZetasDev.z[64 + (t >> 1)];
    const int16_t *Ai = A + (size_t)b * A_stride + (size_t)i * K * 256;
    // the vector operand is the canonical NTT image stored in the packed-secret part of K consecutive rows
    // (any representative < q gives the same residues, and the output is canonicalised)
    const uint16_t *vb = P + (size_t)b * proof_stride + (size_t)v_row0 * RS;
    constexpr int VS = RS; // u16 between the K polynomials
    int32_t r0 = 0, r1 = 0;
    for (int l = 0; l < K; l++) {
        const int32_t a0 = Ai[l * 256 + 2 * t], a1 = Ai[l * 256 + 2 * t + 1];
        const int32_t b0 = vb[l * VS + 2 * t], b1 = vb[l * VS + 2 * t + 1];
        // ntt.c:139-146 with int16 wrap-around of the reference's accumulation
        int32_t x0 = fqmul(fqmul(a1, b1), zeta);
        x0 = (int16_t)(x0 + fqmul(a0, b0));
        int32_t x1 = fqmul(a0, b1);
        x1 = (int16_t)(x1 + fqmul(a1, b0));
        r0 = (int16_t)(r0 + x0);
        r1 = (int16_t)(r1 + x1);
    }
    constexpr int32_t f = (int32_t)((1ULL << 32) % Q);
    r0 = montgomery_reduce(barrett_reduce(r0) * f);
    r1 = montgomery_reduce(barrett_reduce(r1) * f);
    uint16_t *dst = P + (size_t)b * proof_stride + (size_t)(row0 + i) * RS;
    *reinterpret_cast<uint32_t *>(dst + 2 * t) = gf_from_i32(r0) | (gf_from_i32(r1) << 16);
}

__global__ __launch_bounds__(128) void k_matvec_ntt(const int16_t *__restrict__ A, size_t A_stride,
                                                    uint16_t *__restrict__ P, size_t proof_stride, int v_row0, int row0, int K)
{
    matvec_ntt_item(A, A_stride, P, proof_stride, v_row0, row0, K, threadIdx.x, blockIdx.x, blockIdx.y);
}

// NTT(s + r), NTT(e + r), A o NTT(s + r) and the tails of their re-sharing inputs for ONE proof per workgroup: the three
// launches k_ntt256 -> k_matvec_ntt -> k_copy_tails (each 4-5 us of mostly launch latency) in one.   mlwe_prover.cpp:260-288
__global__ __launch_bounds__(256) void k_relation_ntt(NttArgs na, const int16_t *__restrict__ A, size_t A_stride, uint16_t *P,
                                                      size_t proof_stride, RowMap rm)
{
    __shared__ __attribute__((aligned(16))) int16_t lds[NTT_PPB * NTT_LSTRIDE];
    const int b = blockIdx.x, K = rm.K, tid = threadIdx.x;
    NttArgs a = na; // this proof's 2K polynomials as a group of their own
    a.in += (size_t)b * na.in_gstride;
    a.out += (size_t)b * na.out_gstride;
    a.npoly = a.npg;
    ntt256_tile(a, 0, lds);
    __syncthreads(); // the NTT images (global, written by this workgroup) are the matvec's vector operand
    for (int idx = tid; idx < K * 128; idx += 256) matvec_ntt_item(A, A_stride, P, proof_stride, rm.nttsr, rm.nttasr, K, idx & 127, idx >> 7, b);
    if (tid <= NOPEN) { // sr_rnd / er_rnd / ntt_Asr_rnd tails, as k_copy_tails
        uint16_t *Pb = P + (size_t)b * proof_stride + NSEC + tid;
        for (int i = 0; i < K; i++) {
            const uint16_t sv = Pb[(size_t)(rm.sr + i) * RS], ev = Pb[(size_t)(rm.er + i) * RS];
            Pb[(size_t)(rm.nttsr + i) * RS] = sv;
            Pb[(size_t)(rm.nttasr + i) * RS] = sv;
            Pb[(size_t)(rm.ntter + i) * RS] = ev;
        }
    }
}

// =========================================================================
// K1/K2  C[n][c_off + m] = sum_k A[m][k] * B[n][k]  mod q   on the matrix cores.
// GF(3329) values are exact in two int8 limbs of the centred representative (c = c0 + 64 c1), so
//   A*B = sum a0 b0 + 64 sum (a0 b1 + a1 b0) + 4096 sum a1 b1
// is four v_mfma_i32_16x16x64_i8 per 16x16x64 block into three exact i32 accumulators
// (|S0| <= 2^20, |S1| <= 2^21, |S2| <= 2^20 for k <= 832), recombined mod q (4096 = 767 mod q).
// Both operands arrive pre-tiled as limb matrices (kosk_device.hpp); a workgroup computes 128 m x 64 n,
// a wave 64 x 32 (4 x 2 MFMA blocks), double-buffered through LDS with plain 16-byte copies.
// =========================================================================

// canonical u16 rows -> limb matrix; one thread per (row, 16-k chunk)
__global__ __launch_bounds__(256) void k_rows_to_limbs(LimbArgs a)
{
    const int t = threadIdx.x;
    const int ks = blockIdx.x * 4 + (t >> 6);
    const int rt = blockIdx.y; // destination row tile
    if (ks >= a.KS) return;
    const int rr = t & 15, kc = (t >> 4) & 3;
    const int r = rt * 16 + rr;
    const int g = r / a.npg_pad, i = r - g * a.npg_pad;
    uint32_t lo[4] = {0, 0, 0, 0}, hi[4] = {0, 0, 0, 0};
    if (g < a.ngroups && i < a.npg) {
        const int k0 = ks * 64 + kc * 16;
        const uint16_t *src = a.src + (size_t)g * a.src_gstride + (size_t)(a.rows ? (int)a.rows[i] : i) * a.src_rstride + a.src_koff + k0;
        uint16_t v[16];
        if (k0 + 16 <= a.ncols && ((a.src_koff | a.src_rstride) & 7) == 0 && (a.src_gstride & 7) == 0) {
            const uint4 x0 = *reinterpret_cast<const uint4 *>(src), x1 = *reinterpret_cast<const uint4 *>(src + 8);
            const uint32_t w[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
#pragma unroll
            for (int q = 0; q < 8; q++) { v[2 * q] = (uint16_t)w[q]; v[2 * q + 1] = (uint16_t)(w[q] >> 16); }
        } else {
#pragma unroll
            for (int q = 0; q < 16; q++) v[q] = (k0 + q < a.ncols) ? src[q] : (uint16_t)0;
        }
#pragma unroll
        for (int q = 0; q < 16; q++) {
            int c0, c1;
            limb_split(gf_center(v[q] >= Q ? v[q] % Q : v[q]), c0, c1);
            lo[q >> 2] |= ((uint32_t)c0 & 0xFFu) << (8 * (q & 3));
            hi[q >> 2] |= ((uint32_t)c1 & 0xFFu) << (8 * (q & 3));
        }
    }
    uint8_t *d = a.dst + ((size_t)(ks * a.RT + rt) * 2) * 1024 + rr * 64 + ((kc ^ limb_swz(rr)) << 4);
    *reinterpret_cast<uint4 *>(d) = make_uint4(lo[0], lo[1], lo[2], lo[3]);
    *reinterpret_cast<uint4 *>(d + 1024) = make_uint4(hi[0], hi[1], hi[2], hi[3]);
}

constexpr int GM_TM = 128, GM_TN = 64;            // workgroup tile
constexpr int GM_A_BYTES = (GM_TM / 16) * 2048;    // 16 KiB per k-step
constexpr int GM_B_BYTES = (GM_TN / 16) * 2048;    //  8 KiB per k-step

// BLIMB: the data operand is already a limb matrix (lincomb coefficients); otherwise it is converted from
// canonical u16 rows while it is staged (16 values per thread and k-step), which saves a conversion launch.
template <bool BLIMB>
__device__ __forceinline__ void gemm_modq_block(const GemmArgs &a, const int bx, const int by, const int bz,
                                                uint8_t (&lds)[2][GM_A_BYTES + GM_B_BYTES])
{
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wm = w & 1, wn = w >> 1;
    const int grp = a.grouped ? bz : 0;
    const int mt0 = bx * (GM_TM / 16);     // first A row tile of this workgroup
    const int ART = a.Mpad / 16;
    // staging: A 16 KiB = 4 x 16 B per thread and k-step, contiguous in the limb matrix.
    // (named registers on purpose: arrays captured by a lambda end up in scratch memory)
    const uint4 *__restrict__ ap = reinterpret_cast<const uint4 *>(a.A + (size_t)grp * a.a_gstride + (size_t)mt0 * 2048) + tid;
    const size_t a_step = (size_t)ART * 128; // uint4 per k-step
    // B: either 8 KiB of limb tiles (2 x 16 B per thread), or 64 rows x 64 u16 (32 B per thread: row tid>>2, chunk tid&3)
    const uint4 *__restrict__ bp;
    size_t b_step;
    int b_lds; // byte offset of this thread's converted 16 bytes inside the B region (limb 0)
    bool b_ok = true;
    if (BLIMB) {
        const int nt0 = by * (GM_TN / 16) + (a.grouped ? grp * (a.npg_pad / 16) : 0);
        bp = reinterpret_cast<const uint4 *>(a.B + (size_t)nt0 * 2048) + tid;
        b_step = (size_t)a.BRT * 128;
        b_lds = 0;
    } else {
        const int row_l = tid >> 2, kc = tid & 3;
        const int n_loc = by * GM_TN + row_l;
        int g, i;
        if (a.grouped) { g = grp; i = n_loc; b_ok = i < a.npg; }
        else { g = n_loc / a.npg; i = n_loc - g * a.npg; b_ok = n_loc < a.npg * a.ngroups; }
        const size_t off = b_ok ? (size_t)g * a.src_gstride + (size_t)(a.src_rows ? (int)a.src_rows[i] : i) * a.src_rstride + a.src_koff + kc * 16 : 0;
        bp = reinterpret_cast<const uint4 *>(a.src + off);
        b_step = 8; // 64 u16 per k-step
        b_lds = (row_l >> 4) * 2048 + (row_l & 15) * 64 + ((kc ^ limb_swz(row_l & 15)) << 4);
    }
    uint4 ra0, ra1, ra2, ra3, rb0, rb1;
#define GM_GLOAD()                                                           \
    ra0 = ap[0]; ra1 = ap[256]; ra2 = ap[512]; ra3 = ap[768];                \
    if (BLIMB) { rb0 = bp[0]; rb1 = bp[256]; }                               \
    else if (b_ok) { rb0 = bp[0]; rb1 = bp[1]; }                             \
    else { rb0 = make_uint4(0, 0, 0, 0); rb1 = rb0; }                        \
    ap += a_step; bp += b_step;
#define GM_LSTORE(buf)                                                       \
    {                                                                        \
        uint4 *la_ = reinterpret_cast<uint4 *>(lds[buf]) + tid;              \
        la_[0] = ra0; la_[256] = ra1; la_[512] = ra2; la_[768] = ra3;        \
        if (BLIMB) {                                                         \
            uint4 *lb_ = reinterpret_cast<uint4 *>(lds[buf] + GM_A_BYTES) + tid; \
            lb_[0] = rb0; lb_[256] = rb1;                                    \
        } else {                                                             \
            uint4 lo_, hi_;                                                  \
            gm_split16(rb0, rb1, lo_, hi_);                                  \
            uint8_t *lb_ = lds[buf] + GM_A_BYTES + b_lds;                    \
            *reinterpret_cast<uint4 *>(lb_) = lo_;                           \
            *reinterpret_cast<uint4 *>(lb_ + 1024) = hi_;                    \
        }                                                                    \
    }

    v4i s0[4][2], s1[4][2], s2[4][2];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) { s0[i][j] = (v4i){0, 0, 0, 0}; s1[i][j] = s0[i][j]; s2[i][j] = s0[i][j]; }

    // fragment address inside a 1 KiB tile: row l&15, k-chunk l>>4 (swizzled)
    const int frag = (lane & 15) * 64 + (((lane >> 4) ^ limb_swz(lane & 15)) << 4);

    GM_GLOAD();
    GM_LSTORE(0);
    __syncthreads();
    for (int ks = 0; ks < a.KS; ks++) {
        const int buf = ks & 1;
        if (ks + 1 < a.KS) { GM_GLOAD(); }
        const uint8_t *la = lds[buf] + (wm * 4) * 2048 + frag;
        const uint8_t *lb = lds[buf] + GM_A_BYTES + (wn * 2) * 2048 + frag;
        v4i a0[4], a1[4], b0[2], b1[2];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            a0[i] = *reinterpret_cast<const v4i *>(la + i * 2048);
            a1[i] = *reinterpret_cast<const v4i *>(la + i * 2048 + 1024);
        }
#pragma unroll
        for (int j = 0; j < 2; j++) {
            b0[j] = *reinterpret_cast<const v4i *>(lb + j * 2048);
            b1[j] = *reinterpret_cast<const v4i *>(lb + j * 2048 + 1024);
        }
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 2; j++) {
                s0[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a0[i], b0[j], s0[i][j], 0, 0, 0);
                s1[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a0[i], b1[j], s1[i][j], 0, 0, 0);
                s1[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a1[i], b0[j], s1[i][j], 0, 0, 0);
                s2[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a1[i], b1[j], s2[i][j], 0, 0, 0);
            }
        if (ks + 1 < a.KS) {
            if (buf) { GM_LSTORE(0); } else { GM_LSTORE(1); }
        }
        __syncthreads();
    }
#undef GM_GLOAD
#undef GM_LSTORE

    // D[row = m: 4(l>>4)+r][col = n: l&15] -> four consecutive m per lane: one 8-byte store per block
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const int n_loc = by * GM_TN + wn * 32 + j * 16 + (lane & 15); // row inside the group (grouped) or flat
        int g, i;
        bool valid;
        if (a.grouped) { g = grp; i = n_loc; valid = i < a.npg; }
        else { g = n_loc / a.npg; i = n_loc - g * a.npg; valid = n_loc < a.npg * a.ngroups; }
        if (!valid) continue;
        const int gd = a.c_gdiv > 1 ? a.c_gdiv : 1;
        uint16_t *crow = a.C + (size_t)(g / gd) * a.c_gstride +
                         (size_t)(a.c_rows ? (int)a.c_rows[(g % gd) * a.c_rows_gstride + i] : i) * a.c_rstride + a.c_off;
#pragma unroll
        for (int ib = 0; ib < 4; ib++) {
            const int m0 = bx * GM_TM + wm * 64 + ib * 16 + (lane >> 4) * 4;
            if (m0 >= a.M) continue;
            uint32_t v[4];
#pragma unroll
            for (int r = 0; r < 4; r++) v[r] = gf_reduce_limbs(s0[ib][j][r], s1[ib][j][r], s2[ib][j][r]); // k <= 832
            if (m0 + 4 <= a.M) {
                *reinterpret_cast<uint2 *>(crow + m0) = make_uint2(v[0] | (v[1] << 16), v[2] | (v[3] << 16));
            } else { // M % 4 != 0 (the 407-point interpolation operator): never write past the logical output
#pragma unroll
                for (int r = 0; r < 4; r++)
                    if (m0 + r < a.M) crow[m0 + r] = (uint16_t)v[r];
            }
        }
    }
}

template <bool BLIMB>
__global__ __launch_bounds__(256) void k_gemm_modq(GemmArgs a)
{
    __shared__ __attribute__((aligned(16))) uint8_t lds[2][GM_A_BYTES + GM_B_BYTES];
    gemm_modq_block<BLIMB>(a, blockIdx.x, blockIdx.y, blockIdx.z, lds);
}

// ---- table products with the DATA ROWS resident in LDS (k_table_gemm) ------------------------------------------
// C[n][c_off + m] = sum_k T[m][k] * X[n][k] mod q for a table T shared by every row: the Lagrange expansion
// (ss.cpp:23-32, :88-97; T = 1344 x 407) and recon_secrets_ddeg (ss.cpp:44-51; T = 256 x 407).
// The generic kernel above re-streams a 128 x 64 table tile AND a 64-row data tile for every k-step of every output
// tile (87 MAC per byte staged) and needs the data operand as a limb matrix in HBM, written by one launch and re-read
// by each of the 11 table tiles.  Here a workgroup owns 48 data rows: they are converted to limbs ONCE, straight from
// the canonical u16 rows into LDS (42 KiB), and stay there while the workgroup's 8 waves walk the table.  Each wave
// takes 16-row chunks of the table on its own (no workgroup barrier after the prologue): the chunk's MFMA fragments
// (7 k-steps x 2 limbs x 16 bytes per lane) are loaded from the L2-resident table limb matrix straight into registers -- a
// fragment is 16 contiguous bytes of a 1 KiB tile, the wave reads each tile exactly once -- and every register set is
// re-loaded for the NEXT chunk right after its MFMAs have been issued, so a whole chunk (7 k-steps, ~1350 MFMA cycles)
// of loads is in flight behind the arithmetic.  After the last k-step the wave reduces mod q and stores 16 x 48 outputs;
// with two waves per SIMD that epilogue runs under the partner's MFMAs.
// HBM traffic = the data rows once + the output once + the table once (L2-resident afterwards).
// In-place use (the expansion writes points >= 384 of the rows it reads points < 448 of): the rows are read in the
// prologue only; with the table split over several workgroups per row block (msplit) another workgroup may already be
// writing points 384..447 of the same rows -- points < 407 are rewritten with their own values (identity rows of the
// table) and points >= 407 meet zero table columns, so any value read there is harmless.
constexpr int TG_WAVES = 8;
// NBT = data row tiles (of 16) per workgroup: 3 (48 rows, 42 KiB of LDS at 7 k-steps) or 4 (64 rows, 56 KiB).  One workgroup is
// resident per CU (172+ VGPRs x 8 waves), so a launch takes ceil(row blocks / CUs) rounds of NBT units each: the launcher picks
// the NBT with the smaller product -- 29 946 rows (138 proofs): 624 blocks of 48 = 3 rounds x 3, 468 blocks of 64 = 2 rounds x 4.

// TG_RT = table row tiles (of 16) per chunk
template <int KS, int TG_RT, int NBT>
__global__ __launch_bounds__(512, (KS <= 7 && NBT == 3) ? 2 : 1) void k_table_gemm(GemmArgs a, int nchunks, int chunks_per_block, int nblk, int msplit, int wide_stores)
{
    constexpr int TG_CHUNK = 16 * TG_RT, TG_NB = 16 * NBT;
    __shared__ __attribute__((aligned(16))) uint8_t ldsB[KS * NBT * 2048]; // [k-step][row tile][limb][1 KiB]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int ntot = a.npg * a.ngroups;
    // one-dimensional grid in XCD-aware order: the msplit workgroups that share a row block (each converts the same 48 rows) get
    // consecutive virtual ids, i.e. one XCD and one L2 (as a 2D grid they sat on msplit different XCDs: the rows were fetched
    // from HBM msplit times)
    const int vid = xcd_virtual_id();
    const int bxr = vid / msplit, byr = vid - bxr * msplit;
    if (bxr >= nblk) return; // grid padding (whole workgroup, before any barrier)
    const int n0 = bxr * TG_NB;

    // ---- prologue: this workgroup's data rows, u16 -> limbs, into LDS (every row is read exactly once from HBM);
    // all loads of a thread are issued before the first conversion
    constexpr int ITEMS = TG_NB * KS * 4, PER = (ITEMS + 511) / 512;
    uint4 x0[PER], x1[PER];
#pragma unroll
    for (int q = 0; q < PER; q++) {
        const int item = tid + q * 512;
        const int row_l = item / (KS * 4), kc16 = item - row_l * (KS * 4);
        const int n = n0 + row_l;
        x0[q] = make_uint4(0, 0, 0, 0);
        x1[q] = x0[q];
        if (item < ITEMS && n < ntot) {
            const int g = n / a.npg, i = n - g * a.npg;
            const uint16_t *src = a.src + (size_t)g * a.src_gstride + (size_t)(a.src_rows ? (int)a.src_rows[i] : i) * a.src_rstride + a.src_koff + kc16 * 16;
            x0[q] = *reinterpret_cast<const uint4 *>(src);
            x1[q] = *reinterpret_cast<const uint4 *>(src + 8);
        }
    }
#pragma unroll
    for (int q = 0; q < PER; q++) {
        const int item = tid + q * 512;
        if (item < ITEMS) {
            const int row_l = item / (KS * 4), kc16 = item - row_l * (KS * 4);
            uint4 lo, hi;
            gm_split16(x0[q], x1[q], lo, hi);
            uint8_t *d = ldsB + ((kc16 >> 2) * NBT + (row_l >> 4)) * 2048 + (row_l & 15) * 64 + (((kc16 & 3) ^ limb_swz(row_l & 15)) << 4);
            *reinterpret_cast<uint4 *>(d) = lo;
            *reinterpret_cast<uint4 *>(d + 1024) = hi;
        }
    }

    const int c_begin = byr * chunks_per_block;
    const int c_end = c_begin + chunks_per_block < nchunks ? c_begin + chunks_per_block : nchunks;
    const int c_first = c_begin + w;
    const int nmy = c_first < c_end ? (c_end - c_first + TG_WAVES - 1) / TG_WAVES : 0; // chunks c_first, c_first + 8, ...
    const int ART = a.Mpad / 16;
    // fragment address inside a 1 KiB LDS tile: row lane & 15, k-chunk lane >> 4 (swizzled)
    const int frag = (lane & 15) * 64 + (((lane >> 4) ^ limb_swz(lane & 15)) << 4);
    // this lane's table fragments of chunk c, k-step ks: tiles (row tile TG_RT c + i, limb) of k-step ks; the table copy
    // a.Afrag keeps every tile in fragment order, so the wave's load of a tile is one linear 1 KiB read
    v4i fa[KS][2 * TG_RT];
    auto load_chunk_ks = [&](int c, int ks, v4i (&dst)[2 * TG_RT]) {
        const uint8_t *src = a.Afrag + ((size_t)(ks * ART + TG_RT * c) * 2) * 1024 + lane * 16;
#pragma unroll
        for (int q = 0; q < 2 * TG_RT; q++) dst[q] = *reinterpret_cast<const v4i *>(src + q * 1024);
    };
    if (nmy > 0) {
#pragma unroll
        for (int ks = 0; ks < KS; ks++) load_chunk_ks(c_first, ks, fa[ks]);
    }
    __syncthreads(); // the only workgroup barrier: from here on the waves run independently
    if (nmy == 0) return;
    // output rows of this lane's NBT columns (n = n0 + 16 j + (lane & 15))
    uint16_t *crow[NBT];
#pragma unroll
    for (int j = 0; j < NBT; j++) {
        const int n = n0 + j * 16 + (lane & 15);
        crow[j] = nullptr;
        if (n < ntot) {
            const int g = n / a.npg, i = n - g * a.npg;
            crow[j] = a.C + (size_t)g * a.c_gstride + (size_t)(a.c_rows ? (int)a.c_rows[i] : i) * a.c_rstride + a.c_off + (lane >> 4) * 4;
        }
    }

    v4i s0[TG_RT][NBT], s1[TG_RT][NBT], s2[TG_RT][NBT];
    const v4i zero4 = {0, 0, 0, 0};

    // the data fragments of the next k-step are read from LDS while the current k-step multiplies; the rows are the same
    // for every chunk, so the last k-step prefetches k-step 0 again (into a buffer of its own: KS is odd)
    v4i fb[2][2 * NBT], fb0[2 * NBT];
    auto load_b = [&](int ks, v4i (&dst)[2 * NBT]) {
        const uint8_t *lb = ldsB + ks * NBT * 2048 + frag;
#pragma unroll
        for (int j = 0; j < NBT; j++) {
            dst[2 * j] = *reinterpret_cast<const v4i *>(lb + j * 2048);
            dst[2 * j + 1] = *reinterpret_cast<const v4i *>(lb + j * 2048 + 1024);
        }
    };
    // ---- epilogue pieces.  D[row = m: 4 (lane >> 4) + r][col = n: lane & 15]: four consecutive m per lane and block
    const int grp = lane >> 4;
    // blocks j, j + 1 of chunk c: with 16-byte aligned rows the odd 16-lane rows of block j's packed values are swapped with the
    // even rows of block j + 1's (v_permlane16_swap), after which an even-row lane holds EIGHT consecutive m of block j and an
    // odd-row lane eight of block j + 1: one 16-byte store per lane instead of two 8-byte stores
    auto store_pair = [&](int j, int c, uint2 pj, uint2 pk) {
        if (wide_stores) {
            const auto sx = __builtin_amdgcn_permlane16_swap(pj.x, pk.x, false, false);
            const auto sy = __builtin_amdgcn_permlane16_swap(pj.y, pk.y, false, false);
            uint16_t *p = (grp & 1) ? (crow[j + 1] ? crow[j + 1] - 4 : nullptr) : crow[j];
            if (p) *reinterpret_cast<uint4 *>(p + c * TG_CHUNK) = make_uint4(sx[0], sy[0], sx[1], sy[1]);
        } else {
            if (crow[j]) *reinterpret_cast<uint2 *>(crow[j] + c * TG_CHUNK) = pj;
            if (crow[j + 1]) *reinterpret_cast<uint2 *>(crow[j + 1] + c * TG_CHUNK) = pk;
        }
    };
    auto store_one = [&](int j, int c, uint2 pj) {
        if (crow[j]) *reinterpret_cast<uint2 *>(crow[j] + c * TG_CHUNK) = pj;
    };

    static_assert(TG_RT == 1, "the epilogue is written for one table row tile per chunk");
    // (Tried and not kept, profiles/r04_gemm_stamps.txt: the reduction and the stores of chunk c issued inside chunk c + 1's k-steps,
    // two vector instructions behind every MFMA.  The k-steps grew by exactly what the epilogue shrank -- two waves per SIMD leave
    // no idle issue slots under the MFMAs -- so the epilogue stays where it was, behind the chunk's last k-step.)
    load_b(0, fb0);
    for (int ci = 0; ci < nmy; ci++) {
        const int c = c_first + ci * TG_WAVES;
        const int cn = ci + 1 < nmy ? c + TG_WAVES : c; // the chunk to prefetch (the last one re-loads itself: harmless)
#pragma unroll
        for (int ks = 0; ks < KS; ks++) {
            v4i(&bc)[2 * NBT] = ks == 0 ? fb0 : fb[ks & 1];
            v4i(&bn)[2 * NBT] = ks + 1 == KS ? fb0 : fb[(ks & 1) ^ 1];
            load_b(ks + 1 < KS ? ks + 1 : 0, bn);
            __builtin_amdgcn_sched_barrier(0); // the reads for the NEXT k-step go out before this k-step's MFMAs, not after them
            // four limb products per 16 x 16 x 64 block, ordered so that no accumulator is used twice in a row; the first
            // k-step starts from a zero operand instead of zeroed registers
#pragma unroll
            for (int i = 0; i < TG_RT; i++) {
#pragma unroll
                for (int j = 0; j < NBT; j++) s0[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa[ks][2 * i], bc[2 * j], ks == 0 ? zero4 : s0[i][j], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < NBT; j++) s1[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa[ks][2 * i], bc[2 * j + 1], ks == 0 ? zero4 : s1[i][j], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < NBT; j++) s1[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa[ks][2 * i + 1], bc[2 * j], s1[i][j], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < NBT; j++) s2[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa[ks][2 * i + 1], bc[2 * j + 1], ks == 0 ? zero4 : s2[i][j], 0, 0, 0);
            }
            load_chunk_ks(cn, ks, fa[ks]); // in flight for a whole chunk of arithmetic before it is used
            __builtin_amdgcn_sched_barrier(0); // keep the k-steps apart: hoisting every LDS read of the chunk costs 150 VGPRs
        }
        // epilogue of the chunk: combine the limb products, reduce mod q, pack four consecutive m per lane and block, store
        uint2 pk[NBT];
#pragma unroll
        for (int j = 0; j < NBT; j++) {
            uint32_t v[4];
#pragma unroll
            for (int r = 0; r < 4; r++) v[r] = gf_reduce_limbs(s0[0][j][r], s1[0][j][r], s2[0][j][r]);
            pk[j] = make_uint2(v[0] | (v[1] << 16), v[2] | (v[3] << 16));
            if (j & 1) store_pair(j - 1, c, pk[j - 1], pk[j]);
            else if (j == NBT - 1) store_one(j, c, pk[j]);
        }
    }
}

// ---- the same product with PERSISTENT workgroups (round 5; default for KS = 7) ----------------------------------------------------
// k_table_gemm gives every 48-row block a workgroup of its own, one workgroup per CU at a time: a launch takes ceil(blocks / CUs) rounds
// (29 946 rows = 624 blocks = 2.44 rounds of work in 3 rounds of time; the verifier's and the re-sharing products of a 138-proof step
// are 1.03 .. 1.6 rounds of work in 2), and every block starts with a prologue -- 43 KB of rows from HBM, their conversion, the first
// chunk's table fragments -- during which the matrix pipe idles (9-12 k of a block's 55 k cycles, profiles/r04_gemm_stamps.txt).
// Here the launch is ONE workgroup per CU and the flattened list of (row block, 16-row table chunk) units is dealt evenly over them:
// a workgroup walks its share block by block (the first and the last may be partial: another workgroup has the block's other
// chunks and converts the same rows), its eight waves taking the block's chunks round-robin as before.  The NEXT block's rows are
// fetched and converted inside this block's chunk loop, one 32-byte item per thread and chunk iteration -- loaded at the top of
// the iteration, converted and written to the OTHER half of the LDS buffer behind its epilogue -- so the only thing between two
// blocks is a barrier.  Same fragments, same arithmetic, same stores: bit-identical to the one-block-per-workgroup kernel of rounds 2-4 (k_table_gemm, which still serves the 13-k-step product).
// CANON: the source rows hold canonical values (everything the pipelines produce): packed conversion; otherwise gm_split16 (folds).
template <int KS, int NBT, bool CANON>
__global__ __launch_bounds__(512, 1) void k_table_gemm_p(GemmArgs a, int nchunks, int nblk, int wide_stores, uint32_t npg_magic)
{
    // row n -> (group, index inside the group) by multiply-high with floor(2^32 / npg) and one correction step (as ntt_split)
    auto split = [&](int n, int &g, int &i) {
        uint32_t gg = __umulhi((uint32_t)n, npg_magic);
        int r = n - (int)gg * a.npg;
        if (r >= a.npg) { gg++; r -= a.npg; }
        g = (int)gg;
        i = r;
    };
    constexpr int TG_RT = 1, TG_CHUNK = 16, TG_NB = 16 * NBT, BUF = KS * NBT * 2048;
    __shared__ __attribute__((aligned(16))) uint8_t ldsB[2 * BUF]; // two row blocks: [buffer][k-step][row tile][limb][1 KiB]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int ntot = a.npg * a.ngroups;
    const long total = (long)nblk * nchunks;
    const long u0 = (long)blockIdx.x * total / gridDim.x, u1 = (long)(blockIdx.x + 1) * total / gridDim.x;
    if (u0 >= u1) return;
    int rb = (int)(u0 / nchunks), ca = (int)(u0 - (long)rb * nchunks);
    long u = u0;

    constexpr int ITEMS = TG_NB * KS * 4, PER = (ITEMS + 511) / 512;
    auto item_src = [&](int blk, int item) -> const uint16_t * { // nullptr: behind the last row (zero limbs)
        const int row_l = item / (KS * 4), kc16 = item - row_l * (KS * 4);
        const int n = blk * TG_NB + row_l;
        if (item >= ITEMS || n >= ntot) return nullptr;
        int g, i;
        split(n, g, i);
        return a.src + (size_t)g * a.src_gstride + (size_t)(a.src_rows ? (int)a.src_rows[i] : i) * a.src_rstride + a.src_koff + kc16 * 16;
    };
    auto item_put = [&](int buf, int item, const uint4 &x0, const uint4 &x1) {
        if (item >= ITEMS) return;
        const int row_l = item / (KS * 4), kc16 = item - row_l * (KS * 4);
        uint4 lo, hi;
        if constexpr (CANON) gm_split16_pk(x0, x1, lo, hi);
        else gm_split16(x0, x1, lo, hi);
        uint8_t *d = ldsB + buf * BUF + ((kc16 >> 2) * NBT + (row_l >> 4)) * 2048 + (row_l & 15) * 64 + (((kc16 & 3) ^ limb_swz(row_l & 15)) << 4);
        *reinterpret_cast<uint4 *>(d) = lo;
        *reinterpret_cast<uint4 *>(d + 1024) = hi;
    };
    { // the first block's rows: every load of a thread in flight before the first conversion
        uint4 x0[PER], x1[PER];
#pragma unroll
        for (int q = 0; q < PER; q++) {
            const uint16_t *src = item_src(rb, tid + q * 512);
            x0[q] = make_uint4(0, 0, 0, 0);
            x1[q] = x0[q];
            if (src) { x0[q] = *reinterpret_cast<const uint4 *>(src); x1[q] = *reinterpret_cast<const uint4 *>(src + 8); }
        }
#pragma unroll
        for (int q = 0; q < PER; q++) item_put(0, tid + q * 512, x0[q], x1[q]);
    }
    const int ART = a.Mpad / 16;
    const int frag = (lane & 15) * 64 + (((lane >> 4) ^ limb_swz(lane & 15)) << 4);
    v4i fa[KS][2 * TG_RT];
    auto load_chunk_ks = [&](int c, int ks, v4i (&dst)[2 * TG_RT]) {
        const uint8_t *src = a.Afrag + ((size_t)(ks * ART + TG_RT * c) * 2) * 1024 + lane * 16;
#pragma unroll
        for (int q = 0; q < 2 * TG_RT; q++) dst[q] = *reinterpret_cast<const v4i *>(src + q * 1024);
    };
    {
        const int c_first = ca + w < nchunks ? ca + w : 0; // (a wave without a chunk in the first block loads a harmless one)
#pragma unroll
        for (int ks = 0; ks < KS; ks++) load_chunk_ks(c_first, ks, fa[ks]);
    }
    __syncthreads();
    const v4i zero4 = {0, 0, 0, 0}, bias4 = {LIMB_BIAS, LIMB_BIAS, LIMB_BIAS, LIMB_BIAS};
    const int grp = lane >> 4;
    int buf = 0;
    while (u < u1) {
        const int cb = (long)(nchunks - ca) <= u1 - u ? nchunks : ca + (int)(u1 - u); // this block's chunks [ca, cb)
        const bool have_next = u + (cb - ca) < u1;
        const uint8_t *lds_cur = ldsB + buf * BUF;
        // output rows of this lane's NBT columns (n = n0 + 16 j + (lane & 15))
        uint16_t *crow[NBT];
#pragma unroll
        for (int j = 0; j < NBT; j++) {
            const int n = rb * TG_NB + j * 16 + (lane & 15);
            crow[j] = nullptr;
            if (n < ntot) {
                int g, i;
                split(n, g, i);
                crow[j] = a.C + (size_t)g * a.c_gstride + (size_t)(a.c_rows ? (int)a.c_rows[i] : i) * a.c_rstride + a.c_off + (lane >> 4) * 4;
            }
        }
        auto store_pair = [&](int j, int c, uint2 pj, uint2 pk) {
            if (wide_stores) {
                const auto sx = __builtin_amdgcn_permlane16_swap(pj.x, pk.x, false, false);
                const auto sy = __builtin_amdgcn_permlane16_swap(pj.y, pk.y, false, false);
                uint16_t *p = (grp & 1) ? (crow[j + 1] ? crow[j + 1] - 4 : nullptr) : crow[j];
                if (p) *reinterpret_cast<uint4 *>(p + c * TG_CHUNK) = make_uint4(sx[0], sy[0], sx[1], sy[1]);
            } else {
                if (crow[j]) *reinterpret_cast<uint2 *>(crow[j] + c * TG_CHUNK) = pj;
                if (crow[j + 1]) *reinterpret_cast<uint2 *>(crow[j + 1] + c * TG_CHUNK) = pk;
            }
        };
        auto store_one = [&](int j, int c, uint2 pj) {
            if (crow[j]) *reinterpret_cast<uint2 *>(crow[j] + c * TG_CHUNK) = pj;
        };
        v4i fb[2][2 * NBT], fb0[2 * NBT];
        auto load_b = [&](int ks, v4i (&dst)[2 * NBT]) {
            const uint8_t *lb = lds_cur + ks * NBT * 2048 + frag;
#pragma unroll
            for (int j = 0; j < NBT; j++) {
                dst[2 * j] = *reinterpret_cast<const v4i *>(lb + j * 2048);
                dst[2 * j + 1] = *reinterpret_cast<const v4i *>(lb + j * 2048 + 1024);
            }
        };
        int it = 0; // items of the NEXT block's rows this thread has fetched, converted and written so far
        const int c_first = ca + w;
        const int nmy = c_first < cb ? (cb - c_first + TG_WAVES - 1) / TG_WAVES : 0;
        if (nmy > 0) load_b(0, fb0);
        for (int ci = 0; ci < nmy; ci++) {
            const int c = c_first + ci * TG_WAVES;
            // the chunk to prefetch: this wave's next one in this block, else its first one of the next block (which starts at chunk
            // 0), else the current one again (harmless)
            const int cn = ci + 1 < nmy ? c + TG_WAVES : (have_next && w < nchunks ? w : c);
            uint4 nx0 = make_uint4(0, 0, 0, 0), nx1 = nx0;
            const bool fetch = have_next && it < PER;
            if (fetch) {
                const uint16_t *src = item_src(rb + 1, tid + it * 512);
                if (src) { nx0 = *reinterpret_cast<const uint4 *>(src); nx1 = *reinterpret_cast<const uint4 *>(src + 8); }
            }
            v4i s0[NBT], s1[NBT], s2[NBT];
#pragma unroll
            for (int ks = 0; ks < KS; ks++) {
                v4i(&bc)[2 * NBT] = ks == 0 ? fb0 : fb[ks & 1];
                v4i(&bn)[2 * NBT] = ks + 1 == KS ? fb0 : fb[(ks & 1) ^ 1];
                load_b(ks + 1 < KS ? ks + 1 : 0, bn);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < NBT; j++) s0[j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa[ks][0], bc[2 * j], ks == 0 ? bias4 : s0[j], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < NBT; j++) s1[j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa[ks][0], bc[2 * j + 1], ks == 0 ? zero4 : s1[j], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < NBT; j++) s1[j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa[ks][1], bc[2 * j], s1[j], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < NBT; j++) s2[j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa[ks][1], bc[2 * j + 1], ks == 0 ? zero4 : s2[j], 0, 0, 0);
                load_chunk_ks(cn, ks, fa[ks]);
                __builtin_amdgcn_sched_barrier(0);
            }
            uint2 pk[NBT];
#pragma unroll
            for (int j = 0; j < NBT; j++) {
                uint32_t v[4];
#pragma unroll
                for (int r = 0; r < 4; r++) v[r] = gf_reduce_limbs_biased_lazy(s0[j][r], s1[j][r], s2[j][r]); // the bias came in with the first MFMA
                pk[j] = make_uint2(gf_canon_pair(v[0], v[1]), gf_canon_pair(v[2], v[3]));
                if (j & 1) store_pair(j - 1, c, pk[j - 1], pk[j]);
                else if (j == NBT - 1) store_one(j, c, pk[j]);
            }
            if (fetch) { item_put(buf ^ 1, tid + it * 512, nx0, nx1); it++; }
        }
        if (have_next) {
            for (; it < PER; it++) { // a wave with fewer chunk iterations than items (a short first block): the rest here
                const uint16_t *src = item_src(rb + 1, tid + it * 512);
                uint4 nx0 = make_uint4(0, 0, 0, 0), nx1 = nx0;
                if (src) { nx0 = *reinterpret_cast<const uint4 *>(src); nx1 = *reinterpret_cast<const uint4 *>(src + 8); }
                item_put(buf ^ 1, tid + it * 512, nx0, nx1);
            }
            if (nmy == 0 && w < nchunks) { // this wave sat the block out: its table fragments are those of the clamped chunk, load the next block's
#pragma unroll
                for (int ks = 0; ks < KS; ks++) load_chunk_ks(w, ks, fa[ks]);
            }
        }
        u += cb - ca;
        rb++;
        ca = 0;
        buf ^= 1;
        lds_barrier(); // everybody has written the next block's rows and finished reading this block's
    }
}

// ---- K3 (prover) on the matrix cores: out_j[x] = sum_k Coef[j][k] * in_k[x], per proof a [76 x 77] x [77 x 1710] product.
// (Rounds 2-4 ran it as a transposition pass + the generic GEMM, then as the one-shot kernel k_lincomb_fused -- one workgroup per
// (proof, f | NTT f, 128 points); both are gone since round 6, their measurements are in profiles/r05_* and DESIGN.md 15.2.)
constexpr int LC_JPAD = 80;  // padded J in the alpha / power tables
constexpr int LF_A_BYTES = 2 * 8 * 2048, LF_B_BYTES = 2 * 8 * 2048; // 2 k-steps x 8 row tiles x (2 limbs x 1 KiB)

// ---- K3 (prover), streaming form (round 5).  The [point][k] limb tiles of the MFMA A operand are built in LDS from row-wise reads; a
// workgroup is PERSISTENT over a run of consecutive (group, 128-point block) units of the launch (the flattened list is dealt evenly
// over two workgroups per CU: 7 or 8 blocks each at 138 proofs):
//   * the coefficient tiles of a group are copied to LDS once per run of that group's blocks, not once per block (the one-shot kernel:
//     every one of a group's 14 blocks re-read 32 KiB from L2: 124 MB of L2 -> LDS per launch for 62 MB of input), and only the five
//     16-column tiles that J <= 80 needs (20 KiB);
//   * the NEXT block's 77 x 128 input values are already in flight (20 dwords per thread, issued right after the barrier that
//     publishes this block's operand tiles) while the matrix cores and the epilogue work on this block, so a workgroup never
//     sits out a load latency with nothing else to do; the barriers order LDS traffic only (lds_barrier: __syncthreads would wait
//     for the prefetch and for the epilogue's stores);
//   * the u16 -> int8-limb conversion works on BOTH halves of a loaded dword at once with packed 16-bit arithmetic (seven packed
//     instructions and a byte-permute per value pair and limb instead of ~12 scalar ones per value): u = v + 1664 mod q by one
//     unsigned minimum, w = u - 1632 = centred value + 32, low limb = (w & 63) - 32, high limb = w >> 6;
//   * a wave owns 32 points (one MFMA row-tile pair) x all five column tiles: 80 MFMAs per block on every wave (the one-shot kernel's
//     second 64-column half kept two of the four waves idle and computed a column tile nobody stores).
// The k-step 1 tiles hold rows 64 .. M-1 in their first 16-byte chunk only; the other three chunks are zeroed once.
constexpr int LS_JT = 5;                       // 16-column tiles of the coefficient operand that hold j < J <= 80
constexpr int LS_A_BYTES = 2 * 8 * 2048;       // 2 k-steps x 8 point tiles x (2 limbs x 1 KiB)
constexpr int LS_B_BYTES = 2 * LS_JT * 2048;   // 2 k-steps x 5 column tiles
__global__ __launch_bounds__(256, 2) void k_lincomb_stream(const uint16_t *P, size_t proof_stride, int row_f, int row_tf, int M,
                                                           const uint8_t *__restrict__ coef, int BRT, uint16_t *C,
                                                           const int16_t *__restrict__ lin_rows, int J, int K, int row_s, int row_e,
                                                           int row_sr, int row_er, int ngroups)
{
    __shared__ __attribute__((aligned(16))) uint8_t lds[LS_A_BYTES + LS_B_BYTES];
    uint8_t *ldsB = lds + LS_A_BYTES;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    constexpr int MB = (NPTS + 127) / 128;
    const int total = ngroups * MB;
    const int fb0 = (int)(((long)blockIdx.x * total) / gridDim.x), fb1 = (int)(((long)(blockIdx.x + 1) * total) / gridDim.x);
    if (fb0 >= fb1) return;
    // this thread's part of a block: dwords (two points) of rows 16 c4 .. 16 c4 + 15 at point pair pp (k-step 0), and of rows
    // 64 + 4 c4 .. + 3 (k-step 1: the M - 64 <= 15 rows of chunk 4, a quarter of the chunk per thread)
    const int pp = tid & 63, c4 = tid >> 6, xl = pp * 2;
    uint32_t raw[20];
    auto gload = [&](int fb) {
        const int g = fb / MB, m0 = (fb - g * MB) * 128;
        // No predicates: a point pair behind the row's last point (only in a group's last block) reads the row's first pair instead,
        // and the slots of rows M .. 79 of chunk 4 read row M - 1 again (a line this wave has just asked for: no HBM bytes; round 5
        // read the three rows that follow the f rows there: 1.9 % of the launch's traffic).  Neither reaches an output: MFMA rows are
        // independent and those points are never stored; the coefficient columns k >= M are zero (k_coef_limbs).
        const uint16_t *src = P + (size_t)(g >> 1) * proof_stride + (size_t)((g & 1) ? row_tf : row_f) * RS + (m0 + xl < NPTS ? m0 + xl : 0);
#pragma unroll
        for (int q = 0; q < 16; q++) raw[q] = *reinterpret_cast<const uint32_t *>(src + (size_t)(c4 * 16 + q) * RS);
#pragma unroll
        for (int q = 0; q < 4; q++) raw[16 + q] = *reinterpret_cast<const uint32_t *>(src + (size_t)min(64 + c4 * 4 + q, M - 1) * RS);
    };
    // LDS position of point x of the block: MFMA row tile 2 (x >> 5) + ((x >> 2) & 1), row 4 ((x >> 3) & 3) + (x & 3) -- an output lane
    // (row group lane >> 4) then holds eight CONSECUTIVE points across the two tiles of a pair: one 16-byte store per output row
    int a_off[2];
#pragma unroll
    for (int pt = 0; pt < 2; pt++) {
        const int x = xl + pt, xt = (x >> 5) * 2 + ((x >> 2) & 1), xr = ((x >> 3) & 3) * 4 + (x & 3);
        a_off[pt] = xt * 2048 + xr * 64;
    }
    auto stage = [&]() { // raw -> limb tiles of the A operand
        uint32_t lo[20], hi[20];
#pragma unroll
        for (int q = 0; q < 20; q++) limb_split_pk(raw[q], lo[q], hi[q]);
        uint32_t l0[4], l1[4], h0[4], h1[4];
#pragma unroll
        for (int d = 0; d < 4; d++) {
            const uint32_t rl[4] = {lo[4 * d], lo[4 * d + 1], lo[4 * d + 2], lo[4 * d + 3]}, rh[4] = {hi[4 * d], hi[4 * d + 1], hi[4 * d + 2], hi[4 * d + 3]};
            limb_pack4(rl, l0[d], l1[d]);
            limb_pack4(rh, h0[d], h1[d]);
        }
        uint32_t e0, e1, f0, f1;
        {
            const uint32_t rl[4] = {lo[16], lo[17], lo[18], lo[19]}, rh[4] = {hi[16], hi[17], hi[18], hi[19]};
            limb_pack4(rl, e0, e1);
            limb_pack4(rh, f0, f1);
        }
#pragma unroll
        for (int pt = 0; pt < 2; pt++) {
            const int xr = (a_off[pt] >> 6) & 15;
            uint8_t *d0 = lds + a_off[pt] + ((c4 ^ limb_swz(xr)) << 4);                          // k-step 0, chunk c4
            *reinterpret_cast<uint4 *>(d0) = pt ? make_uint4(l1[0], l1[1], l1[2], l1[3]) : make_uint4(l0[0], l0[1], l0[2], l0[3]);
            *reinterpret_cast<uint4 *>(d0 + 1024) = pt ? make_uint4(h1[0], h1[1], h1[2], h1[3]) : make_uint4(h0[0], h0[1], h0[2], h0[3]);
            uint8_t *d1 = lds + 8 * 2048 + a_off[pt] + ((0 ^ limb_swz(xr)) << 4) + c4 * 4;      // k-step 1, chunk 0, bytes 4 c4 ..
            *reinterpret_cast<uint32_t *>(d1) = pt ? e1 : e0;
            *reinterpret_cast<uint32_t *>(d1 + 1024) = pt ? f1 : f0;
        }
    };
    gload(fb0);
    // k-step 1, chunks 1..3 (rows 80 .. 127 of the padded k dimension): zero, once
    for (int i = tid; i < 8 * 2 * 16 * 3; i += 256) { // (tile, limb, row, chunk 1..3)
        const int ch = 1 + i % 3, r = (i / 3) & 15, tl = i / 48; // tl = tile * 2 + limb
        *reinterpret_cast<uint4 *>(lds + 8 * 2048 + tl * 1024 + r * 64 + ((ch ^ limb_swz(r)) << 4)) = make_uint4(0, 0, 0, 0);
    }
    const int frag = (lane & 15) * 64 + (((lane >> 4) ^ limb_swz(lane & 15)) << 4);
    int g_cur = -1;
    // output rows of the group's columns, in LDS (looked up once per group: a global lookup inside the epilogue would make it wait for
    // the prefetch, and five more registers per lane do not fit beside 120 accumulators and the prefetch)
    __shared__ int16_t lrow_s[16 * LS_JT];
    for (int fb = fb0; fb < fb1; fb++) {
        const int g = fb / MB, m0 = (fb - g * MB) * 128, b = g >> 1, which = g & 1;
        if (g != g_cur) { // this group's coefficient tiles (the previous group's readers passed the barrier at the end of its last block)
            g_cur = g;
            if (tid < 16 * LS_JT) lrow_s[tid] = tid < J ? lin_rows[which * 128 + tid] : (int16_t)0; // (read behind the barrier that follows stage())
#pragma unroll
            for (int ks = 0; ks < 2; ks++) {
                const uint4 *src = reinterpret_cast<const uint4 *>(coef + ((size_t)ks * BRT + (size_t)g * 8) * 2048);
                uint4 *dst = reinterpret_cast<uint4 *>(ldsB + ks * LS_JT * 2048);
                for (int i = tid; i < LS_JT * 128; i += 256) dst[i] = src[i];
            }
        }
        stage();
        lds_barrier();
        if (fb + 1 < fb1) gload(fb + 1);
        __builtin_amdgcn_sched_barrier(0); // the prefetch goes out before the arithmetic below, not behind it (issued ahead of the LDS writes of
                                           // stage() instead -- tried in round 6 -- the launch is 4 % slower: 86.0 against 82.8 us per 276 proofs)
        v4i s0[2][LS_JT], s1[2][LS_JT], s2[2][LS_JT];
        const v4i zero4 = {0, 0, 0, 0};
#pragma unroll
        for (int ks = 0; ks < 2; ks++) {
            const uint8_t *la = lds + (ks * 8 + w * 2) * 2048 + frag;
            const uint8_t *lb = ldsB + (ks * LS_JT) * 2048 + frag;
            v4i a0[2], a1[2];
#pragma unroll
            for (int i = 0; i < 2; i++) {
                a0[i] = *reinterpret_cast<const v4i *>(la + i * 2048);
                a1[i] = *reinterpret_cast<const v4i *>(la + i * 2048 + 1024);
            }
#pragma unroll
            for (int j = 0; j < LS_JT; j++) {
                const v4i b0 = *reinterpret_cast<const v4i *>(lb + j * 2048), b1 = *reinterpret_cast<const v4i *>(lb + j * 2048 + 1024);
#pragma unroll
                for (int i = 0; i < 2; i++) {
                    s0[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a0[i], b0, ks == 0 ? zero4 : s0[i][j], 0, 0, 0);
                    s1[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a0[i], b1, ks == 0 ? zero4 : s1[i][j], 0, 0, 0);
                    s2[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a1[i], b1, ks == 0 ? zero4 : s2[i][j], 0, 0, 0);
                    s1[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a1[i], b0, s1[i][j], 0, 0, 0);
                }
            }
        }
        uint16_t *Cb = C + (size_t)b * proof_stride;
        const int m = m0 + w * 32 + (lane >> 4) * 8; // eight consecutive points per lane (see a_off)
        if (m < NPTS) { // the last live group (1704..1711) ends two points inside the row padding (RS = 1728)
#pragma unroll
            for (int j = 0; j < LS_JT; j++) {
                const int jo = j * 16 + (lane & 15);
                if (jo >= J) continue;
                uint16_t *crow = Cb + (size_t)lrow_s[jo] * RS;
                uint32_t v[8];
#pragma unroll
                for (int h = 0; h < 2; h++)
#pragma unroll
                    for (int r = 0; r < 4; r++) v[4 * h + r] = gf_reduce_limbs_lazy(s0[h][j][r], s1[h][j][r], s2[h][j][r]);
                const uint32_t pw[4] = {gf_canon_pair(v[0], v[1]), gf_canon_pair(v[2], v[3]), gf_canon_pair(v[4], v[5]), gf_canon_pair(v[6], v[7])};
                *reinterpret_cast<uint4 *>(crow + m) = make_uint4(pw[0], pw[1], pw[2], pw[3]);
                if (j == LS_JT - 1 && which == 0 && jo >= NCHK) { // r rows (columns 70 .. 69 + 2K, all in the last tile): s + r_i, e + r_{K+i} on the spot (mlwe_prover.cpp:222-245)
                    const int idx = jo - NCHK;
                    const int src_row = idx < K ? row_s + idx : row_e + (idx - K), dst_row = idx < K ? row_sr + idx : row_er + (idx - K);
                    const uint4 sv = *reinterpret_cast<const uint4 *>(Cb + (size_t)src_row * RS + m);
                    *reinterpret_cast<uint4 *>(Cb + (size_t)dst_row * RS + m) = make_uint4(gf_add_pair(sv.x, pw[0]), gf_add_pair(sv.y, pw[1]), gf_add_pair(sv.z, pw[2]), gf_add_pair(sv.w, pw[3]));
                }
            }
        }
        lds_barrier(); // every wave has read this block's tiles (and this group's coefficients) before the next block overwrites them
    }
}

// "B" operand: Coef[j][k] = alpha_j^k for the 70 check rows; for the r rows (j >= 70) the constant term is
// f_71 instead of f_0 (mlwe_prover.cpp:187,196): Coef[j][0] = 0 and Coef[j][71] = alpha_j^71 + 1.
// Written for both groups (f and NTT f) of the proof.  One thread per (j, 16-k chunk).
__global__ __launch_bounds__(1024) void k_coef_limbs(const uint16_t *__restrict__ alpha, int J, int M, uint8_t *__restrict__ B, int BRT)
{
    const int j = threadIdx.x & 127, ch = threadIdx.x >> 7, b = blockIdx.x;
    // the challenge vector is read ONCE per proof (host mode: from the page-locked host table, across PCIe; rounds 2-5: every one of the
    // eight k-chunks asked for it itself -- the same 13-14 us per 276 proofs either way: the launch is one PCIe read plus a chain of <= 30 products)
    __shared__ uint16_t al_s[128];
    if (threadIdx.x < 128) al_s[threadIdx.x] = threadIdx.x < J ? alpha[(size_t)b * LC_JPAD + threadIdx.x] : (uint16_t)0;
    __syncthreads();
    auto gf_mul = [](uint32_t x, uint32_t y) { return gf_reduce_u32(__umul24(x, y)); }; // x, y < q: seven full-rate instructions
    uint32_t lo[4] = {0, 0, 0, 0}, hi[4] = {0, 0, 0, 0};
    if (j < J) {
        const uint32_t al = (uint32_t)al_s[j] % (uint32_t)Q;
        uint32_t p = 1, base = al;
        for (int e = ch * 16; e; e >>= 1) { // alpha^(16 ch)
            if (e & 1) p = gf_mul(p, base);
            base = gf_mul(base, base);
        }
#pragma unroll
        for (int q = 0; q < 16; q++) {
            const int k = ch * 16 + q;
            uint32_t c = k < M ? p : 0;
            if (j >= NCHK) {
                if (k == 0) c = 0;
                if (k == NCHK + 1) c = gf_add(c, 1);
            }
            int c0, c1;
            limb_split(gf_center(c), c0, c1);
            lo[q >> 2] |= ((uint32_t)c0 & 0xFFu) << (8 * (q & 3));
            hi[q >> 2] |= ((uint32_t)c1 & 0xFFu) << (8 * (q & 3));
            p = gf_mul(p, al);
        }
    }
    for (int which = 0; which < 2; which++) {
        uint8_t *d = B + limb_offset((2 * b + which) * 128 + j, ch * 16, 0, BRT);
        *reinterpret_cast<uint4 *>(d) = make_uint4(lo[0], lo[1], lo[2], lo[3]);
        *reinterpret_cast<uint4 *>(d + 1024) = make_uint4(hi[0], hi[1], hi[2], hi[3]);
    }
}

// =========================================================================
// K3  beta / gamma / r / NTT_r : out_j[x] = base_j[x] + sum_{k>=1} alpha_j^k in_k[x]
// mlwe_prover.cpp:159-203 (the k == 0 term of the r rows is in_71, :187,:196)
// =========================================================================
constexpr int LC_JC = 20;    // outputs per thread

// pwT[b][k / 2][j] = { lo16: centred alpha_j^k for the even k, hi16: for the odd k + 1 } -- the operand pairs of one v_dot2_i32_i16 per two terms
// of the sum (round 5; before: one int32 and one v_mad per term).  The entry of k == 0 is 0: that term is the chain's base, not a product.
__global__ __launch_bounds__(128) void k_pow_table(const uint16_t *__restrict__ alpha, int J, int M, int32_t *__restrict__ pwT)
{
    const int j = threadIdx.x, b = blockIdx.x;
    if (j >= LC_JPAD) return;
    int32_t *dst = pwT + (size_t)b * MAXM * LC_JPAD + j;
    const uint32_t al = j < J ? alpha[(size_t)b * LC_JPAD + j] % (uint32_t)Q : 0;
    uint32_t p = 1;
    for (int k = 0; k < MAXM + 1; k += 2) {
        const int32_t c0 = (j < J && k < M && k > 0) ? gf_center(p) : 0;
        p = gf_mul(p, al);
        const int32_t c1 = (j < J && k + 1 < M) ? gf_center(p) : 0;
        p = gf_mul(p, al);
        dst[(size_t)(k >> 1) * LC_JPAD] = (int32_t)(((uint32_t)c0 & 0xFFFFu) | ((uint32_t)c1 << 16));
    }
}

__global__ __launch_bounds__(256) void k_lincomb(LincombArgs a)
{
    // one-dimensional grid in XCD-aware order: the workgroups of a (proof, f | NTT f) group read the same input rows
    const int vid = xcd_virtual_id();
    const int per_group = a.nxb * a.njc;
    const int grp = vid / per_group, rem = vid - grp * per_group;
    if (grp >= a.ngroups) return;
    const int jc = rem / a.nxb;
    const int xi = (rem - jc * a.nxb) * 256 + threadIdx.x;
    const int b = grp >> 1, which = grp & 1;
    if (xi >= a.ncols) return;
    // verifier (a.O): entry xi of the opened matrix's rows -- consecutive threads, consecutive addresses; otherwise column
    // `col` of the row matrix
    const int col = a.col_map ? NSEC + (int)a.col_map[(size_t)b * a.col_map_stride + xi] : xi;
    uint16_t *__restrict__ Pb = a.P + (size_t)b * a.proof_stride;
    uint16_t *__restrict__ Ob = a.O ? a.O + (size_t)b * a.o_stride + xi : nullptr;
    const size_t istride = a.O ? (size_t)OS : (size_t)RS;
    const uint16_t *__restrict__ in0 = (a.O ? Ob : Pb + col) + (size_t)(which ? a.rm.tf : a.rm.f) * istride;
    const int32_t *__restrict__ pw = a.pwT + (size_t)b * MAXM * LC_JPAD + jc * LC_JC;

    int32_t acc[LC_JC];
#pragma unroll
    for (int j = 0; j < LC_JC; j++) acc[j] = 0;
    // The verifier's inputs are the image's RAW u16 (a crafted proof may hold elements >= q): every term k >= 1 goes through
    // gf3329_mul in the reference (mlwe_verifier.cpp:76, :85, :157, :166) and is folded here; the k == 0 term does not, see below.
    // two terms per instruction: acc_j += c_j[k] v[k] + c_j[k + 1] v[k + 1] (v_dot2_i32_i16; centred operands, |sum| < 79 * 1665^2 < 2^31);
    // the pair (0, 1) carries coefficient 0 for k == 0, a row at or beyond M reads row M - 1 again under a zero coefficient
    const int M = a.rm.M;
    // two raw u16 -> their centred representatives as one packed pair: (v + 1664) mod q by one packed unsigned minimum, minus 1664.
    // An element >= q (a crafted proof) is folded first, behind a branch the whole wave takes or skips (round 6; rounds 2-5 folded and
    // centred every value on its own: 24 instructions per pair under 20 products, a third of them quarter-rate multiplies).
    auto centre_pair = [](uint32_t x0, uint32_t x1) -> uint32_t {
        if (__any((int)(max(x0, x1) >= (uint32_t)Q))) { x0 = gf_fold(x0); x1 = gf_fold(x1); }
        const us2 x = __builtin_bit_cast(us2, x0 | (x1 << 16));
        const us2 m = __builtin_elementwise_min(x + (us2){1664, 1664}, x - (us2){1665, 1665});
        return __builtin_bit_cast(uint32_t, __builtin_bit_cast(ss2, m) - (ss2){1664, 1664});
    };
    for (int k0 = 0; k0 < M; k0 += 8) { // eight independent loads in flight, then four pairs (all 80 at once: 52 against 43 us per 276 proofs)
        uint32_t raw[8];
#pragma unroll
        for (int i = 0; i < 8; i++) raw[i] = in0[(size_t)(k0 + i < M ? k0 + i : M - 1) * istride];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int k = k0 + 2 * i;
            if (k >= M) break; // uniform
            const uint32_t vv = centre_pair(raw[2 * i], raw[2 * i + 1]); // (a second half at M: its coefficient is 0, k_pow_table)
            const int32_t *pk = pw + (size_t)(k >> 1) * LC_JPAD;
#pragma unroll
            for (int j = 0; j < LC_JC; j++) asm("v_dot2_i32_i16 %0, %1, %2, %0" : "+v"(acc[j]) : "s"(pk[j]), "v"(vv));
        }
    }
    const uint32_t raw_chk = in0[0], raw_r = in0[(size_t)(NCHK + 1) * istride];
#pragma unroll
    for (int jj = 0; jj < LC_JC; jj++) {
        const int j = jc * LC_JC + jj;
        if (j >= a.J) break;
        int row;
        if (j < NCHK) row = which ? a.rm.gamma(j) : a.rm.beta(j);
        else row = (which ? a.rm.nttr : a.rm.r) + (j - NCHK);
        const uint32_t raw = j < NCHK ? raw_chk : raw_r;
        const uint32_t base = gf_fold(raw);
        uint32_t out = gf_from_i32(acc[jj] + (int32_t)base);
        if (raw >= (uint32_t)Q) {
            // The reference starts its chain from the RAW share (:73, :82, :154, :163) and adds the M - 1 reduced products with
            // gf3329_add, which subtracts q at most once per step: a value u + e q (u < q, e >= 1 multiples of q in excess) becomes
            // u + b - q + e q when u + b wraps and u + b + (e - 1) q when it does not -- every non-wrapping step sheds one
            // excess q until none is left.  The number of wrapping steps is floor((u0 + sum of the canonical products) / q)
            // whatever their order, so the chain's result is the canonical sum plus max(0, e0 - non-wrapping steps) q.
            // Only a crafted proof comes here; the sum is recomputed with canonical operands.
            uint32_t sum = base;
            for (int k = 1; k < a.rm.M; k++) {
                const int32_t pp = pw[(size_t)(k >> 1) * LC_JPAD + jj];
                const int32_t pc = (k & 1) ? (pp >> 16) : (int32_t)(int16_t)(pp & 0xFFFF);
                sum += gf_mul((uint32_t)(pc < 0 ? pc + Q : pc), gf_fold(in0[(size_t)k * istride]));
            }
            const int nonwrap = (a.rm.M - 1) - (int)(sum / (uint32_t)Q), e0 = (int)(raw / (uint32_t)Q);
            if (e0 > nonwrap) out += (uint32_t)(e0 - nonwrap) * Q;
        }
        if (Ob) {
            Ob[(size_t)row * OS] = (uint16_t)out;
            // recon_secrets_ddeg (mlwe_verifier.cpp:106-107) reads the beta / gamma shares of parties 0..406 from the merged row
            // (through gf3329_mul, which reduces: the row matrix gets the folded value, the opened matrix keeps the raw one)
            if (j < NCHK && col < NSEC + XLEN) Pb[(size_t)row * RS + col] = (uint16_t)gf_fold(out);
        } else {
            Pb[(size_t)row * RS + col] = (uint16_t)out;
        }
    }
}

// =========================================================================
// K7  share-wise gates (ss.cpp:101-136 call sites in prove())
// =========================================================================

// after the first expansion: s - eta, e - eta, multiplication gates, u = z2d - z_d
// mlwe_prover.cpp:338-381.  Eight evaluation points per thread (one 16-byte access per row).
struct U16x8 {
    uint32_t v[8];
};
__device__ __forceinline__ U16x8 ld8(const uint16_t *p)
{
    const uint4 x = *reinterpret_cast<const uint4 *>(p);
    U16x8 r;
    r.v[0] = x.x & 0xFFFFu; r.v[1] = x.x >> 16; r.v[2] = x.y & 0xFFFFu; r.v[3] = x.y >> 16;
    r.v[4] = x.z & 0xFFFFu; r.v[5] = x.z >> 16; r.v[6] = x.w & 0xFFFFu; r.v[7] = x.w >> 16;
    return r;
}
__device__ __forceinline__ void st8(uint16_t *p, const U16x8 &r)
{
    *reinterpret_cast<uint4 *>(p) = make_uint4(r.v[0] | (r.v[1] << 16), r.v[2] | (r.v[3] << 16), r.v[4] | (r.v[5] << 16), r.v[6] | (r.v[7] << 16));
}

// grid (1, proofs, 2K): one (s|e, i) gate chain per block; all of its E + Z + 1 input rows are
// loaded before the first store so that the loads overlap.
template <int E>
__global__ __launch_bounds__(256) void k_post_gates(uint16_t *__restrict__ P, size_t proof_stride, RowMap rm)
{
    const int x = threadIdx.x * 8, b = blockIdx.y;
    if (x >= NPTS) return; // the last group covers 1704..1711: columns >= 1710 are row padding
    const int who = (int)blockIdx.z >= rm.K, i = blockIdx.z - (who ? rm.K : 0);
    uint16_t *Pb = P + (size_t)b * proof_stride + x;
    const U16x8 v = ld8(Pb + (size_t)((who ? rm.e : rm.s) + i) * RS);
    U16x8 c[E], zd[E - 1];
#pragma unroll
    for (int m = 0; m < E; m++) c[m] = ld8(Pb + (size_t)((who ? rm.eeta : rm.seta) + i * E + m) * RS);
#pragma unroll
    for (int m = 0; m < E - 1; m++) zd[m] = ld8(Pb + (size_t)(who ? rm.ze(i, m) : rm.zs(i, m)) * RS);
#pragma unroll
    for (int m = 0; m < E; m++) {
        U16x8 d, u;
#pragma unroll
        for (int q = 0; q < 8; q++) {
            d.v[q] = gf_sub(v.v[q], c[m].v[q]);
            if (m > 0) u.v[q] = gf_sub(gf_mul(m == 1 ? gf_sub(v.v[q], c[0].v[q]) : zd[m > 1 ? m - 2 : 0].v[q], d.v[q]), zd[m > 0 ? m - 1 : 0].v[q]);
        }
        st8(Pb + (size_t)((who ? rm.esub : rm.ssub) + i * E + m) * RS, d);
        if (m > 0) st8(Pb + (size_t)(who ? rm.ue(i, m - 1) : rm.us(i, m - 1)) * RS, u);
    }
}

// sr_rnd / er_rnd / ntt_Asr_rnd tails: values at points 256..406 of the re-shared
// rows are those of sr / er / sr                          mlwe_prover.cpp:234-237, :312-314
__global__ __launch_bounds__(192) void k_copy_tails(uint16_t *__restrict__ P, size_t proof_stride, RowMap rm)
{
    const int t = threadIdx.x, i = blockIdx.x, b = blockIdx.y;
    if (t > NOPEN) return;
    uint16_t *Pb = P + (size_t)b * proof_stride + NSEC + t;
    const uint16_t s = Pb[(size_t)(rm.sr + i) * RS], e = Pb[(size_t)(rm.er + i) * RS];
    Pb[(size_t)(rm.nttsr + i) * RS] = s;
    Pb[(size_t)(rm.nttasr + i) * RS] = s;
    Pb[(size_t)(rm.ntter + i) * RS] = e;
}

// NTT(s) = NTT(s+r) - NTT(r), NTT(e) likewise, A r = A(s+r) - A s, t = A s + e
// mlwe_prover.cpp:301-303, :317, :321-323
__global__ __launch_bounds__(256) void k_post_relation(uint16_t *__restrict__ P, size_t proof_stride, RowMap rm)
{
    const int x = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    if (x >= NPTS) return;
    uint16_t *Pb = P + (size_t)b * proof_stride + x;
    for (int i = 0; i < rm.K; i++) {
        const uint32_t ns = gf_sub(Pb[(size_t)(rm.nttsr + i) * RS], Pb[(size_t)(rm.nttr + i) * RS]);
        const uint32_t ne = gf_sub(Pb[(size_t)(rm.ntter + i) * RS], Pb[(size_t)(rm.nttr + rm.K + i) * RS]);
        const uint32_t as = Pb[(size_t)(rm.nttas + i) * RS];
        Pb[(size_t)(rm.ntts + i) * RS] = (uint16_t)ns;
        Pb[(size_t)(rm.ntte + i) * RS] = (uint16_t)ne;
        Pb[(size_t)(rm.nttar + i) * RS] = (uint16_t)gf_sub(Pb[(size_t)(rm.nttasr + i) * RS], as);
        Pb[(size_t)(rm.t + i) * RS] = (uint16_t)gf_add(as, ne);
    }
}

// =========================================================================
// K8  proof wire image                                     mlwe_prover.cpp:480-537
// field element [i][e] = P[rows[e]][256 + sel[i]] ; 64 parties per workgroup,
// LDS transpose so that both the row reads and the image writes are coalesced
// =========================================================================
constexpr int ASM_TILE = 64 * 80; // u16 per tile: 64 parties x the widest field (79)

// width (<= MAXW) elements of one party's record: element e = row rt[e] at this lane's column (byte offset voff inside the row).
// The row indices past the record's width are clamped to its last row (a few redundant loads instead of a branch per element).
template <int MAXW>
__device__ __forceinline__ void asm_gather(const uint16_t *Pb, uint32_t voff, const int16_t *rt, int width, uint16_t *t, bool live)
{
    // the record's row indices first (the compiler reads the 16-bit table with vector loads and waits for them with vmcnt(0): that
    // wait must not sit between the gather loads)
    int rr[MAXW];
#pragma unroll
    for (int e = 0; e < MAXW; e++) rr[e] = rt[e < width ? e : width - 1];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    uint32_t v[MAXW];
#pragma unroll
    for (int e = 0; e < MAXW; e++) {
        const int r = __builtin_amdgcn_readfirstlane(rr[e]); // the same for every lane: a scalar
        const uint16_t *rowp = Pb + (size_t)r * RS;
        asm volatile("global_load_ushort %0, %1, %2" : "=v"(v[e]) : "v"(voff), "s"(rowp));
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // (loads issued by inline assembly are not counted by the compiler)
    if (live) {
#pragma unroll
        for (int e = 0; e < MAXW; e++)
            if (e < width) t[e] = (uint16_t)v[e];
    }
}

// ---- K8, grouped form (round 5; default).  k_assemble_fields gathers the records of the OPENED parties 64 entries of I at a time:
// 64 scattered columns, i.e. about 20 different 128-byte lines per row and load instruction, where a window of 64 adjacent columns
// is ONE line -- three quarters of the launch's line requests for a tenth of its bytes (232 rows x 3 chunks x ~20 lines against 203
// rows x 23 windows x 1 line per proof; tools/probe_wire.hip: a dense window gather of this shape runs at the rate of a linear sweep).
// Here every block is (group of fields, aligned window of 64 party columns) and every row costs one line:
//   * fields of unopened parties, as before, but packed into groups of up to 80 rows (the seven small fields of 3 .. 15 rows are
//     one 63-row block per window instead of seven short ones: as many loads in flight per wave as the big fields have);
//   * fields of opened parties: the window's 64 columns gathered densely, and the records of the window's opened parties (6.6 on
//     average; the host's Fiat-Shamir round leaves them sorted with their positions in I, SEL_OSORT / SEL_OPOS) written from the
//     LDS tile; a window without an opened party does nothing.
// Same image bytes as the per-field kernel of rounds 1-4 (k_assemble_fields, removed in round 6).
// A block = (group, window).  The window's 64 columns of the group's rows are gathered DENSE with 8-byte loads: lane = (row slot
// lane >> 4, column quad lane & 15), so one instruction brings four whole 128-byte lines (four rows) where a 2-byte-per-lane gather
// brings one -- the launch is bound by the number of load instructions, not by bytes (measured: the same dense gather with 2-byte
// loads made the launch 25 % SLOWER than the scattered gathers it replaced, profiles/r05_wire.txt).  Only the columns this block
// writes out reach the LDS tile, already compacted: a column's RANK among them (unopened kind: its position among the window's
// unopened parties; opened kind: among the window's opened ones) comes from a 64-entry table the block builds first, and the tile
// position of (rank, row) is the image order of the write-out -- per field [party][width] blocks for the unopened kind (each field
// is then one linear run), [party][rows] for the opened kind (one record set per opened party).
template <int NQ> // NQ >= ceil(nrows / 4)
__device__ __forceinline__ void asm_group_block(const AssembleArgs &a, const AsmGroup &g, const int b, const int w, uint16_t *tile, int16_t *rank_s,
                                                AsmElem *el_s)
{
    const int lane = threadIdx.x, slot = lane >> 4, cq = lane & 15;
    const uint16_t *orow = a.opened + (size_t)b * a.sel_stride;
    const uint16_t *Pb = a.P + (size_t)b * a.proof_stride;
    const int16_t *rt = a.rowtab + g.rowtab_off;
    uint16_t *img16 = reinterpret_cast<uint16_t *>(a.proof + (size_t)b * a.image_stride);
    const int nrows = g.nrows;
    const int lo = min(64 * w, NPARTY), hi = min(64 * (w + 1), NPARTY);
    const int i0 = orow[SEL_WIN + w], cnt = (int)orow[SEL_WIN + w + 1] - i0; // unopened parties of the window: records [i0, i0 + cnt)
    const int k0 = lo - i0, nk = (hi - lo) - cnt;                               // its opened parties: entries [k0, k0 + nk) of the sorted list
    const int nsel = g.sel ? cnt : nk;
    if (nsel <= 0) return;
    // rank table: rank_s[column] = position among the columns this block writes out, -1 for the others
    rank_s[lane] = -1;
    int my_pos = 0;
    int my_col = 0;
    if (lane < nsel) {
        if (g.sel) my_col = (int)a.rest[(size_t)b * a.sel_stride + i0 + lane] - lo;
        else { my_col = (int)orow[SEL_OSORT + k0 + lane] - lo; my_pos = orow[SEL_OPOS + k0 + lane]; }
    }
    const AsmElem *el = a.elems + g.elem_off;
    el_s[lane] = el[lane]; // the group's element table (padded to 128 entries), for per-lane row indices
    el_s[lane + 64] = el[lane + 64];
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    if (lane < nsel) rank_s[my_col] = (int16_t)lane;
    // the gather: all loads first
    int rr[NQ];
#pragma unroll
    for (int q = 0; q < NQ; q++) rr[q] = rt[min(4 * q + slot, nrows - 1)];
    uint2 v[NQ];
    const uint16_t *src = Pb + NSEC + lo + 4 * cq;
#pragma unroll
    for (int q = 0; q < NQ; q++) v[q] = *reinterpret_cast<const uint2 *>(src + (size_t)rr[q] * RS);
    __builtin_amdgcn_s_waitcnt(0xC07F); // lgkmcnt(0): the rank table is written
    __builtin_amdgcn_wave_barrier();
    int rk[4];
#pragma unroll
    for (int m = 0; m < 4; m++) rk[m] = rank_s[4 * cq + m];
#pragma unroll
    for (int q = 0; q < NQ; q++) {
        const int r = 4 * q + slot;
        if (r < nrows) {
            const uint32_t x[4] = {v[q].x & 0xFFFFu, v[q].x >> 16, v[q].y & 0xFFFFu, v[q].y >> 16};
            if (g.sel) {
                const AsmElem e = el_s[r];
#pragma unroll
                for (int m = 0; m < 4; m++)
                    if (rk[m] >= 0) tile[(int)e.tile + rk[m] * (int)e.width] = (uint16_t)x[m];
            } else {
#pragma unroll
                for (int m = 0; m < 4; m++)
                    if (rk[m] >= 0) tile[rk[m] * nrows + r] = (uint16_t)x[m];
            }
        }
    }
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    if (g.sel) {
        for (int s = 0; s < g.nsub; s++) { // one contiguous run per field; its start is only 2-byte aligned in general
            const int width = g.sub_width[s];
            uint16_t *out = img16 + g.sub_off[s] / 2 + (size_t)i0 * width;
            const uint16_t *srct = tile + 64 * (int)g.sub_col[s];
            const int n16 = cnt * width;
            const int head = (int)((reinterpret_cast<uintptr_t>(out) >> 1) & 1);
            const int body = (n16 - head) >> 1;
            if (lane == 0 && head) out[0] = srct[0];
            uint32_t *out32 = reinterpret_cast<uint32_t *>(out + head);
            for (int q = lane; q < body; q += 64) out32[q] = (uint32_t)srct[head + 2 * q] | ((uint32_t)srct[head + 2 * q + 1] << 16);
            if (lane == 1 && head + 2 * body < n16) out[n16 - 1] = srct[n16 - 1];
        }
    } else {
        const AsmElem e0 = el_s[lane], e1 = el_s[lane + 64];
        for (int j = 0; j < nk; j++) {
            const int i = __builtin_amdgcn_readlane(my_pos, j);
            if (lane < nrows) img16[e0.dst + i * (int)e0.width] = tile[j * nrows + lane];
            if (NQ > 16 && lane + 64 < nrows) img16[e1.dst + i * (int)e1.width] = tile[j * nrows + lane + 64];
        }
    }
}

__global__ __launch_bounds__(64) void k_assemble_groups(AssembleArgs a, uint32_t off_tcomm, uint32_t off_comm, uint32_t off_I, int blocks_per_proof, int nproofs)
{
    const int vid = xcd_virtual_id(); // the windows of a group read adjacent lines of the same rows: one XCD, one L2
    const int b = vid / blocks_per_proof, bx = vid - b * blocks_per_proof;
    if (b >= nproofs) return;
    const int ngb = a.ngroups * NWIN;
    if (bx >= ngb) { // Tcomm / comm of the unopened parties and the list I itself
        // one thread per (unopened party, table, 16-byte half of its digest): 16-byte loads (the digests are 32-byte aligned in both
        // tables), stores as wide as the image's field offsets allow (the launcher's img_align).  Rounds 1-6a moved two bytes per
        // thread and table -- 326 workgroups per proof whose dispatch, not their bytes, cost 41 us per 276 proofs.
        const int q = (bx - ngb) * 64 + threadIdx.x;
        uint8_t *img = a.proof + (size_t)b * a.image_stride;
        if (q < NREST * 4) {
            const int i = q >> 2, tab = (q >> 1) & 1, half = q & 1;
            const size_t src = ((size_t)b * NPARTY + a.rest[(size_t)b * a.sel_stride + i]) * 32 + 16 * half;
            const uint4 v = *reinterpret_cast<const uint4 *>((tab ? a.dig2 : a.dig1) + src);
            uint8_t *dst = img + (tab ? off_comm : off_tcomm) + (size_t)i * 32 + 16 * half;
            if (a.img_align >= 16) *reinterpret_cast<uint4 *>(dst) = v;
            else if (a.img_align >= 8) { // (every Kyber parameter set: the digest fields start 8 bytes off a 16-byte boundary)
                reinterpret_cast<uint2 *>(dst)[0] = make_uint2(v.x, v.y);
                reinterpret_cast<uint2 *>(dst)[1] = make_uint2(v.z, v.w);
            } else if (a.img_align >= 4) { // (K = 3: the view-commitment field starts 4 bytes off)
                const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int k = 0; k < 4; k++) reinterpret_cast<uint32_t *>(dst)[k] = w[k];
            } else {
                const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int k = 0; k < 8; k++) reinterpret_cast<uint16_t *>(dst)[k] = (uint16_t)(w[k >> 1] >> (16 * (k & 1)));
            }
        }
        if (q < NOPEN) reinterpret_cast<uint16_t *>(img + off_I)[q] = a.opened[(size_t)b * a.sel_stride + q];
        return;
    }
    __shared__ __attribute__((aligned(16))) uint16_t tile[ASM_TILE];
    __shared__ int16_t rank_s[64];
    __shared__ AsmElem el_s[128];
    const int gi = bx / NWIN, w = bx - gi * NWIN;
    const AsmGroup &g = a.groups[gi];
    if (g.nrows <= 16) asm_group_block<4>(a, g, b, w, tile, rank_s, el_s);
    else if (g.nrows <= 64) asm_group_block<16>(a, g, b, w, tile, rank_s, el_s);
    else asm_group_block<20>(a, g, b, w, tile, rank_s, el_s);
}

// The SMALL copies of a step between HBM and the library's own page-locked (device-mapped) host buffers -- challenge vectors and
// opened lists in, key records, opened lists of the images and fail masks out; 0.5 .. 700 KB -- as a kernel instead of
// hipMemcpyAsync: on this runtime every copy is a blit kernel with a ~12 us dependency gap in front of it
// (profiles/r04_trace_gaps.txt: ten copies per step, six of them these), a kernel behind a kernel starts 0.2 us later.
// rows x row_bytes (4-byte multiples, 4-byte aligned) with separate strides: the 2-D form serves the image field gather.
__global__ __launch_bounds__(256) void k_copy_small(const uint8_t *__restrict__ src, size_t src_stride, uint8_t *__restrict__ dst, size_t dst_stride,
                                                    uint32_t row_words, uint32_t nrows)
{
    const size_t total = (size_t)row_words * nrows;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const uint32_t r = (uint32_t)(i / row_words), w = (uint32_t)(i - (size_t)r * row_words);
        reinterpret_cast<uint32_t *>(dst + (size_t)r * dst_stride)[w] = reinterpret_cast<const uint32_t *>(src + (size_t)r * src_stride)[w];
    }
}

// plain strided row copy (kernel-level ABI helpers): dst[r][0..count) = src[r][0..count)
__global__ __launch_bounds__(256) void k_rows_copy(const uint16_t *__restrict__ src, size_t src_stride,
                                                  uint16_t *__restrict__ dst, size_t dst_stride, int count)
{
    const int r = blockIdx.y;
    for (int x = blockIdx.x * 256 + threadIdx.x; x < count; x += gridDim.x * 256)
        dst[(size_t)r * dst_stride + x] = src[(size_t)r * src_stride + x];
}

// =========================================================================
// host-side launchers
// =========================================================================
template <int PW, int NR>
static int launch_hash_t(const HashArgs &a, int ngroups, hipStream_t st)
{
    dim3 grid((a.lanes_per_group + 63) / 64, ngroups);
    const long waves = (long)grid.x * grid.y;
    // LDS-DMA staging needs 16-byte aligned 128-byte row segments per wave and readable row padding up to the last
    // wave's 64th lane (true for the row matrix: RS = 1728 = 256 + 23 * 64)
    const bool dma_ok = !a.lane_map && a.row_stride % 8 == 0 && a.group_stride % 8 == 0 && a.col_off % 8 == 0 &&
                        (reinterpret_cast<uintptr_t>(a.rows) & 15) == 0 && a.col_off + (int)grid.x * 64 <= a.row_stride &&
                        (!PW || (reinterpret_cast<uintptr_t>(a.prefix) & 15) == 0);
    if (dma_ok) {
        // up to ~2 waves per SIMD the next block's DMA runs under this block's permutation (two buffers, 18 KiB per wave);
        // beyond that one buffer (4 waves per SIMD fit) and occupancy hides the landing
        const dim3 grid1((unsigned)waves);
        if (waves <= 2 * 1024 + 256) hipLaunchKernelGGL((k_commit_hash_dma<PW, NR, 2>), grid1, dim3(64), 0, st, a);
        else hipLaunchKernelGGL((k_commit_hash_dma<PW, NR, 1>), grid1, dim3(64), 0, st, a);
        return 1;
    }
    // fewer than ~3 waves per SIMD (1024 SIMDs): nothing else hides the row loads -> pipelined variant
    if (waves < 3 * 1024) hipLaunchKernelGGL((k_commit_hash<PW, NR, true>), grid, dim3(64), 0, st, a);
    else hipLaunchKernelGGL((k_commit_hash<PW, NR, false>), grid, dim3(64), 0, st, a);
    return 0;
}

hipError_t launch_commit_hash(const HashArgs &a, int ngroups, int K, bool view, hipStream_t st, int *variant)
{
    // rows hashed per party: Tcomm 2(K+M); view (6+8 eta1)K + 2M   (M = 71+2K)
    int v;
    if (!view) {
        if (K == 2) v = launch_hash_t<0, 154>(a, ngroups, st);
        else if (K == 3) v = launch_hash_t<0, 160>(a, ngroups, st);
        else v = launch_hash_t<0, 166>(a, ngroups, st);
    } else {
        if (K == 2) v = launch_hash_t<16, 210>(a, ngroups, st);
        else if (K == 3) v = launch_hash_t<16, 220>(a, ngroups, st);
        else v = launch_hash_t<16, 246>(a, ngroups, st);
    }
    if (variant) *variant = v;
    return hipGetLastError();
}

hipError_t launch_sha3_msgs(const uint8_t *in, size_t in_stride, int len, uint8_t *out, size_t out_stride,
                            int outlen, int n, int domain, hipStream_t st)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_sha3_msgs, dim3((n + 63) / 64), dim3(64), 0, st, in, in_stride, len, out, out_stride, outlen, n, domain);
    return hipGetLastError();
}

hipError_t launch_sha3_msgs_pair(const uint8_t *in, size_t in_stride, int len, uint8_t *out, size_t out_stride, int outlen, int n,
                                 int domain, hipStream_t st)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_sha3_msgs_pair, dim3((n + 31) / 32), dim3(64), 0, st, in, in_stride, len, out, out_stride, outlen, n, domain);
    return hipGetLastError();
}

bool copy_small_ok(const void *src, size_t src_stride, const void *dst, size_t dst_stride, size_t row_bytes)
{
    return row_bytes % 4 == 0 && src_stride % 4 == 0 && dst_stride % 4 == 0 && (reinterpret_cast<uintptr_t>(src) & 3) == 0 &&
           (reinterpret_cast<uintptr_t>(dst) & 3) == 0 && row_bytes / 4 < 0xFFFFFFFFull;
}
hipError_t launch_copy_small(const void *src, size_t src_stride, void *dst, size_t dst_stride, size_t row_bytes, size_t nrows, hipStream_t st)
{
    if (!row_bytes || !nrows) return hipSuccess;
    if (!copy_small_ok(src, src_stride, dst, dst_stride, row_bytes) || nrows > 0xFFFFFFFFull) return hipErrorInvalidValue;
    if (nrows > 1 && src_stride == row_bytes && dst_stride == row_bytes && row_bytes * nrows / 4 < 0xFFFFFFFFull) { row_bytes *= nrows; nrows = 1; } // contiguous
    const size_t words = row_bytes / 4 * nrows;
    const unsigned nwg = (unsigned)std::min<size_t>((words + 255) / 256, 256);
    hipLaunchKernelGGL(k_copy_small, dim3(nwg), dim3(256), 0, st, reinterpret_cast<const uint8_t *>(src), src_stride, reinterpret_cast<uint8_t *>(dst),
                       dst_stride, (uint32_t)(row_bytes / 4), (uint32_t)nrows);
    return hipGetLastError();
}

hipError_t launch_rows_copy(const uint16_t *src, size_t src_stride, uint16_t *dst, size_t dst_stride, int count,
                            int nrows, hipStream_t st)
{
    if (nrows <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_rows_copy, dim3((count + 255) / 256, nrows), dim3(256), 0, st, src, src_stride, dst, dst_stride, count);
    return hipGetLastError();
}

hipError_t launch_prover_pre(const uint8_t *tape, size_t tape_stride, uint16_t *P, size_t proof_stride, int row_f, int M,
                             int slice0_off, const int16_t *fresh_rows, int slice_begin, int slice_end, bool expand_f,
                             int witness_mode, const int16_t *se, size_t se_stride, const RowMap &rm, int eta1, int nproofs,
                             hipStream_t st, const KeygenFront *kg, const TapeSegs *segs)
{
    PreArgs a{};
    if (segs) a.segs = *segs;
    a.tape = tape; a.tape_stride = tape_stride; a.P = P; a.proof_stride = proof_stride;
    a.row_f = row_f; a.M = M; a.nproofs = nproofs;
    a.slice0_off = slice0_off; a.slice_begin = slice_begin; a.slice_end = slice_end; a.fresh_rows = fresh_rows;
    a.witness_mode = witness_mode;
    a.se = se; a.se_stride = se_stride; a.rm = rm; a.eta1 = eta1;
    a.K = rm.K;
    const int per = 32; // sponges per block (roles G, N, A: the lane-pair sponge)
    a.nbA = expand_f ? (M * nproofs + per - 1) / per : 0;
    a.nbB = 3 * ((slice_end - slice_begin + PRE_SLICES - 1) / PRE_SLICES) * nproofs;
    const int nbC = witness_mode ? 4 * nproofs : 0;
    if (kg) {
        // the witness secrets (role C) read s and e, which role N of this launch produces: they follow in a launch of their own
        a.kg_seeds = kg->seeds; a.kg_seed_stride = kg->seed_stride; a.kg_A = kg->A; a.kg_A_stride = kg->A_stride; a.kg_se = kg->se; a.kg_se_stride = kg->se_stride;
        a.xof = kg->xof;
        a.nbG = nproofs * rm.K * rm.K; // one block (wave) per matrix entry
        a.nbN = (nproofs * 2 * rm.K + per - 1) / per;
        hipLaunchKernelGGL(k_prover_pre, dim3(a.nbA + a.nbB + a.nbG + a.nbN), dim3(64), 0, st, a);
        if (nbC) {
            a.nbA = a.nbB = a.nbG = a.nbN = 0;
            hipLaunchKernelGGL(k_prover_pre, dim3(nbC), dim3(64), 0, st, a);
        }
        return hipGetLastError();
    }
    const int nb = a.nbA + a.nbB + nbC;
    if (nb <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_prover_pre, dim3(nb), dim3(64), 0, st, a);
    return hipGetLastError();
}

// key generation alone (kosk_stage_prover_inputs): roles G and N of the same kernel
hipError_t launch_keygen(const uint8_t *tape, size_t tape_stride, uint8_t *seeds, size_t seed_stride, int16_t *A, size_t A_stride,
                         int16_t *se, size_t se_stride, int K, int eta1, int n, hipStream_t st, XofGuard xof)
{
    PreArgs a{};
    a.xof = xof;
    a.tape = tape; a.tape_stride = tape_stride; a.nproofs = n; a.eta1 = eta1; a.K = K;
    a.kg_seeds = seeds; a.kg_seed_stride = seed_stride; a.kg_A = A; a.kg_A_stride = A_stride; a.kg_se = se; a.kg_se_stride = se_stride;
    const int per = 32;
    a.nbG = n * K * K; // one block (wave) per matrix entry
    a.nbN = (n * 2 * K + per - 1) / per;
    hipLaunchKernelGGL(k_prover_pre, dim3(a.nbG + a.nbN), dim3(64), 0, st, a);
    return hipGetLastError();
}

hipError_t launch_ntt(const NttArgs &args, hipStream_t st)
{
    if (args.npoly <= 0) return hipSuccess;
    NttArgs a = args;
    a.npg_magic = ntt_npg_magic(a.npg);
    // polynomials back to back in both buffers (kosk_ntt256_batch; the verifier's 140 x n secrets): no group / offset arithmetic
    const bool plain = !a.src_off && !a.dst_off && !a.cmp_fail && (a.npg >= a.npoly || (a.in_gstride == (size_t)a.npg * 256 && a.out_gstride == (size_t)a.npg * 256)) &&
                       (reinterpret_cast<uintptr_t>(a.in) & 15) == 0 && (reinterpret_cast<uintptr_t>(a.out) & 15) == 0;
    const dim3 grid((a.npoly + NTT_PPB - 1) / NTT_PPB);
    if (plain) hipLaunchKernelGGL(k_ntt256<true>, grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL(k_ntt256<false>, grid, dim3(256), 0, st, a);
    return hipGetLastError();
}

hipError_t launch_relation_ntt(const NttArgs &na, const int16_t *A, size_t A_stride, uint16_t *P, size_t proof_stride, const RowMap &rm,
                               int nproofs, hipStream_t st)
{
    if (na.npg > NTT_PPB) return hipErrorInvalidValue;
    NttArgs nb = na;
    nb.npg_magic = ntt_npg_magic(nb.npg);
    hipLaunchKernelGGL(k_relation_ntt, dim3(nproofs), dim3(256), 0, st, nb, A, A_stride, P, proof_stride, rm);
    return hipGetLastError();
}

hipError_t launch_matvec_ntt(const int16_t *A, size_t A_stride, uint16_t *P, size_t proof_stride, int v_row0, int row0, int K,
                             int nproofs, hipStream_t st)
{
    hipLaunchKernelGGL(k_matvec_ntt, dim3(K, nproofs), dim3(128), 0, st, A, A_stride, P, proof_stride, v_row0, row0, K);
    return hipGetLastError();
}

hipError_t launch_rows_to_limbs(const LimbArgs &a, hipStream_t st)
{
    if (a.RT <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_rows_to_limbs, dim3((a.KS + 3) / 4, a.RT), dim3(256), 0, st, a);
    return hipGetLastError();
}

// the table-product kernel: shared table, KS == 7 (407-wide inputs) or 13 (813-wide: recon_secrets_2ddeg), aligned u16 rows
static bool table_gemm_ok(const GemmArgs &a)
{
    return a.Afrag && !a.grouped && !a.B && (a.KS == 7 || a.KS == 13) && a.M % 32 == 0 && a.c_gdiv <= 1 && a.src_koff % 8 == 0 && a.src_rstride % 8 == 0 &&
           a.src_gstride % 8 == 0 && (reinterpret_cast<uintptr_t>(a.src) & 15) == 0 && a.c_off % 4 == 0 && a.c_rstride % 4 == 0 &&
           a.c_gstride % 4 == 0 && (reinterpret_cast<uintptr_t>(a.C) & 7) == 0;
}

bool table_gemm_usable(const GemmArgs &a) { return table_gemm_ok(a); }

hipError_t launch_table_gemm(const GemmArgs &a, uint16_t *sink, hipStream_t st)
{
    const int ntot = a.npg * a.ngroups;
    if (ntot <= 0) return hipSuccess;
    static const int ncu = [] {
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) { (void)hipGetLastError(); cus = 256; }
        return cus;
    }();
    const int nblk = (ntot + 47) / 48, nchunks = a.M / 16; // 48 rows per block
    (void)sink;
    // eight consecutive outputs per lane and 16-byte stores where the output rows allow it
    const int ws = a.c_off % 8 == 0 && a.c_rstride % 8 == 0 && a.c_gstride % 8 == 0 && (reinterpret_cast<uintptr_t>(a.C) & 15) == 0;
    if (a.KS == 7) {
        // persistent workgroups (k_table_gemm_p): one per CU, the (row block, table chunk) units dealt evenly.  (The one-block-per-workgroup
        // kernel of rounds 2-4, its 64-row variant and the 8-byte-store epilogue are gone since round 6: profiles/r04_sweeps.txt)
        const long total = (long)nblk * nchunks;
        // at least eight chunks per workgroup where the product is that large (a chunk per wave), never more workgroups than CUs
        long nwg = total / 8 < ncu ? total / 8 : ncu;
        if (nwg < 1) nwg = 1;
        const uint32_t magic = ntt_npg_magic(a.npg);
        if (a.src_canonical) hipLaunchKernelGGL((k_table_gemm_p<7, 3, true>), dim3((unsigned)nwg), dim3(512), 0, st, a, nchunks, nblk, ws, magic);
        else hipLaunchKernelGGL((k_table_gemm_p<7, 3, false>), dim3((unsigned)nwg), dim3(512), 0, st, a, nchunks, nblk, ws, magic);
        return hipGetLastError();
    }
    // 13 k-steps (recon_secrets_2ddeg, 24 rows per proof): few data rows, so the table is split over several workgroups per row block
    int msplit = nblk >= 160 ? 1 : (256 + nblk - 1) / nblk;
    if (msplit > nchunks) msplit = nchunks;
    const int cpb = (nchunks + msplit - 1) / msplit;
    msplit = (nchunks + cpb - 1) / cpb;
    const dim3 grid((unsigned)((nblk * msplit + 7) / 8 * 8));
    hipLaunchKernelGGL((k_table_gemm<13, 1, 3>), grid, dim3(512), 0, st, a, nchunks, cpb, nblk, msplit, ws);
    return hipGetLastError();
}

hipError_t launch_gemm(const GemmArgs &a, hipStream_t st)
{
    const int ntot = a.grouped ? a.npg : a.npg * a.ngroups;
    if (ntot <= 0) return hipSuccess;
    dim3 grid(a.Mpad / GM_TM, (ntot + GM_TN - 1) / GM_TN, a.grouped ? a.ngroups : 1);
    if (a.B) hipLaunchKernelGGL(k_gemm_modq<true>, grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL(k_gemm_modq<false>, grid, dim3(256), 0, st, a);
    return hipGetLastError();
}

hipError_t launch_lincomb_stream(const uint16_t *P, size_t proof_stride, const RowMap &rm, const uint8_t *coef, uint16_t *C,
                                 const int16_t *lin_rows, int J, int nproofs, hipStream_t st)
{
    if (J > 16 * LS_JT || rm.M <= 64 || rm.M > 80) return hipErrorInvalidValue; // (every Kyber parameter set: J = 74..78, M = 75..79)
    // the (group, point block) units dealt evenly over two persistent workgroups per CU
    static const int ncu = [] {
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) { (void)hipGetLastError(); cus = 256; }
        return cus;
    }();
    const int total = (NPTS + 127) / 128 * 2 * nproofs;
    const int nwg = total < 2 * ncu ? total : 2 * ncu;
    hipLaunchKernelGGL(k_lincomb_stream, dim3(nwg), dim3(256), 0, st, P, proof_stride, rm.f, rm.tf, rm.M, coef, 2 * nproofs * 8, C, lin_rows, J,
                       rm.K, rm.s, rm.e, rm.sr, rm.er, 2 * nproofs);
    return hipGetLastError();
}
hipError_t launch_coef_limbs(const uint16_t *alpha, int J, int M, uint8_t *B, int nproofs, hipStream_t st)
{
    hipLaunchKernelGGL(k_coef_limbs, dim3(nproofs), dim3(1024), 0, st, alpha, J, M, B, 2 * nproofs * 8);
    return hipGetLastError();
}

hipError_t launch_pow_table(const uint16_t *alpha, int J, int M, int32_t *pwT, int nproofs, hipStream_t st)
{
    hipLaunchKernelGGL(k_pow_table, dim3(nproofs), dim3(128), 0, st, alpha, J, M, pwT);
    return hipGetLastError();
}

hipError_t launch_lincomb(const LincombArgs &a, int nproofs, hipStream_t st)
{
    LincombArgs b = a;
    b.nxb = (a.ncols + 255) / 256;
    b.njc = (a.J + LC_JC - 1) / LC_JC;
    b.ngroups = nproofs * 2;
    hipLaunchKernelGGL(k_lincomb, dim3((unsigned)((b.nxb * b.njc * b.ngroups + 7) / 8 * 8)), dim3(256), 0, st, b);
    return hipGetLastError();
}

hipError_t launch_post_gates(uint16_t *P, size_t proof_stride, const RowMap &rm, int nproofs, hipStream_t st)
{
    const dim3 g(1, nproofs, 2 * rm.K); // 214 threads x 8 points per gate chain
    if (rm.E == 7) hipLaunchKernelGGL(k_post_gates<7>, g, dim3(256), 0, st, P, proof_stride, rm);
    else hipLaunchKernelGGL(k_post_gates<5>, g, dim3(256), 0, st, P, proof_stride, rm);
    return hipGetLastError();
}
hipError_t launch_copy_tails(uint16_t *P, size_t proof_stride, const RowMap &rm, int nproofs, hipStream_t st)
{
    hipLaunchKernelGGL(k_copy_tails, dim3(rm.K, nproofs), dim3(192), 0, st, P, proof_stride, rm);
    return hipGetLastError();
}
hipError_t launch_post_relation(uint16_t *P, size_t proof_stride, const RowMap &rm, int nproofs, hipStream_t st)
{
    hipLaunchKernelGGL(k_post_relation, dim3((NPTS + 255) / 256, nproofs), dim3(256), 0, st, P, proof_stride, rm);
    return hipGetLastError();
}

hipError_t launch_assemble(const AssembleArgs &a, size_t off_tcomm, size_t off_comm, size_t off_I, int nproofs, hipStream_t st)
{
    if (!a.groups || a.ngroups <= 0) return hipErrorInvalidValue;
    // widest store the image's digest fields allow (field offsets, image stride and base all multiples of it)
    auto ok = [&](size_t m) { return off_tcomm % m == 0 && off_comm % m == 0 && a.image_stride % m == 0 && (reinterpret_cast<uintptr_t>(a.proof) % m) == 0; };
    const_cast<AssembleArgs &>(a).img_align = ok(16) ? 16 : ok(8) ? 8 : ok(4) ? 4 : 2;
    const int bpp = a.ngroups * NWIN + (NREST * 4 + 63) / 64;
    const long nwg = ((long)bpp * nproofs + 7) / 8 * 8;
    hipLaunchKernelGGL(k_assemble_groups, dim3((unsigned)nwg), dim3(64), 0, st, a, (uint32_t)off_tcomm, (uint32_t)off_comm, (uint32_t)off_I, bpp, nproofs);
    return hipGetLastError();
}

} // namespace kosk
