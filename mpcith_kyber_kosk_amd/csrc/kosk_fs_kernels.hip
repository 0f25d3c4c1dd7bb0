// Fiat-Shamir aggregation on the GPU (round 6): the two 343-permutation chains per proof and role
//
//   h1 = sha3_256(Tcomm[0..1454))  -> alpha = BE16(SHAKE256-PRF(h1, 1)) % q          mlwe_prover.cpp:130-153, mlwe_verifier.cpp:37-65
//   ch = sha3_256(ch_seeds[0..1454)) -> I = opened set ("+inc, rescan" probing)       mlwe_prover.cpp:445-474, mlwe_verifier.cpp:634-683
//
// hashed where the commitment kernels wrote the tables, in HBM: nothing but 32 + 300 bytes per proof would have to reach the host,
// and with the challenge vectors / opened lists consumed on the device nothing does -- the four 46.5 KB-per-proof device-to-host
// copies of a step, the host's four hashing rounds and the four host round trips disappear.
// One wave per proof; the sponge is csrc/kosk_fs_dev.hpp (one state per wave, a word per lane).
//   k_fs_chain<FS_DIGEST>  sha3_256 of n long messages (kernel-level entry point kosk_sha3_256_batch_wave; tests, bench)
//   k_fs_chain<FS_ALPHA>   the challenge vector of every proof, [n][80] u16
//   k_fs_chain<FS_OPENED>  prover: I, its ascending complement, the window boundaries and the sorted opened list (what
//                          fs_opened_batch of kosk_host.cpp writes into a row of the opened-list table)
//   k_fs_chain<FS_CHECK>   verifier: I' recomputed and compared with the proof's own list, fail bit FB_OPENED_SET
// kyber/fips202.c:461-485, :723-734 (shake256), :745-754 (sha3_256); kyber/symmetric-shake.c:41-51 (kyber_shake256_prf).
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>

#include "kosk_device.hpp"
#include "kosk_fs_dev.hpp"

namespace kosk {

namespace {

constexpr int FS_PF = 4; // message blocks in flight per wave (8 bytes per lane each)

__device__ __forceinline__ uint2 fs_load_word(const uint8_t *p)
{
    return *reinterpret_cast<const uint2 *>(p);
}

// the last (partial) block's word at byte offset 8 w of the remaining `rem` bytes, padded: dom at byte rem, 0x80 at byte 135
__device__ __forceinline__ uint2 fs_last_word(const uint8_t *tail, int rem, int w, uint32_t dom)
{
    uint32_t lo = 0, hi = 0;
    const int o = 8 * w;
    if (o + 8 <= rem) {
        const uint2 v = fs_load_word(tail + o);
        lo = v.x; hi = v.y;
    } else if (o < rem) {
        for (int i = 0; i < rem - o; i++) {
            const uint32_t byte = tail[o + i];
            if (i < 4) lo |= byte << (8 * i);
            else hi |= byte << (8 * (i - 4));
        }
    }
    if ((rem >> 3) == w) {
        const int sh = 8 * (rem & 7);
        if (sh < 32) lo ^= dom << sh;
        else hi ^= dom << (sh - 32);
    }
    if (w == 16) hi ^= 0x80000000u;
    return make_uint2(lo, hi);
}

template <int MODE, class SP>
__global__ __launch_bounds__(64) void k_fs_chain(FsArgs A)
{
    __shared__ __align__(16) uint32_t xw[FSW_WORDS];
    __shared__ __align__(16) uint32_t st[64];       // the state's words, to be re-interleaved by other lanes
    __shared__ __align__(16) uint8_t sq[3 * 136];   // squeezed PRF bytes
    __shared__ uint16_t pos[MODE >= FS_OPENED ? NPARTY : 1];
    __shared__ uint16_t il[MODE >= FS_OPENED ? NOPEN + 2 : 1];

    const int lane = threadIdx.x, b = blockIdx.x;
    __builtin_amdgcn_s_setprio(3); // a chain is latency, not throughput: its wave issues first wherever it shares a SIMD
    SP sp;
    sp.setup(lane, xw);
    const int word = sp.word();
    const uint32_t half = sp.half();

    // ---- sha3_256 of the table
    const uint8_t *src = A.in + (size_t)b * A.in_stride;
    const int nfull = A.len / 136, rem = A.len - nfull * 136;
    const bool ld = word < 17;
    const uint8_t *mine = src + 8 * (ld ? word : 0);
    uint32_t a = 0;
    uint2 pf[FS_PF];
#pragma unroll
    for (int j = 0; j < FS_PF; j++) pf[j] = (ld && j < nfull) ? fs_load_word(mine + (size_t)136 * j) : make_uint2(0, 0);
    for (int blk = 0; blk < nfull; blk += FS_PF) {
#pragma unroll
        for (int j = 0; j < FS_PF; j++) {
            if (blk + j < nfull) { // (uniform)
                const uint2 m = pf[j];
                const int nb = blk + j + FS_PF;
                pf[j] = (ld && nb < nfull) ? fs_load_word(mine + (size_t)136 * nb) : make_uint2(0, 0);
                a ^= fs_deinterleave_half(m.x, m.y, half); // (lanes beyond word 16 loaded zeros)
                sp.permute(a);
            }
        }
    }
    {
        const uint2 m = ld ? fs_last_word(src + (size_t)136 * nfull, rem, word, 0x06u) : make_uint2(0, 0);
        a ^= fs_deinterleave_half(m.x, m.y, half);
        sp.permute(a);
    }
    if (A.out_digest) { // words 0..3 = lanes 6 x (+ 32)
        st[lane] = a;
        __builtin_amdgcn_wave_barrier();
        if (lane < 4) {
            uint32_t lo, hi;
            fs_interleave(st[SP::lane_of(lane, 0, 0)], st[SP::lane_of(lane, 0, 1)], lo, hi);
            *reinterpret_cast<uint2 *>(A.out_digest + (size_t)b * 32 + 8 * lane) = make_uint2(lo, hi);
        }
        __builtin_amdgcn_wave_barrier();
    }
    if constexpr (MODE == FS_DIGEST) return;

    // ---- SHAKE256-PRF(digest, nonce 1): the digest's words are the new block's words 0..3 as they stand (still interleaved)
    {
        uint32_t lo = 0, hi = 0;
        if (word == 4) lo = 0x1F01u;      // nonce byte 1, then the SHAKE domain byte (33 bytes absorbed)
        if (word == 16) hi = 0x80000000u; // last byte of the 136-byte rate
        const uint32_t pad = fs_deinterleave_half(lo, hi, half);
        a = (word < 4 ? a : 0u) ^ pad;
    }
    constexpr int NSQ = MODE == FS_ALPHA ? 2 : 3;
#pragma unroll 1
    for (int s = 0; s < NSQ; s++) {
        sp.permute(a);
        st[lane] = a;
        __builtin_amdgcn_wave_barrier();
        if (lane < 17) {
            const int x = lane % 5, y = lane / 5;
            uint32_t lo, hi;
            fs_interleave(st[SP::lane_of(x, y, 0)], st[SP::lane_of(x, y, 1)], lo, hi);
            *reinterpret_cast<uint2 *>(sq + 136 * s + 8 * lane) = make_uint2(lo, hi);
        }
        __builtin_amdgcn_wave_barrier();
    }

    if constexpr (MODE == FS_ALPHA) {
        // alpha_i = BE16 % q, i < J (mlwe_prover.cpp:137-142); the entries behind J stay zero
        for (int i = lane; i < 80; i += 64) {
            const uint32_t v = i < A.J ? (((uint32_t)sq[2 * i] << 8) | sq[2 * i + 1]) % (uint32_t)Q : 0u;
            A.alpha[(size_t)b * A.alpha_stride + i] = (uint16_t)v;
        }
        return;
    } else {
        // ---- the opened set: candidate BE16 % N, then the reference's "+inc, rescan" probing = the first free party at or behind
        // the candidate, cyclically, in list order (mlwe_prover.cpp:459-474; kosk_host.cpp opened_from_ch)
        uint32_t c[3];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const int i = lane + 64 * k;
            c[k] = i < NOPEN ? (((uint32_t)sq[2 * i] << 8) | sq[2 * i + 1]) % (uint32_t)NPARTY : 0u;
        }
        for (int p = lane; p < NPARTY; p += 64) pos[p] = 0xFFFF;
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const int cnt = k < 2 ? 64 : NOPEN - 128;
            for (int j = 0; j < cnt; j++) {
                uint32_t v = (uint32_t)__builtin_amdgcn_readlane((int)c[k], j);
                while ((uint32_t)__builtin_amdgcn_readfirstlane((int)pos[v]) != 0xFFFFu) v = v + 1 == (uint32_t)NPARTY ? 0u : v + 1;
                if (lane == 0) {
                    pos[v] = (uint16_t)(64 * k + j);
                    il[64 * k + j] = (uint16_t)v;
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
        if constexpr (MODE == FS_CHECK) {
            // I' == I of the proof image (mlwe_verifier.cpp:678-683)
            const uint16_t *given = reinterpret_cast<const uint16_t *>(A.proof + (size_t)b * A.image_stride + A.off_I);
            bool bad = false;
            for (int i = lane; i < NOPEN; i += 64) bad |= il[i] != given[i];
            if (__builtin_amdgcn_ballot_w64(bad) != 0 && lane == 0) atomicOr(A.fail + b, 1u << FB_OPENED_SET);
            return;
        } else {
            uint16_t *Ib = A.I + (size_t)b * A.sel_stride, *rb = A.rest + (size_t)b * A.sel_stride;
            for (int i = lane; i < NOPEN; i += 64) Ib[i] = il[i];
            // the ascending complement, the number of unopened parties below every multiple of 64 (window boundaries), and the opened
            // parties ascending with their positions in I (kosk_params.hpp: SEL_WIN, SEL_OSORT, SEL_OPOS)
            uint32_t nrest = 0, nopen = 0;
            for (int w = 0; w < NWIN; w++) {
                const int p = 64 * w + lane;
                const uint32_t at = p < NPARTY ? pos[p] : 0u;
                const bool in = p < NPARTY, opened = in && at != 0xFFFFu, closed = in && !opened;
                const uint64_t mo = __builtin_amdgcn_ballot_w64(opened), mc = __builtin_amdgcn_ballot_w64(closed);
                const uint64_t below = ((uint64_t)1 << lane) - 1;
                if (lane == 0) Ib[SEL_WIN + w] = (uint16_t)nrest;
                if (closed) rb[nrest + __popcll(mc & below)] = (uint16_t)p;
                if (opened) {
                    const uint32_t k = nopen + __popcll(mo & below);
                    Ib[SEL_OSORT + k] = (uint16_t)p;
                    Ib[SEL_OPOS + k] = (uint16_t)at;
                }
                nrest += __popcll(mc);
                nopen += __popcll(mo);
            }
            if (lane == 0) Ib[SEL_WIN + NWIN] = (uint16_t)nrest;
        }
    }
}

} // namespace

template <class SP>
static void fs_launch(const FsArgs &A, int mode, int n, hipStream_t st)
{
    switch (mode) {
    case FS_DIGEST: hipLaunchKernelGGL((k_fs_chain<FS_DIGEST, SP>), dim3(n), dim3(64), 0, st, A); break;
    case FS_ALPHA: hipLaunchKernelGGL((k_fs_chain<FS_ALPHA, SP>), dim3(n), dim3(64), 0, st, A); break;
    case FS_OPENED: hipLaunchKernelGGL((k_fs_chain<FS_OPENED, SP>), dim3(n), dim3(64), 0, st, A); break;
    default: hipLaunchKernelGGL((k_fs_chain<FS_CHECK, SP>), dim3(n), dim3(64), 0, st, A); break;
    }
}

hipError_t launch_fs_chain(const FsArgs &A, int mode, int n, hipStream_t st)
{
    if (n <= 0) return hipSuccess;
    if (mode < FS_DIGEST || mode > FS_CHECK) return hipErrorInvalidValue;
    // variant B of kosk_fs_dev.hpp (DPP column sums + ds_bpermute exchanges).  Variant A (exchanges through LDS memory) measured 4 % slower
    // (2.26-2.35 against 2.14-2.28 us per permutation, profiles/r06_fs_chain.txt) and is compiled only for tools/fs_chain_time.py's A/B
    // (KOSK_FS_SPONGE=lds, a debug knob)
    static const bool lds = getenv("KOSK_FS_SPONGE") && !strcmp(getenv("KOSK_FS_SPONGE"), "lds");
    if (lds) fs_launch<FsSpongeLds>(A, mode, n, st);
    else fs_launch<FsSpongeBperm>(A, mode, n, st);
    return hipGetLastError();
}

} // namespace kosk
