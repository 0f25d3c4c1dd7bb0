// Wave-cooperative Keccak sponge: ONE Keccak-f[1600] state per 64-lane wave, one 32-bit word per lane.
//
// Why: the Fiat-Shamir aggregation hashes sha3_256(Tcomm[0..1454)) and sha3_256(ch_seeds) (mlwe_prover.cpp:130-135, :445-449;
// mlwe_verifier.cpp:40-44, :648-652) are ONE sequential chain of 343 permutations per proof and round (sha3_256_long of SURVEY.md
// 2.1 K4b).  The pipeline's one-state-per-lane sponge needs ~9 us per permutation when a wave runs alone (180 instructions per
// round at one issue slot per ~5 cycles): 3 ms per chain.  Here every vector instruction acts on the WHOLE state, so a
// round is 8 vector instructions plus two exchanges through LDS:
//
//   lane l = 6 x + y + 32 h  holds half h (0: even bits, 1: odd bits of the bit-interleaved form) of the 64-bit word (x, y);
//   the lanes y = 5 and x = 5 hold zeros and stay zero (their exchange addresses point at a zero pad).
//   theta   p = a ^ a[lane ^ 1]                        pair sums (y0,y1), (y2,y3); (y4, zero lane) = a4: one DPP-fused xor
//           T[slot] = p;  Cm = xor of the 3 words of column x-1 (this half), Cp = the same of column x+1 (OTHER half): one
//           16-byte LDS read each;  a ^= Cm ^ rotl32(Cp, h == 0)    (rotl64 by 1, interleaved: E' = rotl32(O, 1), O' = E)
//   rho     a = rotl32(a, k)                            64-bit offset r: k = r >> 1 (+1 on the odd half when r is odd; the halves
//                                                       change places when r is odd, which the pi write address absorbs)
//   pi/chi  B[pi(x, y), h'] = a (and a ghost copy 5 words further: rows x' = 0..4, 0, 1);  (b0, b1, b2) = B[x .. x+2];
//           a = b0 ^ (~b1 & b2) ^ rc                    chi is one v_bitop3, iota one xor with a per-lane register
//
// tools/fs_chain_model.py is the lane-level model of exactly these tables and exchanges (checked against hashlib in the CPU suite).
// Semantics: kyber/fips202.c:82-344 (KeccakF1600_StatePermute), :461-485 (absorb / squeeze), :745-754 (sha3_256).
#pragma once
#include <hip/hip_runtime.h>

#include "kosk_keccak_dev.hpp"

namespace kosk {

// LDS map of one wave's exchange area, in 32-bit words
constexpr int FSW_T = 0;      // theta: [half][column x][4]: (y0^y1, y2^y3, y4, scratch)
constexpr int FSW_B = 40;     // pi/chi: [half][row y][10]: x' = 0..4, ghosts of x' = 0, 1, three scratch words
constexpr int FSW_ZERO = 140; // four zero words (what the idle lanes read)
constexpr int FSW_JUNK = 144; // 16 words the idle lanes write to
constexpr int FSW_WORDS = 160;

struct FsLane {
    uint32_t wT, rTm, rTp, wB, rB; // word indices into the wave's exchange area
    uint32_t sh_theta, sh_rho;     // v_alignbit shift amounts (rotate right by sh = rotate left by 32 - sh)
    uint32_t half;                 // 0 even bits, 1 odd bits
    int word;                      // x + 5 y of an active lane, 63 for an idle one
};

__device__ __forceinline__ FsLane fs_lane_setup(int lane)
{
    FsLane L;
    const int h = lane >> 5, r = lane & 31, x = r / 6, y = r % 6;
    const bool act = x < 5 && y < 5;
    L.half = (uint32_t)h;
    if (!act) {
        L.word = 63;
        L.wT = FSW_JUNK + (lane & 7);
        L.rTm = L.rTp = FSW_ZERO;
        L.wB = FSW_JUNK + 8 + (lane & 1);
        L.rB = FSW_ZERO;
        L.sh_theta = L.sh_rho = 0;
        return L;
    }
    L.word = x + 5 * y;
    const int slot = y == 1 ? 0 : y == 3 ? 1 : y == 4 ? 2 : 3;
    L.wT = FSW_T + (h * 5 + x) * 4 + slot;
    L.rTm = FSW_T + (h * 5 + (x + 4) % 5) * 4;
    L.rTp = FSW_T + ((1 - h) * 5 + (x + 1) % 5) * 4;
    L.sh_theta = h == 0 ? 31u : 0u;
    // the rho offset of word (x, y) by the walk of FIPS 202 3.2.2 (once per kernel; equals kRho[x + 5 y] of kosk_math.hpp)
    int rot = 0;
    for (int t = 0, wx = 1, wy = 0; t < 24; t++) {
        if (wx == x && wy == y) rot = ((t + 1) * (t + 2) / 2) & 63;
        const int nx = wy, ny = (2 * wx + 3 * wy) % 5;
        wx = nx; wy = ny;
    }
    const int k = (rot >> 1) + (((rot & 1) && h == 1) ? 1 : 0);
    L.sh_rho = (uint32_t)((32 - k) & 31);
    const int h2 = h ^ (rot & 1), x2 = y, y2 = (2 * x + 3 * y) % 5;
    L.wB = FSW_B + (h2 * 5 + y2) * 10 + x2;
    L.rB = FSW_B + (h * 5 + y) * 10 + x;
    return L;
}

// even bits (sel 0) or odd bits (sel 1) of the 64-bit value hi:lo
__device__ __forceinline__ uint32_t fs_deinterleave_half(uint32_t lo, uint32_t hi, uint32_t sel)
{
    uint32_t a = (lo >> sel) & 0x55555555u, b = (hi >> sel) & 0x55555555u;
    a = (a | (a >> 1)) & 0x33333333u; b = (b | (b >> 1)) & 0x33333333u;
    a = (a | (a >> 2)) & 0x0F0F0F0Fu; b = (b | (b >> 2)) & 0x0F0F0F0Fu;
    a = (a | (a >> 4)) & 0x00FF00FFu; b = (b | (b >> 4)) & 0x00FF00FFu;
    a = (a | (a >> 8)) & 0x0000FFFFu; b = (b | (b >> 8)) & 0x0000FFFFu;
    return a | (b << 16);
}
__device__ __forceinline__ uint32_t fs_spread16(uint32_t v) // bit i of the low 16 bits -> bit 2 i
{
    v &= 0xFFFFu;
    v = (v | (v << 8)) & 0x00FF00FFu;
    v = (v | (v << 4)) & 0x0F0F0F0Fu;
    v = (v | (v << 2)) & 0x33333333u;
    v = (v | (v << 1)) & 0x55555555u;
    return v;
}
// the 64-bit word whose even bits are e and whose odd bits are o
__device__ __forceinline__ void fs_interleave(uint32_t e, uint32_t o, uint32_t &lo, uint32_t &hi)
{
    lo = fs_spread16(e) | (fs_spread16(o) << 1);
    hi = fs_spread16(e >> 16) | (fs_spread16(o >> 16) << 1);
}

// this lane's half of the 24 round constants (zero everywhere but on the two lanes of word (0, 0))
struct FsRc {
    uint32_t v[24];
};
__device__ __forceinline__ FsRc fs_rc_setup(const FsLane &L)
{
    FsRc rc;
#pragma unroll
    for (int r = 0; r < 24; r++) {
        const uint64_t c = kKeccakRcDev[r];
        const uint32_t mine = fs_deinterleave_half((uint32_t)c, (uint32_t)(c >> 32), L.half);
        rc.v[r] = L.word == 0 ? mine : 0u;
    }
    return rc;
}

// Keccak-f[1600] on the wave's state.  `x` = the wave's exchange area (FSW_WORDS words of LDS, the four words at FSW_ZERO zeroed once).
// All 64 lanes must be active.  LDS operations of one wave execute in issue order, so the stores of an exchange are visible to the
// loads behind them without a barrier; the wave barriers only keep the compiler from moving an access across an exchange.
__device__ __forceinline__ void fs_permute(uint32_t &a, uint32_t *x, const FsLane &L, const FsRc &rc)
{
#pragma unroll
    for (int r = 0; r < 24; r++) {
        const uint32_t p = a ^ (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a, 0xB1 /* quad_perm:[1,0,3,2] */, 0xF, 0xF, true);
        x[L.wT] = p;
        __builtin_amdgcn_wave_barrier();
        const uint4 m = *reinterpret_cast<const uint4 *>(x + L.rTm);
        const uint4 q = *reinterpret_cast<const uint4 *>(x + L.rTp);
        __builtin_amdgcn_wave_barrier();
        const uint32_t cm = kx3(m.x, m.y, m.z), cp = kx3(q.x, q.y, q.z);
        a = kx3(a, cm, __builtin_amdgcn_alignbit(cp, cp, L.sh_theta));
        a = __builtin_amdgcn_alignbit(a, a, L.sh_rho);
        x[L.wB] = a;
        x[L.wB + 5] = a;
        __builtin_amdgcn_wave_barrier();
        const uint32_t b0 = x[L.rB], b1 = x[L.rB + 1], b2 = x[L.rB + 2];
        __builtin_amdgcn_wave_barrier();
        a = kchi(b0, b1, b2) ^ rc.v[r];
    }
}

// ---- Variant B: no LDS memory at all.  Columns of FIVE lanes that never straddle a 16-lane row,
//   lane(x, y, h) = 32 h + (x < 3 ? 5 x : 16 + 5 (x - 3)) + y,
// so a column's sum is three DPP-fused xors (row_shr:1, :2, :1; it lands on the lane y = 4), and both exchanges are ds_bpermute_b32
// gathers: theta two (the sums of columns x - 1 and x + 1), pi / chi three (the pre-images of words x, x + 1, x + 2 of the lane's row,
// rotated by their owners before they are sent).  The idle lanes (15, 26..31 of each half) hold junk no active lane ever reads.
// tools/fs_chain_model.py: WaveB.
__device__ __forceinline__ int fs_lane_b(int x, int y, int h) { return 32 * h + (x < 3 ? 5 * x : 16 + 5 * (x - 3)) + y; }

struct FsLaneB {
    uint32_t sCm, sCp, s0, s1, s2; // byte addresses (lane * 4) for ds_bpermute_b32
    uint32_t sh_theta, sh_rho;
    uint32_t half;
    int word;
};

__device__ __forceinline__ int fs_rho_of(int x, int y)
{
    int rot = 0;
    for (int t = 0, wx = 1, wy = 0; t < 24; t++) {
        if (wx == x && wy == y) rot = ((t + 1) * (t + 2) / 2) & 63;
        const int nx = wy, ny = (2 * wx + 3 * wy) % 5;
        wx = nx; wy = ny;
    }
    return rot;
}

__device__ __forceinline__ FsLaneB fs_lane_setup_b(int lane)
{
    FsLaneB L;
    const int h = lane >> 5, r = lane & 31;
    const int x = r < 15 ? r / 5 : r >= 16 && r < 26 ? 3 + (r - 16) / 5 : 5;
    const int y = r < 15 ? r % 5 : r >= 16 && r < 26 ? (r - 16) % 5 : 0;
    L.half = (uint32_t)h;
    if (x >= 5) {
        L.word = 63;
        L.sCm = L.sCp = L.s0 = L.s1 = L.s2 = (uint32_t)lane * 4;
        L.sh_theta = L.sh_rho = 0;
        return L;
    }
    L.word = x + 5 * y;
    L.sCm = 4u * (uint32_t)fs_lane_b((x + 4) % 5, 4, h);
    L.sCp = 4u * (uint32_t)fs_lane_b((x + 1) % 5, 4, 1 - h);
    L.sh_theta = h == 0 ? 31u : 0u;
    const int rot = fs_rho_of(x, y);
    const int k = (rot >> 1) + (((rot & 1) && h == 1) ? 1 : 0);
    L.sh_rho = (uint32_t)((32 - k) & 31);
    uint32_t src[3];
    for (int j = 0; j < 3; j++) { // word (x + j, y) of the permuted state sits, rotated, on the lane of its pre-image under pi
        const int X = (x + j) % 5, Y = y, ys = X, xs = (3 * (Y - 3 * X + 15)) % 5, hs = h ^ (fs_rho_of(xs, ys) & 1);
        src[j] = 4u * (uint32_t)fs_lane_b(xs, ys, hs);
    }
    L.s0 = src[0]; L.s1 = src[1]; L.s2 = src[2];
    return L;
}

template <int CTRL>
__device__ __forceinline__ uint32_t fs_dpp_xor(uint32_t moved, uint32_t other) // dpp(moved) ^ other, one v_xor_b32_dpp
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)moved, CTRL, 0xF, 0xF, true) ^ other;
}

__device__ __forceinline__ void fs_permute_b(uint32_t &a, const FsLaneB &L, const FsRc &rc)
{
#pragma unroll
    for (int r = 0; r < 24; r++) {
        const uint32_t t1 = fs_dpp_xor<0x111>(a, a);   // row_shr:1
        const uint32_t t2 = fs_dpp_xor<0x112>(t1, t1); // row_shr:2
        const uint32_t c = fs_dpp_xor<0x111>(t2, a);   // the column's sum on its lane y = 4
        const uint32_t cm = (uint32_t)__builtin_amdgcn_ds_bpermute((int)L.sCm, (int)c);
        const uint32_t cp = (uint32_t)__builtin_amdgcn_ds_bpermute((int)L.sCp, (int)c);
        a = kx3(a, cm, __builtin_amdgcn_alignbit(cp, cp, L.sh_theta));
        a = __builtin_amdgcn_alignbit(a, a, L.sh_rho);
        const uint32_t b0 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)L.s0, (int)a);
        const uint32_t b1 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)L.s1, (int)a);
        const uint32_t b2 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)L.s2, (int)a);
        a = kchi(b0, b1, b2) ^ rc.v[r];
    }
}

// The two variants behind one face (k_fs_chain is a template over it)
struct FsSpongeLds {
    FsLane L;
    FsRc rc;
    uint32_t *x;
    __device__ __forceinline__ void setup(int lane, uint32_t *xw)
    {
        L = fs_lane_setup(lane);
        rc = fs_rc_setup(L);
        x = xw;
        if (lane < 4) xw[FSW_ZERO + lane] = 0;
        __builtin_amdgcn_wave_barrier();
    }
    __device__ __forceinline__ void permute(uint32_t &a) const { fs_permute(a, x, L, rc); }
    __device__ __forceinline__ uint32_t half() const { return L.half; }
    __device__ __forceinline__ int word() const { return L.word; }
    static __device__ __forceinline__ int lane_of(int x_, int y_, int h_) { return 6 * x_ + y_ + 32 * h_; }
};
struct FsSpongeBperm {
    FsLaneB L;
    FsRc rc;
    __device__ __forceinline__ void setup(int lane, uint32_t *)
    {
        L = fs_lane_setup_b(lane);
        FsLane t;
        t.half = L.half; t.word = L.word;
        rc = fs_rc_setup(t);
    }
    __device__ __forceinline__ void permute(uint32_t &a) const { fs_permute_b(a, L, rc); }
    __device__ __forceinline__ uint32_t half() const { return L.half; }
    __device__ __forceinline__ int word() const { return L.word; }
    static __device__ __forceinline__ int lane_of(int x_, int y_, int h_) { return fs_lane_b(x_, y_, h_); }
};

} // namespace kosk
