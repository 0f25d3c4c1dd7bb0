// gfx950 kernels that exist only on the verifier side (mlwe_verifier.cpp:4-686).
// The heavy steps reuse kosk_kernels.hip: k_commit_hash (opened lanes through a
// lane map), k_lincomb (opened columns), k_ntt256, k_matvec_ntt and k_gemm_modq
// (recon_secrets_*, recompute_share_secrets_ddeg, and -- with a per-proof
// operand -- the application of the interpolation operator that replaces NTL
// interpolate()/eval(), mlwe_verifier.cpp:201-219 etc.).
#include <hip/hip_runtime.h>

#include <utility>

#include "kosk_device.hpp"
#include "kosk_keccak_dev.hpp"
#include "kosk_keccak_split_dev.hpp"
#include "kosk_math.hpp"

#include <cstdlib>
#include "kosk_limb_dev.hpp"

namespace kosk {

constexpr int DIS_TILE = 64 * 80;

// inverse of k_assemble_fields: proof image -> rows at the listed party columns.
// u16 elements >= q never come out of an honest prover; what happens to one follows the reference record by record
// (INTEGRATION.md 6 has the table):
//  * records of UNOPENED parties go to the row matrix folded mod q -- the reference only ever multiplies them (gf3329_mul's
//    `% 3329`, ss.cpp:47, :66) or converts them to ZZ_p (mlwe_verifier.cpp:193-198, :329-332, :402-408, :515-521), both of which
//    reduce -- except the s + r / e + r shares, which it also compares RAW with the recomputed canonical shares (:232-246): an
//    element >= q there sets that check's fail bit (FieldDesc::noncanon_bit);
//  * records of OPENED parties go to the opened matrix RAW: the reference hashes them raw and compares / adds / subtracts them
//    with its non-reducing gf3329_add / gf3329_sub (k_lincomb, k_check_opened and the gate block below reproduce that).
// VerifyArgs::strict (KOSK_STRICT_ENCODING=1) is the behaviour of rounds 1-4: any element >= q in a record the reference reads
// marks the proof malformed (fail bit 0).
__global__ __launch_bounds__(64) void k_disassemble_fields(VerifyArgs v, const FieldDesc *__restrict__ fields, FieldPlan plan,
                                                          const int16_t *__restrict__ rowtab,
                                                          const uint8_t *__restrict__ proof, size_t image_stride, uint32_t off_tcomm,
                                                          uint32_t off_comm, uint8_t *__restrict__ dig1, uint8_t *__restrict__ dig2,
                                                          GateOffsets go)
{
    { // last blocks: multiplication-gate outputs u = z_2d - z_d of the OPENED parties (mlwe_verifier.cpp:468-502), read from
      // the party-major image records (30 + 24 contiguous bytes per party) instead of 54 scattered matrix columns
        const int nfb = plan.nrest * NWIN + plan.nopen * ((NOPEN + 63) / 64) + (NREST * 4 + 63) / 64;
        if ((int)blockIdx.x >= nfb) {
            const int t = ((int)blockIdx.x - nfb) * 64 + threadIdx.x, b = blockIdx.y;
            if (t >= NOPEN) return;
            const RowMap &rm = v.rm;
            const uint8_t *img = proof + (size_t)b * image_stride;
            uint16_t *Pc = v.P + (size_t)b * v.proof_stride + NSEC + v.opened[(size_t)b * v.sel_stride + t];
            uint16_t *Oc = v.O + (size_t)b * v.o_stride + t; // the same values for the view hash (coalesced over t)
            // raw operands: gf3329_mul reduces the product of any two u16 (:475-483), gf3329_sub does not (:487-492) -- the raw
            // difference enters the view hash (:624-629) and, through gf3329_mul again (folded), recon_secrets_2ddeg
            auto rd = [&](uint32_t off, int idx) { return (uint32_t)reinterpret_cast<const uint16_t *>(img + off)[idx]; };
            for (int who = 0; who < 2; who++)
                for (int i = 0; i < rm.K; i++) {
                    const uint32_t osub = who ? go.esub : go.ssub, oz = who ? go.ze : go.zs;
                    uint32_t prev = rd(osub, (t * rm.K + i) * rm.E);
                    for (int j = 0; j < rm.Z; j++) {
                        const uint32_t z2 = gf_mul(prev, rd(osub, (t * rm.K + i) * rm.E + j + 1)); // < 2^32 for any two u16
                        const uint32_t zd = rd(oz, (t * rm.K + i) * rm.Z + j);
                        const uint16_t uv = (uint16_t)ref_sub_u16(z2, zd);
                        Pc[(size_t)(who ? rm.ue(i, j) : rm.us(i, j)) * RS] = (uint16_t)gf_fold(uv); // recon_secrets_2ddeg reads the merged row, through gf3329_mul: folded
                        Oc[(size_t)(who ? rm.ue(i, j) : rm.us(i, j)) * OS] = uv;
                        prev = zd;
                    }
                }
            return;
        }
    }
    { // blocks past the field tiles: Tcomm / comm of the unopened parties into the two digest tables  mlwe_verifier.cpp:36-38, :645-647
        const int nfield_blocks = plan.nrest * NWIN + plan.nopen * ((NOPEN + 63) / 64);
        if ((int)blockIdx.x >= nfield_blocks) {
            // one thread per (unopened party, table, 16-byte half of its digest); the image side is read with 4-byte loads (its digest
            // fields are 4-byte aligned for every Kyber parameter set; 2-byte loads otherwise), the table side written with one
            // 16-byte store.  Rounds 1-6a: two bytes per thread and table, 326 workgroups per proof (see k_assemble_groups).
            const int q = ((int)blockIdx.x - nfield_blocks) * 64 + threadIdx.x, b = blockIdx.y;
            if (q >= NREST * 4) return;
            const uint8_t *img = proof + (size_t)b * image_stride;
            const int i = q >> 2, tab = (q >> 1) & 1, half = q & 1;
            const uint8_t *src = img + (tab ? off_comm : off_tcomm) + (size_t)i * 32 + 16 * half;
            uint32_t w[4];
            if ((reinterpret_cast<uintptr_t>(src) & 3) == 0) {
#pragma unroll
                for (int k = 0; k < 4; k++) w[k] = reinterpret_cast<const uint32_t *>(src)[k];
            } else {
#pragma unroll
                for (int k = 0; k < 4; k++) w[k] = (uint32_t)reinterpret_cast<const uint16_t *>(src)[2 * k] | ((uint32_t)reinterpret_cast<const uint16_t *>(src)[2 * k + 1] << 16);
            }
            const size_t dst = ((size_t)b * NPARTY + v.rest[(size_t)b * v.sel_stride + i]) * 32 + 16 * half;
            *reinterpret_cast<uint4 *>((tab ? dig2 : dig1) + dst) = make_uint4(w[0], w[1], w[2], w[3]);
            return;
        }
    }
    __shared__ __attribute__((aligned(16))) uint16_t tile[DIS_TILE]; // image order
    const int b = blockIdx.y, x = blockIdx.x, lane = threadIdx.x;
    const bool kind = x < plan.nrest * NWIN; // unopened parties of one aligned window (see k_assemble_fields)
    const uint16_t *orow = v.opened + (size_t)b * v.sel_stride;
    int i0, cnt, f;
    if (kind) {
        const int win = x % NWIN;
        f = plan.rest_ids[x / NWIN];
        i0 = orow[SEL_WIN + win];
        cnt = (int)orow[SEL_WIN + win + 1] - i0;
    } else {
        const int y = x - plan.nrest * NWIN, nch = (NOPEN + 63) / 64;
        f = plan.open_ids[y / nch];
        i0 = (y % nch) * 64;
        cnt = min(64, NOPEN - i0);
    }
    if (cnt <= 0) return;
    const FieldDesc fd = fields[f];
    if (kind && fd.limit) {
        // Records the reference's verify() never reads (FieldDesc::limit: beta / gamma shares of parties >= 407, t / eta shares behind the
        // 407th and u shares behind the 813th unopened party) are not range-checked and no kernel of the verifier reads their place in the
        // row matrix (recon_secrets_*: party columns 0 .. 406 / 812; the interpolations: the first 407 / 813 entries of `rest`): a window
        // that holds only such records has nothing to do -- 43 % of the image's bytes and 44 % of the launch's stores (round 6; rounds
        // 2-5 scattered them all).  Every party of window w is >= 64 w, every record index >= i0.
        if ((fd.limit_by_party ? 64 * (x % NWIN) : i0) >= fd.limit) return;
        if (!fd.limit_by_party) cnt = min(cnt, fd.limit - i0); // the window's tail behind the last record that is read
    }
    const uint16_t *in = reinterpret_cast<const uint16_t *>(proof + (size_t)b * image_stride + fd.off) + (size_t)i0 * fd.width;
    const int n16 = cnt * fd.width;
    const int head = (int)((reinterpret_cast<uintptr_t>(in) >> 1) & 1);
    const int body = (n16 - head) >> 1;
    bool bad = false;
    // the tile holds the image's u16 as they are; what an element >= q means is decided per record below
    auto canon = [&](uint32_t x_) { return (uint16_t)x_; };
    if (lane == 0 && head) tile[0] = canon(in[0]);
    const uint32_t *in32 = reinterpret_cast<const uint32_t *>(in + head);
    // every load of the block in flight before the first LDS write (a 64-party block of the widest field is 40 dwords per lane;
    // rounds 2-4 went eight at a time: five dependent round trips per block)
    auto fetch_all = [&]<int NX>(std::integral_constant<int, NX>) {
        uint32_t xv[NX];
#pragma unroll
        for (int u = 0; u < NX; u++) {
            const int q = u * 64 + lane;
            xv[u] = q < body ? in32[q] : 0;
        }
#pragma unroll
        for (int u = 0; u < NX; u++) {
            const int q = u * 64 + lane;
            if (q < body) {
                tile[head + 2 * q] = canon(xv[u] & 0xFFFFu);
                tile[head + 2 * q + 1] = canon(xv[u] >> 16);
            }
        }
    };
    if (body <= 64 * 8) fetch_all(std::integral_constant<int, 8>{});
    else if (body <= 64 * 24) fetch_all(std::integral_constant<int, 24>{});
    else fetch_all(std::integral_constant<int, 40>{});
    if (lane == 1 && head + 2 * body < n16) tile[n16 - 1] = canon(in[n16 - 1]);
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    const uint16_t *sel = kind ? v.rest + (size_t)b * v.sel_stride : orow;
    if (lane < cnt) {
        const int16_t *rt = rowtab + fd.rowtab_off;
        const uint16_t *t = tile + lane * fd.width;
        bool big = false; // an element >= q in this record
        if (kind) {
            const int party = sel[i0 + lane];
            uint16_t *dst = v.P + (size_t)b * v.proof_stride + NSEC + party;
            if (!(fd.limit && fd.limit_by_party && party >= fd.limit)) { // (a record behind a limit by index is not in cnt any more)
#pragma unroll 8
                for (int e = 0; e < fd.width; e++) { const uint32_t x_ = t[e]; big |= x_ >= (uint32_t)Q; dst[(size_t)rt[e] * RS] = (uint16_t)gf_fold(x_); }
            }
        } else {
            // records of the opened parties: into the opened matrix, consecutive lanes = consecutive entries of a row
            uint16_t *dst = v.O + (size_t)b * v.o_stride + i0 + lane;
#pragma unroll 8
            for (int e = 0; e < fd.width; e++) { const uint32_t x_ = t[e]; big |= x_ >= (uint32_t)Q; dst[(size_t)rt[e] * OS] = (uint16_t)x_; }
        }
        bad = big;
    }
    if (bad) {
        if (v.strict) atomicOr(&v.fail[b], 1u << FB_MALFORMED);
        else if (fd.noncanon_bit >= 0) atomicOr(&v.fail[b], 1u << fd.noncanon_bit);
    }
}


// ---- SHA3-256 of the opened parties' Tcomm / view messages, one opened party per thread ---------------
// word w (u16) of opened party i's message, K / VIEW compile-time so that every source is resolved statically
template <int K, bool VIEW>
struct OpenedMsg {
    static constexpr int M = NCHK + 2 * K + 1, Z = 2 * (K == 2 ? 3 : 2);
    static constexpr int BASE = 2 * K + 2 * M;                      // s, e, f, NTT f
    static constexpr int WORDS = VIEW ? BASE + 4 * K + 4 * Z * K : BASE; // + beta, gamma, s+r, e+r, gates
    template <int W>
    static __device__ __forceinline__ uint32_t word(const OpenedHashArgs &a, const uint8_t *img, const uint16_t *col, const uint16_t *ocol, int i)
    {
        auto im = [&](uint32_t off, int idx) { return (uint32_t)reinterpret_cast<const uint16_t *>(img + off)[idx]; };
        if constexpr (W < K) return im(a.off_s, i * K + W);
        else if constexpr (W < 2 * K) return im(a.off_e, i * K + (W - K));
        else if constexpr (W < 2 * K + M) return im(a.off_f, i * M + (W - 2 * K));
        else if constexpr (W < BASE) return im(a.off_nttf, i * M + (W - 2 * K - M));
        else if constexpr (W < BASE + K) return ocol[(size_t)(a.rm.beta0 + (W - BASE)) * OS];
        else if constexpr (W < BASE + 2 * K) return ocol[(size_t)(a.rm.gamma0 + (W - BASE - K)) * OS];
        else if constexpr (W < BASE + 3 * K) return col[(size_t)(a.rm.sr + (W - BASE - 2 * K)) * RS];
        else if constexpr (W < BASE + 4 * K) return col[(size_t)(a.rm.er + (W - BASE - 3 * K)) * RS];
        else {
            constexpr int g = W - (BASE + 4 * K), j = g / (4 * Z), q = g % (4 * Z), part = q / Z, z = q % Z;
            if constexpr (part == 0) return im(a.off_zs, (i * K + j) * Z + z);
            else if constexpr (part == 1) return im(a.off_ze, (i * K + j) * Z + z);
            else if constexpr (part == 2) return ocol[(size_t)(a.rm.gate + j * 4 * Z + 2 * Z + z) * OS];
            else return ocol[(size_t)(a.rm.gate + j * 4 * Z + 3 * Z + z) * OS];
        }
    }
};

// The same hashes on the LANE-PAIR sponge (kosk_keccak_split_dev.hpp; round 5, default): one state on two adjacent lanes -- the even
// lane holds the low halves of the 25 words, the odd lane the high halves -- 120 instead of 180 vector instructions per lane and
// round.  These launches are 3 (5 with pairs) waves per proof, every wave alone on its SIMD: their time is the dependent chain of
// three / four permutations at one wave's issue rate (~9 us each), so two thirds of the instructions are two thirds of the time.
// A lane absorbs only its half of every 64-bit word: the even lane message words 4 L, 4 L + 1, the odd lane 4 L + 2, 4 L + 3.
template <int K, bool VIEW, int W0, bool HI>
__device__ __forceinline__ void opened_absorb_half(KHalf &s, const OpenedHashArgs &a, const uint8_t *img, const uint16_t *col,
                                                   const uint16_t *ocol, int i, const uint8_t *prefix)
{
    using Msg = OpenedMsg<K, VIEW>;
    constexpr int PW = VIEW ? 16 : 0, TOTAL = PW + Msg::WORDS;
    auto put = [&](auto lc) {
        constexpr int L = decltype(lc)::value;       // 64-bit word of the rate block
        constexpr int gw = W0 + 4 * L + (HI ? 2 : 0); // first of this lane's two message words
        if constexpr (gw < TOTAL) {
            uint32_t v;
            if constexpr (gw < PW) v = *reinterpret_cast<const uint32_t *>(prefix + 2 * gw);
            else {
                v = Msg::template word<gw - PW>(a, img, col, ocol, i);
                if constexpr (gw + 1 < TOTAL) v |= Msg::template word<gw + 1 - PW>(a, img, col, ocol, i) << 16;
            }
            s.w[L] ^= v;
        }
    };
    [&]<int... Ls>(std::integer_sequence<int, Ls...>) { (put(std::integral_constant<int, Ls>{}), ...); }(std::make_integer_sequence<int, 17>{});
}

template <int K, bool VIEW>
__global__ __launch_bounds__(64) void k_opened_hash_pair(OpenedHashArgs a)
{
    const int pr = (blockIdx.x * 64 + threadIdx.x) >> 1, b = blockIdx.y;
    const bool hi = threadIdx.x & 1;
    const bool live = pr < NOPEN;
    const int i = live ? pr : NOPEN - 1; // idle pairs hash the last opened party again and store nothing (all 64 lanes stay active: DPP)
    using Msg = OpenedMsg<K, VIEW>;
    constexpr int PW = VIEW ? 16 : 0, TOTAL = PW + Msg::WORDS, NBLK = TOTAL / 68 + 1;
    const int party = a.opened[(size_t)b * a.sel_stride + i];
    const uint8_t *img = a.proof + (size_t)b * a.image_stride;
    const uint16_t *col = a.P + (size_t)b * a.proof_stride + NSEC + party;
    const uint16_t *ocol = a.O + (size_t)b * a.o_stride + i;
    const size_t dig = ((size_t)b * NPARTY + party) * 32;
    const uint8_t *prefix = VIEW ? a.prefix + dig : nullptr;
    KHalf s;
#pragma unroll
    for (int k = 0; k < 25; k++) s.w[k] = 0;
    [&]<int... Bs>(std::integer_sequence<int, Bs...>) {
        (([&] {
             if (hi) opened_absorb_half<K, VIEW, Bs * 68, true>(s, a, img, col, ocol, i, prefix);
             else opened_absorb_half<K, VIEW, Bs * 68, false>(s, a, img, col, ocol, i, prefix);
             if constexpr (Bs == NBLK - 1) {
                 constexpr int padbyte = 2 * TOTAL - (NBLK - 1) * 136;
                 constexpr uint32_t padv = 0x06u << (8 * (padbyte % 4));
                 if (hi == ((padbyte % 8) >= 4)) s.w[padbyte / 8] ^= padv;
                 if (hi) s.w[16] ^= 0x80000000u;
             }
             keccak_f1600_split(s, hi);
         }()),
         ...);
    }(std::make_integer_sequence<int, NBLK>{});
    if (!live) return;
    uint32_t *o = reinterpret_cast<uint32_t *>(a.out + dig) + (hi ? 1 : 0);
#pragma unroll
    for (int L = 0; L < 4; L++) o[2 * L] = s.w[L];
    if (a.out_compact) {
        uint32_t *oc = reinterpret_cast<uint32_t *>(a.out_compact + ((size_t)b * NOPEN + i) * 32) + (hi ? 1 : 0);
#pragma unroll
        for (int L = 0; L < 4; L++) oc[2 * L] = s.w[L];
    }
}

template <int K>
static void launch_opened_hash_k(const OpenedHashArgs &a, bool view, int nproofs, hipStream_t st)
{
    dim3 grid((2 * NOPEN + 63) / 64, nproofs);
    if (view) hipLaunchKernelGGL((k_opened_hash_pair<K, true>), grid, dim3(64), 0, st, a);
    else hipLaunchKernelGGL((k_opened_hash_pair<K, false>), grid, dim3(64), 0, st, a);
}
hipError_t launch_opened_hash(const OpenedHashArgs &a, int K, bool view, int nproofs, hipStream_t st)
{
    if (K == 2) launch_opened_hash_k<2>(a, view, nproofs, st);
    else if (K == 3) launch_opened_hash_k<3>(a, view, nproofs, st);
    else launch_opened_hash_k<4>(a, view, nproofs, st);
    return hipGetLastError();
}


// ---- interpolation operators over the nodes x_j = 256 + rest[j] (see InterpArgs) ---------------
__device__ __forceinline__ uint32_t gf_neg_if(uint32_t v, int odd) { return (odd & 1) && v ? (uint32_t)Q - v : v; }
// a b mod q for a, b < 2^16 with a b < 2^32 in seven full-rate instructions (24-bit multiply + gf_reduce_u32) where `a * b % Q` costs three
// quarter-rate 32-bit multiplies; the canonical representative of a difference |d| < q
__device__ __forceinline__ uint32_t gf_mul_fast(uint32_t a, uint32_t b) { return gf_reduce_u32(__umul24(a, b)); }
__device__ __forceinline__ uint32_t gf_diff(int d) { return (uint32_t)(d < 0 ? d + Q : d); }

// weights of both sets, l(k) and the node map of set 0
__device__ __forceinline__ void interp_setup_block(const InterpArgs &a, const int bx, const int b, const int set, uint16_t *is)
{
    const int t = bx * 256 + threadIdx.x;
    const int n = set ? DEG2 + 1 : DEG + 1;
    const uint16_t *rest = a.rest + (size_t)b * a.sel_stride;
    // the sorted opened list in LDS: both loops below walk it once per thread (from global memory every trip waited for
    // its own load: 40 us for the whole set-up)
    if (threadIdx.x < NOPEN) is[threadIdx.x] = a.isort[(size_t)b * a.sel_stride + threadIdx.x];
    __syncthreads();
    const int h0 = a.hrange[b * 4 + 2 * set], h1 = a.hrange[b * 4 + 2 * set + 1];
    const int lo = rest[0], hi = rest[n - 1]; // parties; the nodes are [lo, hi] minus the holes is[h0..h1)
    if (t < 832) { // w_j = prod_holes (x_j - h) / ((x_j - lo)! (hi - x_j)! (-1)^(hi - x_j))
        uint32_t w = 0;
        if (t < n) {
            const int xj = rest[t];
            uint32_t pr = 1;
            for (int h = h0; h < h1; h++) pr = gf_mul_fast(pr, gf_diff(xj - (int)is[h])); // |x_j - h| < 1 454 (the launch is bound by these products: round 6)
            w = gf_mul(pr, gf_mul(a.invfact[xj - lo], a.invfact[hi - xj]));
            w = gf_neg_if(w, hi - xj);
        }
        a.w[((size_t)b * 2 + set) * 832 + t] = (uint16_t)w;
    }
    if (set == 0 && t < 416) { // l(k) = prod_{p in [lo,hi]} (k - 256 - p) / prod_holes (k - 256 - h)
        uint32_t l = 0;
        int node = -1;
        if (t <= DEG) {
            const int kp = t - NSEC; // evaluation point as a "party" coordinate (negative for packed secrets)
            bool is_hole = false;
            uint32_t den = 1;
            int below = 0;
            for (int h = h0; h < h1; h++) {
                const int d = kp - (int)is[h];
                if (d == 0) is_hole = true;
                else den = gf_mul_fast(den, gf_diff(d)); // d in [-1 709, 150]
                below += (int)is[h] < kp;
            }
            if (kp < lo) { // (-1)^cnt (hi - kp)! / (lo - 1 - kp)!
                l = gf_mul(a.fact[hi - kp], a.invfact[lo - 1 - kp]);
                l = gf_neg_if(l, hi - lo + 1);
                l = gf_mul(l, a.inv[den]);
            } else if (is_hole) { // inside the span on an opened party: skip the zero factor on both sides
                l = gf_mul(a.fact[kp - lo], a.fact[hi - kp]);
                l = gf_neg_if(l, hi - kp);
                l = gf_mul(l, a.inv[den]);
            } else { // kp is a node: p(k) is the given share itself
                node = kp - lo - below;
            }
        }
        a.ell[(size_t)b * 416 + t] = (uint16_t)l;
        a.node_of[(size_t)b * 416 + t] = (int16_t)node;
    }
}

// weights of both node sets, l(k) and the node map of set 0: 8 blocks per proof
__global__ __launch_bounds__(256) void k_interp_setup(InterpArgs a, int nproofs)
{
    __shared__ uint16_t is_s[256];
    const int id = blockIdx.x;
    interp_setup_block(a, id & 3, (id >> 2) % nproofs, (id >> 2) / nproofs, is_s);
}

// The weighted shares y[r][j] = w[b][set][j] * P[b][rows[r]][256 + rest[b][j]] of both interpolations (degree-d rows, then
// the u rows of degree 2d), written as the MFMA operand of k_interp_apply: int8 limb tiles [k-step][column tile][limb] of
// 1 KiB in fragment order (lane 16 (j / 16 % 4) + r % 16 holds its 16 consecutive nodes at byte 16 lane), zero where
// j >= the node count or r >= the row count.  One thread per (column, 16 nodes).
constexpr int IA_NT1 = 4, IA_NT2 = 2, IA_KS1 = 7, IA_KS2 = 13; // 64 columns x 448 nodes, 32 columns x 832 nodes (K = 4: 52 and 32)
constexpr int IA_Y1_BYTES = IA_KS1 * IA_NT1 * 2048, IA_Y2_BYTES = IA_KS2 * IA_NT2 * 2048;

__global__ __launch_bounds__(256) void k_gather_frags(const uint16_t *__restrict__ P, size_t proof_stride,
                                                      const int16_t *__restrict__ rows1, int nrows1, uint8_t *__restrict__ out1,
                                                      const int16_t *__restrict__ rows2, int nrows2, uint8_t *__restrict__ out2,
                                                      const uint16_t *__restrict__ rest, int sel_stride, const uint16_t *__restrict__ w)
{
    constexpr int T1 = IA_NT1 * 16 * IA_KS1 * 4;
    int t = blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.y;
    const int set = t >= T1;
    if (set) t -= T1;
    const int KS = set ? IA_KS2 : IA_KS1, NT = set ? IA_NT2 : IA_NT1;
    if (t >= NT * 16 * KS * 4) return;
    // consecutive threads: consecutive 16-node chunks of ONE row (round 6).  A wave's sixteen gather instructions then walk two or three
    // rows' node columns side by side -- about 25 different 128-byte lines per instruction; with consecutive threads on consecutive rows
    // (rounds 3-5: the coalesced order of the two stores below) every lane of every gather had a line of its own, and the launch ran at
    // the texture path's line rate: 25 us per 276 proofs for 30 MB.
    const int kc16 = t % (KS * 4), r = t / (KS * 4);
    const int ncols = set ? DEG2 + 1 : DEG + 1, nrows = set ? nrows2 : nrows1;
    uint4 x0 = make_uint4(0, 0, 0, 0), x1 = x0;
    if (r < nrows && kc16 * 16 < ncols) {
        const uint16_t *src = P + (size_t)b * proof_stride + (size_t)(set ? rows2 : rows1)[r] * RS + NSEC;
        const uint16_t *wb = w + ((size_t)b * 2 + set) * 832 + kc16 * 16;   // zero behind the nodes (k_interp_setup)
        const uint16_t *rb = rest + (size_t)b * sel_stride + kc16 * 16;
        const uint4 w0 = *reinterpret_cast<const uint4 *>(wb), w1 = *reinterpret_cast<const uint4 *>(wb + 8);
        const uint4 r0 = *reinterpret_cast<const uint4 *>(rb), r1 = *reinterpret_cast<const uint4 *>(rb + 8);
        const uint32_t ww[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w}, rw[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
        uint32_t v[16];
#pragma unroll
        for (int q = 0; q < 16; q++) v[q] = src[(rw[q >> 1] >> (16 * (q & 1))) & 0xFFFFu];
        uint32_t o[8];
#pragma unroll
        for (int q = 0; q < 16; q += 2) {
            const uint32_t lo = gf_mul_fast(ww[q >> 1] & 0xFFFFu, v[q]), hi = gf_mul_fast(ww[q >> 1] >> 16, v[q + 1]); // w < q, v any u16
            o[q >> 1] = lo | (hi << 16);
        }
        x0 = make_uint4(o[0], o[1], o[2], o[3]);
        x1 = make_uint4(o[4], o[5], o[6], o[7]);
    }
    uint4 lo, hi;
    gm_split16(x0, x1, lo, hi);
    uint8_t *d = (set ? out2 + (size_t)b * IA_Y2_BYTES : out1 + (size_t)b * IA_Y1_BYTES) +
                 (size_t)(((kc16 >> 2) * NT + (r >> 4)) * 2) * 1024 + ((kc16 & 3) * 16 + (r & 15)) * 16;
    *reinterpret_cast<uint4 *>(d) = lo;
    *reinterpret_cast<uint4 *>(d + 1024) = hi;
}

// ---- interpolation of the unopened shares, applied without ever storing the operator -------------------------------
// NTL interpolate + eval of mlwe_verifier.cpp:201-219, :337-350, :410-440 (degree d: the values at points 0..406 of
// every interpolated sharing) and :523-543 (degree 2d: the Cauchy sums of the u shares at the 256 packed positions).
// In barycentric form the operator of one proof is the Cauchy matrix C[k][j] = 1/(k - x_j) over its nodes
// x_j = 256 + rest[j] (407 or 813 of them); it differs from proof to proof and multiplies only 34-52 (24-32) vectors, so
// materialising it (40 MB written and read back per 46 proofs, 17 M table look-ups behind per-element stores) cost more
// than the product.  Here every lane builds its own MFMA operand fragment -- row = evaluation point lane & 15, sixteen
// consecutive nodes lane >> 4 -- from the inverse table in LDS, limb-splits it in registers and multiplies right away;
// the weighted shares (k_gather_cols2) sit in LDS as the other operand.  A wave owns 16 evaluation points, a workgroup
// 64.  Epilogue: reduce mod q; degree d: times l(k), or the share itself where k is a node (what k_interp_fixup did),
// into the row matrix; degree 2d: the raw sums for k_check_batch.
// IA_OFF: the table of limb pairs is indexed by k - x_j + IA_OFF with k - 256 in [-256, 150] and x_j in [0, 1453]
constexpr int IA_OFF = NPARTY - 1 + NSEC, IA_TAB = IA_OFF + (DEG - NSEC) + 1; // 1709, 1860 entries (+ one of padding)

struct InterpApplyArgs {
    InterpArgs ia;
    uint16_t *P;
    size_t proof_stride;
    const int16_t *src_rows, *dst_rows; // degree d: shares read from / values written to these rows of the row matrix
    int n1;                             // sharings of degree d per proof (<= 64)
    const uint8_t *y1;                  // [proof][IA_Y1_BYTES] weighted shares as fragment tiles (k_gather_frags)
    int n2;                             // sharings of degree 2d per proof (<= 32)
    const uint8_t *y2;                  // [proof][IA_Y2_BYTES]
    uint16_t *out2;                     // [proof][n2][256]
};

template <int SET>
__device__ __forceinline__ void interp_apply_block(const InterpApplyArgs &g, const int b, const int mblk, uint16_t *tab_s, uint16_t *rest_s)
{
    constexpr int KS = SET ? IA_KS2 : IA_KS1, NT = SET ? IA_NT2 : IA_NT1;
    constexpr int NEVAL = SET ? NSEC : DEG + 1;
    const InterpArgs &a = g.ia;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int ncol = SET ? g.n2 : g.n1;
    { // limb-pair table of the inverses: all of a thread's loads in flight at once
        constexpr int ND = (IA_TAB + 2) / 2, TRIPS = (ND + 255) / 256;
        uint32_t tmp[TRIPS];
#pragma unroll
        for (int i = 0; i < TRIPS; i++) {
            const int idx = tid + i * 256;
            tmp[i] = reinterpret_cast<const uint32_t *>(a.invlimb)[idx < ND ? idx : ND - 1];
        }
#pragma unroll
        for (int i = 0; i < TRIPS; i++) {
            const int idx = tid + i * 256;
            if (idx < ND) reinterpret_cast<uint32_t *>(tab_s)[idx] = tmp[i];
        }
    }
    // nodes (sel_stride leaves readable entries behind them: further unopened parties, which meet zero shares)
    for (int j = tid; j < KS * 64; j += 256) rest_s[j] = a.rest[(size_t)b * a.sel_stride + j];
    __syncthreads();

    const int m0 = (mblk * 4 + wv) * 16;
    if (m0 >= NEVAL) return;
    const int kq = m0 + (lane & 15) - NSEC + IA_OFF; // this lane's evaluation point, offset for the table
    const uint8_t *yt = (SET ? g.y2 + (size_t)b * IA_Y2_BYTES : g.y1 + (size_t)b * IA_Y1_BYTES) + lane * 16;
    v4i s0[NT], s1[NT], s2[NT];
#pragma unroll
    for (int j = 0; j < NT; j++) { s0[j] = (v4i){0, 0, 0, 0}; s1[j] = s0[j]; s2[j] = s0[j]; }
    v4i fb[2][2 * NT]; // the other operand's fragments, one k-step ahead
    auto load_b = [&](int ks, v4i (&dst)[2 * NT]) {
#pragma unroll
        for (int q = 0; q < 2 * NT; q++) dst[q] = *reinterpret_cast<const v4i *>(yt + (size_t)(ks * NT * 2 + q) * 1024);
    };
    load_b(0, fb[0]);
#pragma unroll
    for (int ks = 0; ks < KS; ks++) {
        if (ks + 1 < KS) load_b(ks + 1, fb[(ks + 1) & 1]);
        // operand fragment: 1/(k - x_j) for the 16 nodes j = 64 ks + 16 (lane >> 4) + q, as (low limb | high limb << 8)
        const uint16_t *rn = rest_s + ks * 64 + (lane >> 4) * 16;
        const uint4 r0 = *reinterpret_cast<const uint4 *>(rn), r1 = *reinterpret_cast<const uint4 *>(rn + 8);
        const uint32_t rw[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
        uint32_t e[16];
#pragma unroll
        for (int q = 0; q < 16; q++) {
            const int xj = (int)((rw[q >> 1] >> (16 * (q & 1))) & 0xFFFFu);
            e[q] = tab_s[min((uint32_t)(kq - xj), (uint32_t)IA_TAB)]; // the clamp only matters for a malformed list (entry IA_TAB is 0)
        }
        uint32_t lo[4], hi[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const uint32_t t01 = e[4 * q] | (e[4 * q + 1] << 16), t23 = e[4 * q + 2] | (e[4 * q + 3] << 16);
            lo[q] = __builtin_amdgcn_perm(t23, t01, 0x06040200u);
            hi[q] = __builtin_amdgcn_perm(t23, t01, 0x07050301u);
        }
        const v4i a0 = {(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3]}, a1 = {(int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
        v4i(&bc)[2 * NT] = fb[ks & 1];
#pragma unroll
        for (int j = 0; j < NT; j++) {
            s0[j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a0, bc[2 * j], s0[j], 0, 0, 0);
            s1[j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a0, bc[2 * j + 1], s1[j], 0, 0, 0);
            s2[j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a1, bc[2 * j + 1], s2[j], 0, 0, 0);
            s1[j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a1, bc[2 * j], s1[j], 0, 0, 0);
        }
    }
    // D[row = evaluation point m0 + 4 (lane >> 4) + r][col = sharing 16 j + (lane & 15)];
    // |S0 + 64 S1 + 767 S2| < 2^29 for k <= 832, so adding 90 000 q makes it a positive u32
    const int kb = m0 + (lane >> 4) * 4;
    int node[4] = {-1, -1, -1, -1};
    uint32_t ell[4] = {0, 0, 0, 0}, xn[4] = {0, 0, 0, 0};
    if constexpr (!SET) {
#pragma unroll
        for (int r = 0; r < 4; r++)
            if (kb + r <= DEG) { node[r] = a.node_of[(size_t)b * 416 + kb + r]; ell[r] = a.ell[(size_t)b * 416 + kb + r]; }
#pragma unroll
        for (int r = 0; r < 4; r++)
            if (node[r] >= 0) xn[r] = rest_s[node[r]]; // node indices are < 407
    }
#pragma unroll
    for (int j = 0; j < NT; j++) {
        const int col = j * 16 + (lane & 15);
        if (col >= ncol) continue;
        uint32_t v[4];
#pragma unroll
        for (int r = 0; r < 4; r++) v[r] = gf_reduce_limbs(s0[j][r], s1[j][r], s2[j][r]);
        if constexpr (SET) {
            *reinterpret_cast<uint2 *>(g.out2 + ((size_t)b * ncol + col) * NSEC + kb) = make_uint2(v[0] | (v[1] << 16), v[2] | (v[3] << 16));
        } else {
            uint16_t *Pb = g.P + (size_t)b * g.proof_stride;
            const uint16_t *srow = Pb + (size_t)g.src_rows[col] * RS + NSEC;
            uint16_t *drow = Pb + (size_t)g.dst_rows[col] * RS;
#pragma unroll
            for (int r = 0; r < 4; r++)
                if (kb + r <= DEG) drow[kb + r] = node[r] >= 0 ? srow[xn[r]] : (uint16_t)gf_mul(ell[r], v[r]);
        }
    }
}

// blocks per proof: 7 (degree d: 407 points in groups of 64) + 4 (degree 2d: 256 points)
__global__ __launch_bounds__(256) void k_interp_apply(InterpApplyArgs g, int nproofs)
{
    __shared__ __attribute__((aligned(4))) uint16_t tab_s[IA_TAB + 2];
    __shared__ __attribute__((aligned(16))) uint16_t rest_s[IA_KS2 * 64];
    // XCD-aware order: the 11 workgroups of a proof read the same fragment tiles of weighted shares (108 KB)
    const int vid = xcd_virtual_id();
    const int b = vid / 11, m = vid % 11;
    if (b >= nproofs) return;
    if (m < 7) interp_apply_block<0>(g, b, m, tab_s, rest_s);
    else interp_apply_block<1>(g, b, m - 7, tab_s, rest_s);
}

// ---- checks ---------------------------------------------------------------------
// recomputed s+r / e+r shares against the unopened ones in the proof   mlwe_verifier.cpp:232-246

// relation checks on the opened columns   mlwe_verifier.cpp:273-284, :304-312, :365-376, :447-466
// One workgroup per (proof, i < K); every operand of the thread is in flight before the first comparison (round 6: one block per proof
// walked i and m with the loads of one step at a time -- a chain of ~10 round trips on a grid of one workgroup per CU, 16 us per 276 proofs).
__global__ __launch_bounds__(192) void k_check_opened(VerifyArgs v)
{
    const int t = threadIdx.x, b = blockIdx.x, i = blockIdx.y;
    if (t >= NOPEN) return;
    const RowMap &rm = v.rm;
    // recomputed sharings (every party) come from the row matrix, what the proof holds for the opened parties and what the
    // lincomb made of it from the opened matrix
    const uint16_t *Pb = v.P + (size_t)b * v.proof_stride + NSEC + v.opened[(size_t)b * v.sel_stride + t];
    const uint16_t *Ob = v.O + (size_t)b * v.o_stride + t;
    auto at = [&](int row) { return (uint32_t)Pb[(size_t)row * RS]; };
    auto op = [&](int row) { return (uint32_t)Ob[(size_t)row * OS]; };
    // op() values are the image's RAW u16 (and the raw NTT_r of k_lincomb): the reference compares and combines them with its
    // non-reducing gf3329_add / gf3329_sub, reproduced by ref_add_u16 / ref_sub_u16 (identical to gf_add / gf_sub on canonical values)
    constexpr int MAXE = 7; // 2 eta1 + 1, eta1 <= 3
    const int E = rm.E;
    const uint32_t p_nttsr = at(rm.nttsr + i), p_ntter = at(rm.ntter + i), p_nttasr = at(rm.nttasr + i), p_t = at(rm.t + i);
    const uint32_t o_ntts = op(rm.ntts + i), o_ntte = op(rm.ntte + i), o_nttr_s = op(rm.nttr + i), o_nttr_e = op(rm.nttr + rm.K + i);
    const uint32_t o_nttas = op(rm.nttas + i), o_nttar = op(rm.nttar + i), o_s = op(rm.s + i), o_e = op(rm.e + i);
    uint32_t p_seta[MAXE], p_eeta[MAXE], o_ssub[MAXE], o_esub[MAXE];
#pragma unroll
    for (int m = 0; m < MAXE; m++) { // (a slot at or beyond E reads the last gate's operands again and is not compared)
        const int mm = i * E + min(m, E - 1);
        p_seta[m] = at(rm.seta + mm); p_eeta[m] = at(rm.eeta + mm);
        o_ssub[m] = op(rm.ssub + mm); o_esub[m] = op(rm.esub + mm);
    }
    uint32_t bits = 0;
    if (o_ntts != ref_sub_u16(p_nttsr, o_nttr_s)) bits |= 1u << FB_NTT_S_E;          // :275
    if (o_ntte != ref_sub_u16(p_ntter, o_nttr_e)) bits |= 1u << FB_NTT_S_E;          // :279
    if (p_nttasr != ref_add_u16(o_nttas, o_nttar)) bits |= 1u << FB_A_SR;            // :306
    if (p_t != ref_add_u16(o_nttas, o_ntte)) bits |= 1u << FB_T_RELATION;            // :370
#pragma unroll
    for (int m = 0; m < MAXE; m++) {
        if (m < E && o_ssub[m] != ref_sub_u16(o_s, p_seta[m])) bits |= 1u << FB_SUB_ETA; // :451
        if (m < E && o_esub[m] != ref_sub_u16(o_e, p_eeta[m])) bits |= 1u << FB_SUB_ETA; // :459
    }
    if (bits) atomicOr(&v.fail[b], bits);
}

// interpolated packed secrets: t against the public key, range constants   mlwe_verifier.cpp:354-363, :418-429


// Opened list I of every proof, straight from the image: validated (range, duplicates), its complement
// (the unopened parties ascending), the opened parties ascending, and for the two interpolations how many
// opened parties lie below the first / the last node.  Replaces a device->host->device round trip.
// mlwe_verifier.cpp never validates I (it indexes with it); a malformed list can never be reproduced by
// the final Fiat-Shamir comparison, so it is rejected here and the kernels continue on I = 0..149.
__global__ __launch_bounds__(256) void k_opened_setup(const uint8_t *__restrict__ proof, size_t image_stride, uint32_t off_I,
                                                     uint16_t *__restrict__ I, uint16_t *__restrict__ rest,
                                                     uint16_t *__restrict__ isort, uint16_t *__restrict__ hrange,
                                                     size_t sel_stride, uint32_t *__restrict__ fail)
{
    __shared__ uint32_t cnt[NPARTY];
    __shared__ int bad, wave_used[4], node[3];
    const int t = threadIdx.x, b = blockIdx.x;
    for (int p = t; p < NPARTY; p += 256) cnt[p] = 0;
    if (t == 0) bad = 0;
    __syncthreads();
    uint32_t v = t;
    if (t < NOPEN) {
        v = reinterpret_cast<const uint16_t *>(proof + (size_t)b * image_stride + off_I)[t];
        if (v >= NPARTY || atomicAdd(&cnt[v], 1u) != 0) atomicOr(&bad, 1);
    }
    __syncthreads();
    const bool malformed = bad != 0;
    if (malformed) {
        __syncthreads();
        for (int p = t; p < NPARTY; p += 256) cnt[p] = p < NOPEN;
        v = t;
        __syncthreads();
    }
    if (t < NOPEN) I[(size_t)b * sel_stride + t] = (uint16_t)v;
    // parties 6t..6t+5 per thread; exclusive scan of the opened counts over the block
    int u[6], mine = 0;
#pragma unroll
    for (int k = 0; k < 6; k++) {
        const int p = 6 * t + k;
        u[k] = p < NPARTY && cnt[p] != 0;
        mine += u[k];
    }
    int incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(incl, d, 64);
        if ((t & 63) >= d) incl += o;
    }
    if ((t & 63) == 63) wave_used[t >> 6] = incl;
    __syncthreads();
    int used_before = incl - mine;
    for (int w = 0; w < (t >> 6); w++) used_before += wave_used[w];
#pragma unroll
    for (int k = 0; k < 6; k++) {
        const int p = 6 * t + k;
        if (p >= NPARTY) break;
        if ((p & 63) == 0) I[(size_t)b * sel_stride + SEL_WIN + (p >> 6)] = (uint16_t)(p - used_before); // unopened below window p/64
        if (u[k]) {
            isort[(size_t)b * sel_stride + used_before] = (uint16_t)p;
            used_before++;
        } else {
            const int j = p - used_before;
            rest[(size_t)b * sel_stride + j] = (uint16_t)p;
            if (j == 0) node[0] = p;
            if (j == DEG) node[1] = p;
            if (j == DEG2) node[2] = p;
        }
    }
    __syncthreads();
    if (t == 0) {
        // every party below the first unopened one is opened; below the D-th unopened one there are D unopened
        uint16_t *hr = hrange + (size_t)b * 4;
        hr[0] = (uint16_t)node[0];
        hr[1] = (uint16_t)(node[1] - DEG);
        hr[2] = (uint16_t)node[0];
        hr[3] = (uint16_t)(node[2] - DEG2);
        fail[b] = malformed ? 1u << FB_MALFORMED : 0u;
        I[(size_t)b * sel_stride + SEL_WIN + NWIN] = (uint16_t)NREST;
    }
}

// The four checks that follow the interpolation GEMMs in one launch (blockIdx.x selects the role):
// [0,6) recomputed s+r / e+r shares vs the proof's (:232-246); 6: t vs pk and the eta constants (:303-324);
// then nu blocks each for the interpolated and the reconstructed u secrets, which must vanish (:523-556).
__global__ __launch_bounds__(256) void k_check_batch(VerifyArgs v, const uint16_t *__restrict__ t_pk,
                                                     const uint16_t *__restrict__ u1, const uint16_t *__restrict__ u2, int nu)
{
    constexpr int NB_REST = (NREST + 255) / 256;
    const int role = blockIdx.x, b = blockIdx.y, t = threadIdx.x;
    const RowMap &rm = v.rm;
    if (role < NB_REST) {
        const int i = role * 256 + t;
        if (i >= NREST) return;
        const uint16_t *Pb = v.P + (size_t)b * v.proof_stride + NSEC + v.rest[(size_t)b * v.sel_stride + i];
        bool bad = false;
        uint32_t x[4 * MAXK]; // every operand in flight before the first comparison (a slot at or beyond K repeats row K - 1)
#pragma unroll
        for (int r = 0; r < MAXK; r++) {
            const int rr = min(r, rm.K - 1);
            x[4 * r] = Pb[(size_t)(rm.sr + rr) * RS]; x[4 * r + 1] = Pb[(size_t)(rm.sr_in + rr) * RS];
            x[4 * r + 2] = Pb[(size_t)(rm.er + rr) * RS]; x[4 * r + 3] = Pb[(size_t)(rm.er_in + rr) * RS];
        }
#pragma unroll
        for (int r = 0; r < MAXK; r++) bad |= x[4 * r] != x[4 * r + 1] || x[4 * r + 2] != x[4 * r + 3];
        if (bad) atomicOr(&v.fail[b], 1u << FB_SR_ER_SHARES);
    } else if (role == NB_REST) {
        const uint16_t *Pb = v.P + (size_t)b * v.proof_stride + t;
        uint32_t bits = 0;
        constexpr int MAXE = 7;
        const int K = rm.K, E = rm.E;
        uint32_t pt[MAXK], tk[MAXK];
#pragma unroll
        for (int i = 0; i < MAXK; i++) {
            const int ii = min(i, K - 1);
            pt[i] = Pb[(size_t)(rm.t + ii) * RS];
            tk[i] = t_pk[((size_t)b * K + ii) * 256 + t];
        }
#pragma unroll
        for (int i = 0; i < MAXK; i++) bits |= pt[i] != tk[i] ? 1u << FB_T_PK : 0u;
        // the 2 K E range constants: one gate index m at a time with all 2 K rows of it in flight
#pragma unroll
        for (int m = 0; m < MAXE; m++) {
            const int mm = min(m, E - 1);
            const uint32_t c = gf_encode(mm - v.eta1);
            uint32_t y[2 * MAXK];
#pragma unroll
            for (int i = 0; i < MAXK; i++) {
                const int ii = min(i, K - 1);
                y[2 * i] = Pb[(size_t)(rm.seta + ii * E + mm) * RS];
                y[2 * i + 1] = Pb[(size_t)(rm.eeta + ii * E + mm) * RS];
            }
#pragma unroll
            for (int i = 0; i < 2 * MAXK; i++) bits |= y[i] != c ? 1u << FB_ETA_CONST : 0u;
        }
        if (bits) atomicOr(&v.fail[b], bits);
    } else {
        const int q = role - NB_REST - 1; // row of u1 (q < nu) or of u2
        const uint16_t *x = (q < nu ? u1 : u2) + ((size_t)b * nu + (q < nu ? q : q - nu)) * 256;
        if (x[t] != 0) atomicOr(&v.fail[b], 1u << (q < nu ? FB_U_INTERP : FB_U_RECON));
    }
}

// ---- launchers --------------------------------------------------------------------
hipError_t launch_check_batch(const VerifyArgs &v, const uint16_t *t_pk, const uint16_t *u1, const uint16_t *u2, int nu, int nproofs,
                              hipStream_t st)
{
    hipLaunchKernelGGL(k_check_batch, dim3((NREST + 255) / 256 + 1 + 2 * nu, nproofs), dim3(256), 0, st, v, t_pk, u1, u2, nu);
    return hipGetLastError();
}
size_t interp_y_bytes(int set) { return set ? IA_Y2_BYTES : IA_Y1_BYTES; }
int interp_table_len() { return IA_TAB + 2; }
int interp_table_off() { return IA_OFF; }
hipError_t launch_gather_frags(const uint16_t *P, size_t proof_stride, const int16_t *rows1, int nrows1, uint8_t *out1,
                               const int16_t *rows2, int nrows2, uint8_t *out2, const uint16_t *rest, int sel_stride,
                               const uint16_t *w, int nproofs, hipStream_t st)
{
    if (nrows1 > 16 * IA_NT1 || nrows2 > 16 * IA_NT2) return hipErrorInvalidValue;
    const int threads = IA_NT1 * 16 * IA_KS1 * 4 + IA_NT2 * 16 * IA_KS2 * 4;
    hipLaunchKernelGGL(k_gather_frags, dim3((threads + 255) / 256, nproofs), dim3(256), 0, st, P, proof_stride, rows1, nrows1, out1, rows2,
                       nrows2, out2, rest, sel_stride, w);
    return hipGetLastError();
}
hipError_t launch_opened_setup(const uint8_t *proof, size_t image_stride, size_t off_I, uint16_t *I, uint16_t *rest, uint16_t *isort,
                               uint16_t *hrange, size_t sel_stride, uint32_t *fail, int nproofs, hipStream_t st)
{
    hipLaunchKernelGGL(k_opened_setup, dim3(nproofs), dim3(256), 0, st, proof, image_stride, (uint32_t)off_I, I, rest, isort, hrange, sel_stride, fail);
    return hipGetLastError();
}
hipError_t launch_disassemble(const VerifyArgs &v, const FieldDesc *fields, const FieldPlan &plan, const int16_t *rowtab,
                              const uint8_t *proof, size_t image_stride, size_t off_tcomm, size_t off_comm,
                              uint8_t *dig1, uint8_t *dig2, const GateOffsets &go, int nproofs, hipStream_t st)
{
    hipLaunchKernelGGL(k_disassemble_fields, dim3(plan.nrest * NWIN + plan.nopen * ((NOPEN + 63) / 64) + (NREST * 4 + 63) / 64 + (NOPEN + 63) / 64, nproofs), dim3(64), 0,
                       st, v, fields, plan, rowtab, proof, image_stride, (uint32_t)off_tcomm, (uint32_t)off_comm, dig1, dig2, go);
    return hipGetLastError();
}
hipError_t launch_interp_setup(const InterpArgs &a, int nproofs, hipStream_t st)
{
    hipLaunchKernelGGL(k_interp_setup, dim3(8 * nproofs), dim3(256), 0, st, a, nproofs);
    return hipGetLastError();
}
hipError_t launch_interp_apply(const InterpArgs &a, uint16_t *P, size_t proof_stride, const int16_t *src_rows, const int16_t *dst_rows,
                               int n1, const uint8_t *y1, int n2, const uint8_t *y2, uint16_t *out2, int nproofs, hipStream_t st)
{
    if (n1 > 16 * IA_NT1 || n2 > 16 * IA_NT2 || n1 < 1 || n2 < 1) return hipErrorInvalidValue;
    InterpApplyArgs g{};
    g.ia = a;
    g.P = P;
    g.proof_stride = proof_stride;
    g.src_rows = src_rows;
    g.dst_rows = dst_rows;
    g.n1 = n1;
    g.y1 = y1;
    g.n2 = n2;
    g.y2 = y2;
    g.out2 = out2;
    hipLaunchKernelGGL(k_interp_apply, dim3((11 * nproofs + 7) / 8 * 8), dim3(256), 0, st, g, nproofs);
    return hipGetLastError();
}
hipError_t launch_check_opened(const VerifyArgs &v, int nproofs, hipStream_t st)
{
    hipLaunchKernelGGL(k_check_opened, dim3(nproofs, v.rm.K), dim3(192), 0, st, v);
    return hipGetLastError();
}

} // namespace kosk
