// The reference's second-level entry points (mlwe_prover.hpp:77-99, mlwe_verifier.hpp:14-15) on host structs:
//   prepare_randomness   mlwe_prover.cpp:4-39     -> mpcith_randomness  (f, NTT f and their 2M sharings)
//   prepare_range_proof  mlwe_prover.cpp:41-59    -> mpcith_range_proof (sharings of the 2 eta1 + 1 range constants)
//   prove                mlwe_prover.cpp:81-538   with the two structs and an mlwe_inst as inputs
//   verify               mlwe_verifier.cpp:4-686  with A and t taken from an mlwe_inst instead of a packed pk
// (SURVEY.md 8(b) row 2, and the persisted offline/online split of 8(f3).)  The compute is the same GPU
// pipeline as kyber_verifiable_keygen; only the sharing front is cut at the struct boundary: the offline rows are
// downloaded into / uploaded from the reference's struct layouts:
//   share_vec          { size_t len; u16 share_x[1454]; u16 share_y[1454]; }                    ss.hpp:33-37
//   mpcith_randomness  { u16 f[M][256]; u16 NTT_f[M][256]; share_vec f_shares[M], NTT_f_shares[M]; }   mlwe_prover.hpp:39-44
//   mpcith_range_proof { share_vec s_eta_shares[K][2 eta1 + 1], e_eta_shares[K][2 eta1 + 1]; }         :46-49
//   mlwe_inst          { polyvec A[K], t; polyvec s, e; }  = i16 [K*K + 3K][256]                       :34-37
#include <cstring>
#include <vector>

#include "kosk_ctx.hpp"
#include "kosk_math.hpp"

namespace kosk {

#define HIPCHK(x) KOSK_HIPCHK(x)

static constexpr size_t SHARE_VEC_BYTES = 8 + 2 * 2 * NPARTY; // 5824

size_t randomness_bytes(const Params &P) { return (size_t)P.M * (2 * 512 + 2 * SHARE_VEC_BYTES); }
size_t range_proof_bytes(const Params &P) { return (size_t)2 * P.K * P.E * SHARE_VEC_BYTES; }
size_t mlwe_inst_bytes(const Params &P) { return (size_t)(P.K * P.K + 3 * P.K) * 512; }

// randomness for the fresh sharings [s0, s1) (and the M seeds when `seeds`) into the full-size tape image of every
// proof, from compact per-proof tapes or from the randombytes callback in the reference's call order
static int fill_tape_part(Ctx &c, int n, bool seeds, int s0, int s1, const uint8_t *tapes, size_t tape_stride)
{
    const Params &P = c.P;
    const size_t seed_bytes = seeds ? (size_t)32 * P.M : 0, slice_bytes = (size_t)302 * (s1 - s0);
    if (tapes && tape_stride < seed_bytes + slice_bytes) { c.err = "tape_stride smaller than the randomness this call consumes"; return -1; }
    for (int b = 0; b < n; b++) {
        uint8_t *tp = c.h_tape + (size_t)b * c.tape_stride;
        uint8_t *dseed = tp + 64, *dslice = tp + 64 + 32 * P.M + (size_t)302 * s0;
        if (tapes) {
            const uint8_t *src = tapes + (size_t)b * tape_stride;
            if (seeds) memcpy(dseed, src, seed_bytes);
            memcpy(dslice, src + seed_bytes, slice_bytes);
        } else {
            auto draw = [&](uint8_t *dst, size_t len) {
                if (c.rb) c.rb(c.rb_user, dst, len);
                else os_randombytes(dst, len);
            };
            if (seeds)
                for (int i = 0; i < P.M; i++) draw(dseed + 32 * i, 32);           // mlwe_prover.cpp:9
            for (int i = 0; i < s1 - s0; i++) draw(dslice + (size_t)302 * i, 302); // ss.cpp:5
        }
    }
    HIPCHK(hipMemcpyAsync(c.d_tape, c.h_tape, (size_t)n * c.tape_stride, hipMemcpyHostToDevice, c.stream));
    c.tape_cur = c.d_tape;
    c.tape_cur_stride = c.tape_stride;
    return 0;
}

static void put_share_vec(uint8_t *dst, const uint16_t *row)
{
    const uint64_t len = NPARTY; // ss.cpp:32 sets len = MPCITH_N
    memcpy(dst, &len, 8);
    uint16_t *x = reinterpret_cast<uint16_t *>(dst + 8), *y = x + NPARTY;
    for (int p = 0; p < NPARTY; p++) x[p] = (uint16_t)(NSEC + p); // ss.cpp:9,30
    memcpy(y, row + NSEC, 2 * NPARTY);
}

int prepare_randomness(Ctx &c, int n, const uint8_t *tapes, size_t tape_stride, uint8_t *out)
{
    if (n < 1 || n > c.call_cap) { c.err = "batch size out of range"; return -1; }
    HIPCHK(hipSetDevice(c.device));
    const Params &P = c.P;
    const RowMap &rm = c.rm;
    const int M = P.M;
    if (fill_tape_part(c, n, true, 0, 2 * M, tapes, tape_stride)) return -1;
    if (issue_sharing_front(c, n, FRONT_RANDOMNESS)) return -1;
    HIPCHK(stream_sync(c));
    if (rm.tf != rm.f + M) { c.err = "internal: f / NTT f rows not adjacent"; return -1; }
    std::vector<uint16_t> rows((size_t)2 * M * RS);
    for (int b = 0; b < n; b++) {
        HIPCHK(hipMemcpyAsync(rows.data(), c.d_P + (size_t)b * c.proof_stride + (size_t)rm.f * RS, rows.size() * 2, hipMemcpyDeviceToHost, c.stream));
        HIPCHK(hipStreamSynchronize(c.stream));
        uint8_t *o = out + (size_t)b * randomness_bytes(P);
        uint8_t *of = o, *ontt = o + (size_t)M * 512, *ofs = o + (size_t)M * 1024, *onfs = ofs + (size_t)M * SHARE_VEC_BYTES;
        for (int i = 0; i < M; i++) {
            memcpy(of + (size_t)i * 512, &rows[(size_t)i * RS], 512);
            memcpy(ontt + (size_t)i * 512, &rows[(size_t)(M + i) * RS], 512);
            put_share_vec(ofs + (size_t)i * SHARE_VEC_BYTES, &rows[(size_t)i * RS]);
            put_share_vec(onfs + (size_t)i * SHARE_VEC_BYTES, &rows[(size_t)(M + i) * RS]);
        }
    }
    return 0;
}

int prepare_range_proof(Ctx &c, int n, const uint8_t *tapes, size_t tape_stride, uint8_t *out)
{
    if (n < 1 || n > c.call_cap) { c.err = "batch size out of range"; return -1; }
    HIPCHK(hipSetDevice(c.device));
    const Params &P = c.P;
    const RowMap &rm = c.rm;
    const int KE = P.K * P.E;
    if (fill_tape_part(c, n, false, 2 * P.M, 2 * P.M + 2 * KE, tapes, tape_stride)) return -1;
    if (issue_sharing_front(c, n, FRONT_RANGE)) return -1;
    HIPCHK(stream_sync(c));
    if (rm.eeta != rm.seta + KE) { c.err = "internal: eta rows not adjacent"; return -1; }
    std::vector<uint16_t> rows((size_t)2 * KE * RS);
    for (int b = 0; b < n; b++) {
        HIPCHK(hipMemcpyAsync(rows.data(), c.d_P + (size_t)b * c.proof_stride + (size_t)rm.seta * RS, rows.size() * 2, hipMemcpyDeviceToHost, c.stream));
        HIPCHK(hipStreamSynchronize(c.stream));
        uint8_t *o = out + (size_t)b * range_proof_bytes(P);
        for (int r = 0; r < 2 * KE; r++) put_share_vec(o + (size_t)r * SHARE_VEC_BYTES, &rows[(size_t)r * RS]); // s rows [i][j], then e rows
    }
    return 0;
}

static void get_share_vec(uint16_t *row, const uint8_t *src)
{
    memcpy(row + NSEC, src + 8 + 2 * NPARTY, 2 * NPARTY); // share_y; share_x is implied (ss.cpp:9)
}

// A (canonical), s, e of n mlwe_inst images -> the device key buffers the prover reads
static int upload_inst(Ctx &c, int n, const uint8_t *inst, bool with_se, bool with_t)
{
    const Params &P = c.P;
    const int K = P.K;
    const size_t ib = mlwe_inst_bytes(P);
    std::vector<int16_t> A((size_t)n * c.key_stride), se((size_t)n * c.se_stride);
    std::vector<uint16_t> t((size_t)n * K * 256);
    for (int b = 0; b < n; b++) {
        const int16_t *src = reinterpret_cast<const int16_t *>(inst + (size_t)b * ib);
        for (size_t i = 0; i < (size_t)K * K * 256; i++) A[(size_t)b * c.key_stride + i] = (int16_t)(((int)src[i] % Q + Q) % Q);
        const int16_t *ts = src + (size_t)K * K * 256, *ss = ts + (size_t)K * 256; // t, then s, e (mlwe_prover.hpp:34-37)
        for (int i = 0; i < K * 256; i++) t[(size_t)b * K * 256 + i] = (uint16_t)(((int)ts[i] % Q + Q) % Q); // encode_to_gf3329
        memcpy(&se[(size_t)b * c.se_stride], ss, (size_t)2 * K * 512);
    }
    HIPCHK(hipMemcpyAsync(c.d_A, A.data(), A.size() * 2, hipMemcpyHostToDevice, c.stream));
        HIPCHK(hipStreamSynchronize(c.stream));
    if (with_se) HIPCHK(hipMemcpyAsync(c.d_se, se.data(), se.size() * 2, hipMemcpyHostToDevice, c.stream));
        HIPCHK(hipStreamSynchronize(c.stream));
    if (with_t) HIPCHK(hipMemcpyAsync(c.d_t, t.data(), t.size() * 2, hipMemcpyHostToDevice, c.stream));
        HIPCHK(hipStreamSynchronize(c.stream));
    return 0;
}

int prove_prepared(Ctx &c, int n, const uint8_t *inst, const uint8_t *rand_in, const uint8_t *range_in, const uint8_t *tapes,
                   size_t tape_stride, uint8_t *pi)
{
    if (n < 1 || n > c.call_cap) { c.err = "batch size out of range"; return -1; }
    HIPCHK(hipSetDevice(c.device));
    const Params &P = c.P;
    const RowMap &rm = c.rm;
    const int M = P.M, K = P.K, E = P.E, KE = K * E;
    if (upload_inst(c, n, inst, true, false)) return -1;
    // offline rows: secrets at x < 256, party shares at x = 256.., the rest of each row is produced by nothing else
    std::vector<uint16_t> rows((size_t)2 * M * RS), erows((size_t)2 * KE * RS);
    for (int b = 0; b < n; b++) {
        const uint8_t *r = rand_in + (size_t)b * randomness_bytes(P);
        const uint8_t *rf = r, *rntt = r + (size_t)M * 512, *rfs = r + (size_t)M * 1024, *rnfs = rfs + (size_t)M * SHARE_VEC_BYTES;
        std::fill(rows.begin(), rows.end(), 0);
        for (int i = 0; i < M; i++) {
            memcpy(&rows[(size_t)i * RS], rf + (size_t)i * 512, 512);
            memcpy(&rows[(size_t)(M + i) * RS], rntt + (size_t)i * 512, 512);
            get_share_vec(&rows[(size_t)i * RS], rfs + (size_t)i * SHARE_VEC_BYTES);
            get_share_vec(&rows[(size_t)(M + i) * RS], rnfs + (size_t)i * SHARE_VEC_BYTES);
        }
        for (uint16_t v : rows)
            if (v >= Q) { c.err = "mpcith_randomness holds a non-canonical value"; return -1; }
        HIPCHK(hipMemcpyAsync(c.d_P + (size_t)b * c.proof_stride + (size_t)rm.f * RS, rows.data(), rows.size() * 2, hipMemcpyHostToDevice, c.stream));
        HIPCHK(hipStreamSynchronize(c.stream));
        const uint8_t *g = range_in + (size_t)b * range_proof_bytes(P);
        std::fill(erows.begin(), erows.end(), 0);
        for (int q = 0; q < 2 * KE; q++) {
            const int m = q % E; // constant m - eta1 at every packed position (mlwe_prover.cpp:47-50)
            const uint16_t cst = (uint16_t)gf_encode((int16_t)(m - P.eta1));
            for (int x = 0; x < NSEC; x++) erows[(size_t)q * RS + x] = cst;
            get_share_vec(&erows[(size_t)q * RS], g + (size_t)q * SHARE_VEC_BYTES);
        }
        for (uint16_t v : erows)
            if (v >= Q) { c.err = "mpcith_range_proof holds a non-canonical value"; return -1; }
        HIPCHK(hipMemcpyAsync(c.d_P + (size_t)b * c.proof_stride + (size_t)rm.seta * RS, erows.data(), erows.size() * 2, hipMemcpyHostToDevice, c.stream));
        HIPCHK(hipStreamSynchronize(c.stream));
    }
    if (fill_tape_part(c, n, false, 2 * M + 2 * KE, P.nfresh, tapes, tape_stride)) return -1;
    if (prove_resident(c, n, true)) return -1;
    return fetch_proofs(c, n, pi);
}

int stage_verifier_inst(Ctx &c, int n, const uint8_t *pi, const uint8_t *inst)
{
    if (n < 1 || n > c.call_cap) { c.err = "batch size out of range"; return -1; }
    HIPCHK(hipSetDevice(c.device));
    if (ensure_verify_workspace(c)) return -1;
    const Params &P = c.P;
    parallel_for(c.pool, n, c.nthreads, [&](int b) { memcpy(c.h_proof + (size_t)b * c.image_stride, pi + (size_t)b * P.proof_bytes, P.proof_bytes); });
    HIPCHK(hipMemcpyAsync(c.d_proof, c.h_proof, (size_t)n * c.image_stride, hipMemcpyHostToDevice, c.stream));
    HIPCHK(stream_sync(c));
    return upload_inst(c, n, inst, false, true);
}

} // namespace kosk
