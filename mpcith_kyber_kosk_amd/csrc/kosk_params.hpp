// Parameter sets, HBM row map and proof wire layout shared by host and device.
//
// Reference: params.hpp:12-36 (MPCITH_N/T/L/K/V), kyber/params.h:29-53,
// ss.hpp:15-31 (evaluation-point picture), mlwe_prover.hpp:57-75 (mpcith_proof).
//
// HBM data model.  The reference keeps one `share_vec` per sharing
// (ss.hpp:33-37) and a per-party AoS view (mlwe_prover.hpp:85-94).  Here one
// sharing is ONE ROW holding the sharing polynomial's value at every
// evaluation point x = 0..1709 as canonical u16: x < 256 are the packed
// secrets, x = 256+p is party p's share (exactly share_x of ss.cpp:9,30).
// Rows are RS = 1728 u16 apart (27 x 128 B), so a 64-party column block of any
// row is one aligned 128-byte line and the 407 inputs of the Lagrange
// expansion are the first 407 entries of the very row it completes.
#pragma once
#include <cstddef>
#include <cstdint>

#if defined(__HIPCC__)
#define KOSK_HD __host__ __device__
#else
#define KOSK_HD
#endif

namespace kosk {

constexpr int Q = 3329;
constexpr int NSEC = 256;             // KYBER_N packed secrets
constexpr int NPARTY = 1454;          // MPCITH_N
constexpr int NOPEN = 150;            // MPCITH_T
constexpr int NREST = NPARTY - NOPEN; // 1304
constexpr int NCHK = 70;              // MPCITH_K
constexpr int DEG = 406;              // DEG_D, ss.hpp:56
constexpr int DEG2 = 812;             // DEG_2D
constexpr int NPTS = NSEC + NPARTY;   // 1710 evaluation points
constexpr int RS = 1728;              // row stride in u16
constexpr int XLEN = DEG + 1;         // 407 expansion inputs
constexpr int EXP_OFF = 384;          // first point written by the expand GEMM (party 128)
constexpr int EXP_M = RS - EXP_OFF;   // 1344 = 21 x 64 outputs per row
constexpr int NFIELDS = 24;
// verifier: the values of the 150 OPENED parties of every row, in the order of the proof's list I, live in a compact
// second matrix O[proof][row][OS] (same row ids as the row matrix): what the proof image holds party-major lands there
// with coalesced writes, and the opened-party steps (lincomb, view hash, relation checks) read it without column gathers
constexpr int OS = 160;
// an opened-list row [proof][sel_stride] holds I in [0,150) and, from SEL_WIN on, for each aligned window of 64
// parties w = 0..23 the number of unopened parties below 64w (so window w owns complement entries [win[w], win[w+1]))
constexpr int SEL_WIN = 160, NWIN = (NPARTY + 63) / 64;
// (prover, round 5) behind the window boundaries: the OPENED parties in ascending order (SEL_OSORT, 150 entries) and the position of each
// of them in the list I (SEL_OPOS): window w of the grouped image kernel owns the entries [min(64 w, N) - win[w], min(64 (w + 1), N) - win[w + 1])
constexpr int SEL_OSORT = 192, SEL_OPOS = 352;
constexpr int MAXK = 4, MAXM = 79, MAXJ = NCHK + 2 * MAXK;

struct Params {
    int K, eta1, M, V, E, Z, J; // J = 70 + 2K lincomb outputs
    size_t pk_bytes, sk_bytes, proof_bytes, tape_bytes;
    int tape_calls, nfresh; // nfresh = number of 302-byte tape slices
    int tcomm_words, view_words; // u16 words hashed per party (without / with the 32-byte prefix)
    size_t off[NFIELDS], size[NFIELDS];
};

// field ids in mpcith_proof declaration order (mlwe_prover.hpp:57-75)
enum Field {
    F_F = 0, F_NTTF, F_BETA, F_GAMMA, F_TCOMM, F_I, F_S, F_E, F_T, F_NTTS, F_NTTE, F_NTTAR,
    F_NTTAS, F_SR, F_ER, F_SETA, F_EETA, F_SSUB, F_ESUB, F_ZS, F_ZE, F_US, F_UE, F_COMM
};

inline bool make_params(int K, Params &p)
{
    if (K < 2 || K > 4) return false;
    p.K = K;
    p.eta1 = (K == 2) ? 3 : 2;
    p.V = 2 * K;
    p.M = NCHK + p.V + 1;
    p.J = NCHK + p.V;
    p.E = 2 * p.eta1 + 1;
    p.Z = 2 * p.eta1;
    p.pk_bytes = 384u * K + 32;
    p.sk_bytes = 384u * K + p.pk_bytes + 64;
    const size_t T = NOPEN, R = NREST, M = p.M, k = K, E = p.E, Z = p.Z;
    const size_t sz[NFIELDS] = {T * M * 2, T * M * 2, R * NCHK * 2, R * NCHK * 2, R * 32, T * 2,
                                T * k * 2, T * k * 2, R * k * 2, T * k * 2, T * k * 2, T * k * 2, T * k * 2,
                                R * k * 2, R * k * 2, R * k * E * 2, R * k * E * 2, T * k * E * 2, T * k * E * 2,
                                T * k * Z * 2, T * k * Z * 2, R * k * Z * 2, R * k * Z * 2, R * 32};
    size_t o = 0;
    for (int i = 0; i < NFIELDS; i++) {
        p.off[i] = o;
        p.size[i] = sz[i];
        o += sz[i];
    }
    p.proof_bytes = o;
    p.nfresh = 2 * p.M + 2 * K * p.E + 2 * K + K + 2 * K * p.Z;
    p.tape_calls = 1 + p.M + p.nfresh;
    p.tape_bytes = 64 + 32u * p.M + 302u * p.nfresh;
    p.tcomm_words = 2 * (K + p.M);
    p.view_words = (6 + 4 * p.Z) * K + 2 * p.M;
    return true;
}

// Row ids inside one proof's row matrix.  Rows [0, RV) are, in order, exactly
// the u16 words of the reference's view hash input after its 32-byte prefix
// (mlwe_prover.cpp:397-443); rows [0, 2K+2M) are the Tcomm hash input
// (mlwe_prover.cpp:116-127).  Everything else follows.
struct RowMap {
    int K, M, E, Z;
    int s, e, f, tf, beta0, gamma0, sr, er, gate, RV;
    int beta1, gamma1, r, nttr, seta, eeta, nttsr, ntter, nttasr, nttas;
    int ntts, ntte, nttar, t, ssub, esub;
    int shat; // NTT(s) secrets (K rows, only x < 256 used)
    int sr_in, er_in, t_in, seta_in, eeta_in; // verifier: unopened values as given in the proof (before re-computation)
    int nrows;

    KOSK_HD int beta(int j) const { return j < K ? beta0 + j : beta1 + (j - K); }
    KOSK_HD int gamma(int j) const { return j < K ? gamma0 + j : gamma1 + (j - K); }
    KOSK_HD int zs(int i, int k) const { return gate + i * 4 * Z + k; }
    KOSK_HD int ze(int i, int k) const { return gate + i * 4 * Z + Z + k; }
    KOSK_HD int us(int i, int k) const { return gate + i * 4 * Z + 2 * Z + k; }
    KOSK_HD int ue(int i, int k) const { return gate + i * 4 * Z + 3 * Z + k; }
};

inline RowMap make_rowmap(const Params &p)
{
    RowMap r{};
    const int K = p.K, M = p.M, E = p.E, Z = p.Z;
    r.K = K; r.M = M; r.E = E; r.Z = Z;
    int n = 0;
    auto take = [&](int cnt) { int b = n; n += cnt; return b; };
    r.s = take(K); r.e = take(K); r.f = take(M); r.tf = take(M);
    r.beta0 = take(K); r.gamma0 = take(K); r.sr = take(K); r.er = take(K);
    r.gate = take(4 * Z * K);
    r.RV = n;
    r.beta1 = take(NCHK - K); r.gamma1 = take(NCHK - K);
    r.r = take(2 * K); r.nttr = take(2 * K);
    r.seta = take(K * E); r.eeta = take(K * E);
    r.nttsr = take(K); r.ntter = take(K); r.nttasr = take(K); r.nttas = take(K);
    r.ntts = take(K); r.ntte = take(K); r.nttar = take(K); r.t = take(K);
    r.ssub = take(K * E); r.esub = take(K * E);
    r.shat = take(K);
    r.sr_in = take(K); r.er_in = take(K); r.t_in = take(K);
    r.seta_in = take(K * E); r.eeta_in = take(K * E);
    r.nrows = n;
    return r;
}

} // namespace kosk
