// kosk_compat.hpp -- source-compatible C++ face of the reference's kosk.hpp (kosk.hpp:13-24) on top of
// the C ABI in kosk_mi355x.h.  A caller written against the reference (e.g. main.cpp:66-94) recompiles
// unchanged against this header:  -DKYBER_K=2|3|4 -lkosk_mi355x  replaces the reference's own
// kosk.cpp / mlwe_prover.cpp / mlwe_verifier.cpp / ss.cpp / utils/*.c and the NTL dependency.
#ifndef KOSK_COMPAT_HPP
#define KOSK_COMPAT_HPP

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "kosk_mi355x.h"

#ifndef KYBER_K
#define KYBER_K 2 /* params.hpp:8-10 */
#endif
#define KYBER_N 256
#define KYBER_Q 3329
#define KYBER_SYMBYTES 32
#define KYBER_POLYBYTES 384
#define KYBER_POLYVECBYTES (KYBER_K * KYBER_POLYBYTES)
#define KYBER_PUBLICKEYBYTES (KYBER_POLYVECBYTES + KYBER_SYMBYTES)                              /* kyber/params.h:49 */
#define KYBER_SECRETKEYBYTES (KYBER_POLYVECBYTES + KYBER_PUBLICKEYBYTES + 2 * KYBER_SYMBYTES)   /* kyber/params.h:51 */
#if KYBER_K == 2
#define KYBER_ETA1 3
#else
#define KYBER_ETA1 2
#endif
#define MPCITH_N 1454 /* params.hpp:13 */
#define MPCITH_T 150
#define MPCITH_K 70
#define MPCITH_V (KYBER_K * 2)
/* sizeof(mpcith_proof), mlwe_prover.hpp:30,57-75 */
#define MPCITH_PROOF_SIZE                                                                                           \
    ((size_t)2 * (2 * MPCITH_T * (MPCITH_K + MPCITH_V + 1) + 2 * (MPCITH_N - MPCITH_T) * MPCITH_K + MPCITH_T +     \
                  6 * MPCITH_T * KYBER_K + 3 * (MPCITH_N - MPCITH_T) * KYBER_K +                                    \
                  2 * (MPCITH_N - MPCITH_T) * KYBER_K * (2 * KYBER_ETA1 + 1) + 2 * MPCITH_T * KYBER_K * (2 * KYBER_ETA1 + 1) + \
                  2 * MPCITH_T * KYBER_K * 2 * KYBER_ETA1 + 2 * (MPCITH_N - MPCITH_T) * KYBER_K * 2 * KYBER_ETA1) + \
     (size_t)2 * (MPCITH_N - MPCITH_T) * KYBER_SYMBYTES)

typedef struct { int16_t coeffs[KYBER_N]; } poly;          /* kyber/poly.h */
typedef struct { poly vec[KYBER_K]; } polyvec;             /* kyber/polyvec.h */
typedef struct { polyvec A[KYBER_K], t; polyvec s, e; } mlwe_inst; /* mlwe_prover.hpp:34-37 */
typedef struct {
    uint8_t pk[KYBER_PUBLICKEYBYTES];
    uint8_t sk[KYBER_SECRETKEYBYTES];
} kyber_keypair; /* kosk.hpp:13-16 */

/* ---- second-level types (ss.hpp:33-37, mlwe_prover.hpp:39-75): same members, same layout ---- */
typedef struct {
    size_t len;
    uint16_t share_x[MPCITH_N];
    uint16_t share_y[MPCITH_N];
} share_vec;
typedef struct {
    uint16_t f[MPCITH_K + MPCITH_V + 1][KYBER_N];
    uint16_t NTT_f[MPCITH_K + MPCITH_V + 1][KYBER_N];
    share_vec f_shares[MPCITH_K + MPCITH_V + 1];
    share_vec NTT_f_shares[MPCITH_K + MPCITH_V + 1];
} mpcith_randomness;
typedef struct {
    share_vec s_eta_shares[KYBER_K][KYBER_ETA1 * 2 + 1];
    share_vec e_eta_shares[KYBER_K][KYBER_ETA1 * 2 + 1];
} mpcith_range_proof;
typedef struct {
    uint16_t f_shares[MPCITH_T][MPCITH_K + MPCITH_V + 1], NTT_f_shares[MPCITH_T][MPCITH_K + MPCITH_V + 1];
    uint16_t beta_shares[MPCITH_N - MPCITH_T][MPCITH_K], gamma_shares[MPCITH_N - MPCITH_T][MPCITH_K];
    uint8_t Tcomm[MPCITH_N - MPCITH_T][KYBER_SYMBYTES];
    uint16_t I[MPCITH_T];
    uint16_t s_shares[MPCITH_T][KYBER_K], e_shares[MPCITH_T][KYBER_K], t_shares[MPCITH_N - MPCITH_T][KYBER_K];
    uint16_t NTT_s_shares[MPCITH_T][KYBER_K], NTT_e_shares[MPCITH_T][KYBER_K];
    uint16_t NTT_Ar_shares[MPCITH_T][KYBER_K], NTT_As_shares[MPCITH_T][KYBER_K];
    uint16_t sr_shares[MPCITH_N - MPCITH_T][KYBER_K], er_shares[MPCITH_N - MPCITH_T][KYBER_K];
    uint16_t s_eta_shares[MPCITH_N - MPCITH_T][KYBER_K][KYBER_ETA1 * 2 + 1], e_eta_shares[MPCITH_N - MPCITH_T][KYBER_K][KYBER_ETA1 * 2 + 1];
    uint16_t s_sub_eta_shares[MPCITH_T][KYBER_K][KYBER_ETA1 * 2 + 1], e_sub_eta_shares[MPCITH_T][KYBER_K][KYBER_ETA1 * 2 + 1];
    uint16_t z_s_ddeg_shares[MPCITH_T][KYBER_K][KYBER_ETA1 * 2], z_e_ddeg_shares[MPCITH_T][KYBER_K][KYBER_ETA1 * 2];
    uint16_t u_s_2ddeg_shares[MPCITH_N - MPCITH_T][KYBER_K][KYBER_ETA1 * 2], u_e_2ddeg_shares[MPCITH_N - MPCITH_T][KYBER_K][KYBER_ETA1 * 2];
    uint8_t comm[MPCITH_N - MPCITH_T][KYBER_SYMBYTES];
} mpcith_proof;
static_assert(sizeof(mpcith_proof) == MPCITH_PROOF_SIZE, "mpcith_proof must be the wire image (mlwe_prover.cpp:540-543)");
static_assert(sizeof(share_vec) == 8 + 4 * MPCITH_N, "share_vec layout");
static_assert(sizeof(mlwe_inst) == (size_t)(KYBER_K * KYBER_K + 3 * KYBER_K) * 512, "mlwe_inst layout");
#define MPCITH_PRE_RANDOMNESS_SIZE (sizeof(mpcith_randomness) + sizeof(mpcith_range_proof)) /* mlwe_prover.hpp:31 */

extern "C" void randombytes(uint8_t *out, size_t outlen); /* kyber/randombytes.h:7: supplied by the caller's link line */

namespace kosk_compat {
inline void rb_tramp(void *, uint8_t *out, size_t len) { randombytes(out, len); }
inline kosk_ctx *ctx()
{
    static kosk_ctx *c = [] {
        kosk_ctx *h = nullptr;
        if (kosk_create(&h, 0, KYBER_K, 1)) {
            fprintf(stderr, "kosk_create: %s\n", kosk_last_error(nullptr));
            abort(); /* the reference has no error channel either (randombytes.c:49-52 aborts) */
        }
        kosk_set_randombytes(h, rb_tramp, nullptr);
        return h;
    }();
    return c;
}
} // namespace kosk_compat

/* kosk.hpp:18-19 */
inline void kyber_keygen(kyber_keypair *keypair, mlwe_inst *raw_key)
{
    uint8_t seed[64];
    randombytes(seed, 64);
    static int16_t A[KYBER_K * KYBER_K * KYBER_N], s[KYBER_K * KYBER_N], e[KYBER_K * KYBER_N], t[KYBER_K * KYBER_N];
    if (kosk_keygen(KYBER_K, seed, keypair->pk, keypair->sk, A, s, e, t)) abort();
    for (int i = 0; i < KYBER_K; i++) {
        for (int j = 0; j < KYBER_K; j++) memcpy(raw_key->A[i].vec[j].coeffs, A + (i * KYBER_K + j) * KYBER_N, sizeof(int16_t) * KYBER_N);
        memcpy(raw_key->s.vec[i].coeffs, s + i * KYBER_N, sizeof(int16_t) * KYBER_N);
        memcpy(raw_key->e.vec[i].coeffs, e + i * KYBER_N, sizeof(int16_t) * KYBER_N);
        memcpy(raw_key->t.vec[i].coeffs, t + i * KYBER_N, sizeof(int16_t) * KYBER_N);
    }
}
/* kosk.hpp:20-21 */
inline void kyber_verifiable_keygen(kyber_keypair *keypair, uint8_t *pi)
{
    if (kosk_verifiable_keygen_batch(kosk_compat::ctx(), 1, nullptr, 0, keypair->pk, keypair->sk, pi)) {
        fprintf(stderr, "kyber_verifiable_keygen: %s\n", kosk_last_error(kosk_compat::ctx()));
        abort();
    }
}
/* kosk.hpp:23-24 */
inline bool kyber_kosk_verify(const uint8_t *pi, const uint8_t *pk)
{
    uint8_t ok = 0;
    if (kosk_verify_batch(kosk_compat::ctx(), 1, pi, pk, &ok)) {
        fprintf(stderr, "kyber_kosk_verify: %s\n", kosk_last_error(kosk_compat::ctx()));
        abort();
    }
    return ok == 1;
}


/* ---- second-level entry points, used directly by main.cpp:21-47 ---- */
namespace kosk_compat {
inline void must(int rc, const char *what)
{
    if (rc) {
        fprintf(stderr, "%s: %s\n", what, kosk_last_error(ctx()));
        abort();
    }
}
} // namespace kosk_compat
/* mlwe_prover.hpp:77 */
inline void prepare_randomness(mpcith_randomness *rand)
{
    kosk_compat::must(kosk_prepare_randomness(kosk_compat::ctx(), 1, nullptr, 0, reinterpret_cast<uint8_t *>(rand)), "prepare_randomness");
}
/* mlwe_prover.hpp:78 */
inline void prepare_range_proof(mpcith_range_proof *eta_shares)
{
    kosk_compat::must(kosk_prepare_range_proof(kosk_compat::ctx(), 1, nullptr, 0, reinterpret_cast<uint8_t *>(eta_shares)), "prepare_range_proof");
}
/* mlwe_prover.hpp:96-99 */
inline void prove(mpcith_proof *pi, const mlwe_inst *mlwe, const mpcith_randomness *rand, const mpcith_range_proof *eta_share)
{
    kosk_compat::must(kosk_prove_prepared(kosk_compat::ctx(), 1, reinterpret_cast<const uint8_t *>(mlwe),
                                          reinterpret_cast<const uint8_t *>(rand), reinterpret_cast<const uint8_t *>(eta_share),
                                          nullptr, 0, reinterpret_cast<uint8_t *>(pi)), "prove");
}
/* mlwe_verifier.hpp:14-15 */
inline bool verify(const mpcith_proof *pi, const mlwe_inst *mlwe)
{
    uint8_t ok = 0;
    kosk_compat::must(kosk_verify_inst(kosk_compat::ctx(), 1, reinterpret_cast<const uint8_t *>(pi),
                                       reinterpret_cast<const uint8_t *>(mlwe), &ok), "verify");
    return ok == 1;
}
/* mlwe_prover.hpp:72-75; the struct IS the wire image (mlwe_prover.cpp:540-543) */
inline void encode_mpcith_proof(uint8_t *buf, const mpcith_proof *pi) { memcpy(buf, pi, sizeof(mpcith_proof)); }
inline void decode_mpcith_proof(mpcith_proof *pi, const uint8_t *buf) { memcpy(pi, buf, sizeof(mpcith_proof)); }
/* mlwe_prover.hpp:50-55.  The reference's decode forgets eta_shares (mlwe_prover.cpp:70-79); this one restores both. */
inline void encode_preprocessed_randomness(uint8_t *buf, const mpcith_randomness *rand, const mpcith_range_proof *eta_shares)
{
    memcpy(buf, rand, sizeof(mpcith_randomness));
    memcpy(buf + sizeof(mpcith_randomness), eta_shares, sizeof(mpcith_range_proof));
}
inline void decode_preprocessed_randomness(mpcith_randomness *rand, mpcith_range_proof *eta_shares, const uint8_t *buf)
{
    memcpy(rand, buf, sizeof(mpcith_randomness));
    memcpy(eta_shares, buf + sizeof(mpcith_randomness), sizeof(mpcith_range_proof));
}

#endif // KOSK_COMPAT_HPP
