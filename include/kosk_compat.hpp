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

#endif // KOSK_COMPAT_HPP
