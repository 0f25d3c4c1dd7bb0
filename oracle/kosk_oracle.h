/*
 * kosk_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Single-thread plain-C restatement of the reference's KOSK hot path
 * (ZGC-SP/mpcith_kyber_kosk: kosk.cpp, mlwe_prover.cpp, mlwe_verifier.cpp,
 * ss.cpp, utils/gf3329.c, kyber/{fips202,ntt,reduce,poly,polyvec,cbd,indcpa,
 * symmetric-shake}.c).  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library; the product path
 * (mpcith_kyber_kosk_amd/csrc) never links or calls it.
 *
 * Parity pin: sha3_256(pk / sk / proof) for KYBER_K = 2,3,4 on the tape
 * SHAKE256("kosk-tape-v1:0"), recorded from the compiled reference in
 * SURVEY.md section 8(c) / BASELINE.md section 2, checked by
 * tests/test_oracle_golden.py; primitives are additionally checked against
 * oracle/_ref (the reference's own kyber/ and utils/gf3329.c sources compiled
 * in place) by tests/test_oracle_vs_ref.py when that build exists.
 *
 * Unlike the reference (KYBER_K is a compile-time macro, params.hpp:8-10) the
 * oracle takes kyber_k in {2,3,4} at run time so one library serves all sets.
 */
#ifndef KOSK_ORACLE_H
#define KOSK_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KO_Q 3329
#define KO_NSEC 256      /* KYBER_N = MPCITH_L: packed secrets per sharing  */
#define KO_PARTIES 1454  /* MPCITH_N                  params.hpp:13,21,28   */
#define KO_OPENED 150    /* MPCITH_T                  params.hpp:14         */
#define KO_REST (KO_PARTIES - KO_OPENED)
#define KO_NCHK 70       /* MPCITH_K                  params.hpp:16         */
#define KO_DEG 406       /* DEG_D                     ss.hpp:56             */
#define KO_DEG2 812      /* DEG_2D                    ss.hpp:57             */
#define KO_NFIELDS 24

typedef struct {
    int K;       /* KYBER_K                                            */
    int eta1;    /* KYBER_ETA1                 kyber/params.h:29-41    */
    int M;       /* MPCITH_K + MPCITH_V + 1 = 71 + 2K                  */
    int V;       /* MPCITH_V = 2K                                      */
    int E;       /* 2*eta1 + 1 range constants                         */
    int Z;       /* 2*eta1 multiplication gates per polynomial         */
    size_t pk_bytes, sk_bytes, proof_bytes, tape_bytes;
    int tape_calls;
    size_t tcomm_msg_bytes, view_msg_bytes;
    /* wire image of mpcith_proof (mlwe_prover.hpp:57-75), field by field */
    size_t off[KO_NFIELDS];
    size_t size[KO_NFIELDS];
} ko_params;

enum { /* field ids, declaration order of mpcith_proof */
    KO_F_F = 0, KO_F_NTTF, KO_F_BETA, KO_F_GAMMA, KO_F_TCOMM, KO_F_I,
    KO_F_S, KO_F_E, KO_F_T, KO_F_NTTS, KO_F_NTTE, KO_F_NTTAR, KO_F_NTTAS,
    KO_F_SR, KO_F_ER, KO_F_SETA, KO_F_EETA, KO_F_SSUB, KO_F_ESUB,
    KO_F_ZS, KO_F_ZE, KO_F_US, KO_F_UE, KO_F_COMM
};

int ko_get_params(int kyber_k, ko_params *p);

/* ---- randomness tape: stands in for randombytes() (kyber/randombytes.h:7) */
typedef struct {
    const uint8_t *buf;
    size_t len, pos;
    size_t calls;
    int overrun;
} ko_tape;
void ko_tape_init(ko_tape *t, const uint8_t *buf, size_t len);

/* ---- L0 primitives ----------------------------------------------------- */
void ko_keccak_f1600(uint64_t st[25]);
void ko_sha3_256(uint8_t out[32], const uint8_t *in, size_t inlen);
void ko_sha3_512(uint8_t out[64], const uint8_t *in, size_t inlen);
void ko_shake128(uint8_t *out, size_t outlen, const uint8_t *in, size_t inlen);
void ko_shake256(uint8_t *out, size_t outlen, const uint8_t *in, size_t inlen);
void ko_shake256_prf(uint8_t *out, size_t outlen, const uint8_t key[32], uint8_t nonce);

uint16_t ko_gf_add(uint16_t a, uint16_t b);
uint16_t ko_gf_sub(uint16_t a, uint16_t b);
uint16_t ko_gf_mul(uint16_t a, uint16_t b);
uint16_t ko_gf_inv(uint16_t a);
uint16_t ko_gf_encode(int16_t a);
int16_t ko_gf_decode(uint16_t a);

/* ---- L1 Kyber ring arithmetic ------------------------------------------ */
int16_t ko_montgomery_reduce(int32_t a);
int16_t ko_barrett_reduce(int16_t a);
const int16_t *ko_zetas(void);
void ko_ntt(int16_t r[256]);       /* ntt.c:80-95, no final reduction      */
void ko_poly_ntt(int16_t r[256]);  /* poly.c:261-265, Barrett-centred      */
void ko_poly_tomont(int16_t r[256]);
void ko_poly_reduce(int16_t r[256]);
void ko_polyvec_basemul_acc(int16_t r[256], const int16_t *a, const int16_t *b, int K);
void ko_poly_tobytes(uint8_t r[384], const int16_t a[256]);
void ko_poly_frombytes(int16_t r[256], const uint8_t a[384]);
void ko_gen_matrix(int16_t *A /* [K][K][256] */, const uint8_t seed[32], int transposed, int K);
void ko_cbd(int16_t r[256], const uint8_t *buf, int eta);

/* ---- L2 packed Shamir sharing ------------------------------------------ */
const uint16_t *ko_table_share_ddeg(void);   /* [1303][407] */
const uint16_t *ko_table_recon_ddeg(void);   /* [256][407]  */
const uint16_t *ko_table_recon_2ddeg(void);  /* [256][813]  */
void ko_share_secrets_ddeg(uint16_t share_y[KO_PARTIES], const uint16_t secret[256], ko_tape *t);
void ko_recompute_share_secrets_ddeg(uint16_t share_y[KO_PARTIES], const uint16_t y[KO_DEG + 1]);
void ko_recon_secrets_ddeg(uint16_t secret[256], const uint16_t share_y[KO_PARTIES]);
void ko_recon_secrets_2ddeg(uint16_t secret[256], const uint16_t share_y[KO_PARTIES]);
/* values at 0..neval-1 of the unique degree<n polynomial through (x_i,y_i);
 * replaces NTL interpolate+eval (mlwe_verifier.cpp:201-219)                */
void ko_interp_eval(uint16_t *out, int neval, const uint16_t *x, const uint16_t *y, int n);

/* ---- L3/L4 protocol ----------------------------------------------------- */
typedef struct {
    int16_t A[4][4][256]; /* A[i].vec[j]                                    */
    int16_t t[4][256];
    int16_t s[4][256];
    int16_t e[4][256];
} ko_mlwe;

typedef struct ko_pre ko_pre; /* mpcith_randomness + mpcith_range_proof     */
ko_pre *ko_pre_alloc(void);
void ko_pre_free(ko_pre *);
const uint16_t *ko_pre_f(const ko_pre *, int i);           /* [256]   */
const uint16_t *ko_pre_ntt_f(const ko_pre *, int i);       /* [256]   */
const uint16_t *ko_pre_f_shares(const ko_pre *, int i);    /* [1454]  */
const uint16_t *ko_pre_ntt_f_shares(const ko_pre *, int i);
const uint16_t *ko_pre_s_eta_shares(const ko_pre *, int i, int j);
const uint16_t *ko_pre_e_eta_shares(const ko_pre *, int i, int j);

/* optional stage outputs of prove(), for stage-level parity tests */
typedef struct {
    uint8_t tcomm[KO_PARTIES][32];
    uint8_t h1[32];
    uint16_t alpha[KO_NCHK + 8];
    uint8_t view_digest[KO_PARTIES][32];
    uint8_t ch[32];
    uint16_t sr_rec[4][256], er_rec[4][256];
} ko_trace;

void ko_keygen(int K, ko_tape *t, uint8_t *pk, uint8_t *sk, ko_mlwe *raw);
/* kosk.cpp:16-69 from the two halves of hash_g's output */
void ko_keygen_from_seeds(int K, const uint8_t public_seed[32], const uint8_t noise_seed[32], uint8_t *pk, uint8_t *sk, ko_mlwe *raw);
void ko_prepare_randomness(int K, ko_tape *t, ko_pre *pre);
void ko_prepare_range_proof(int K, ko_tape *t, ko_pre *pre);
void ko_prove(int K, ko_tape *t, uint8_t *pi, const ko_mlwe *mlwe, const ko_pre *pre, ko_trace *trace);
int ko_verify(int K, const uint8_t *pi, const ko_mlwe *mlwe, char *why, size_t whylen);
void ko_verifiable_keygen(int K, ko_tape *t, uint8_t *pk, uint8_t *sk, uint8_t *pi, ko_trace *trace);
int ko_kosk_verify(int K, const uint8_t *pi, const uint8_t *pk, char *why, size_t whylen);

/* timing harness for bench.py cpu_baseline: nproofs x (verifiable_keygen +
 * kosk_verify), clock()-timed like main.cpp:18-94; returns #verified       */
int ko_bench(int K, int nproofs, const uint8_t *tapes, size_t tape_stride,
             double *sec_keygen_prove, double *sec_verify);

/* test hook (no reference counterpart): a crafting prover that plants u16 values >= q into its own shares before it
 * commits to them, so that the verifier's non-reducing arithmetic meets hash-consistent inputs (see kosk_oracle.c) */
void ko_craft_clear(void);
int ko_craft_add(int kind, int idx, int party, int mult);

#ifdef __cplusplus
}
#endif
#endif
