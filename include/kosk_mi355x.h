/*
 * kosk_mi355x.h -- C ABI of the MI355X-native KOSK prover/verifier.
 *
 * Drop-in boundary for the reference's hot path (ZGC-SP/mpcith_kyber_kosk).
 * Every entry point names the reference interface it replaces (file:line
 * relative to the reference root).  Plain pointers and sizes only; the library
 * owns all device memory.  All functions return 0 on success and a negative
 * value on error (kosk_last_error() gives the text) unless stated otherwise.
 * There is no CPU fallback: kosk_create() fails when no HIP device is present.
 *
 * Parameter sets: kyber_k in {2,3,4} is a run-time argument here; in the
 * reference it is the compile-time macro KYBER_K (params.hpp:8-10).
 */
#ifndef KOSK_MI355X_H
#define KOSK_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct kosk_ctx kosk_ctx;

/* Replaces the link-time symbol randombytes() (kyber/randombytes.h:7).  The
 * library calls it in exactly the reference's order and lengths for each
 * proof: 64, M x 32, then 302-byte draws. */
typedef void (*kosk_randombytes_fn)(void *user, uint8_t *out, size_t len);

/* sizes: KYBER_PUBLICKEYBYTES / KYBER_SECRETKEYBYTES (kyber/params.h:49-52),
 * MPCITH_PROOF_SIZE (mlwe_prover.hpp:30), randomness per verifiable keygen */
size_t kosk_pk_bytes(int kyber_k);
size_t kosk_sk_bytes(int kyber_k);
size_t kosk_proof_bytes(int kyber_k);
size_t kosk_tape_bytes(int kyber_k);
/* byte offset / size of field `idx` (0..23, declaration order of mpcith_proof,
 * mlwe_prover.hpp:57-75) inside the proof image; returns 0 on success */
int kosk_proof_field(int kyber_k, int idx, size_t *offset, size_t *size);

int kosk_create(kosk_ctx **ctx, int device, int kyber_k, int max_batch);

/* ---- Per-handle options (round 6).  The reference has no run-time configuration at all (kosk.hpp:18-24 takes pointers, nothing
 * else); what a HOST must decide per handle -- how its calls are batched, what the verifier accepts, where the Fiat-Shamir rounds
 * run, how its threads wait -- is this struct, not the process environment.  kosk_options_init() fills in "not given" for every
 * field (the library's defaults apply); kosk_create_ex() with opt == NULL is kosk_create().  Fields the caller sets win over the
 * environment variables of the same name, which kosk_create() still honours. */
enum { KOSK_FS_HOST = 0, KOSK_FS_DEVICE = 1 };
typedef struct kosk_options {
    uint32_t size;              /* sizeof(kosk_options) as the caller compiled it (set by kosk_options_init) */
    int32_t streams;            /* sub-batches of a batch call in flight on separate HIP streams, 1..8; 0: not given = 1 */
    int32_t combine;            /* handles per cohort whose resident calls are merged, 2..16 (call combining, see below); 0: not given, 1: off */
    int32_t combine_wait_us;    /* longest wait of a call for the cohort's other members (default 5000); < 0: not given */
    int32_t combine_idle_us;    /* a member that left a call longer ago than this is not waited for (default 1000); < 0: not given */
    int32_t combine_prewake_us; /* how long the sleeping callers of a merged run may spin for its return (default 400); < 0: not given */
    int32_t strict_encoding;    /* verifier: 1 (DEFAULT) a u16 element >= q in any record the reference reads marks the proof malformed;
                                 * 0 the reference-following mode (accepts what the reference accepts: non-canonical encodings are then
                                 * malleable); < 0: not given */
    int32_t fs_mode;            /* KOSK_FS_HOST: the Fiat-Shamir hashes run on the host's cores (digest tables cross PCIe, four host
                                 * round trips per prove + verify); KOSK_FS_DEVICE: one wave per proof hashes the tables in HBM, alpha /
                                 * I / the verifier's I' == I stay on the device, the resident calls have no host round trip; < 0: not given */
    int32_t host_threads;       /* Fiat-Shamir / key-assembly workers of this handle; 0: not given */
    int32_t blocking_sync;      /* 1: the handle's host waits sleep on events instead of spinning (few cores per GPU); < 0: not given */
    int32_t hooks_unmerged;     /* 1: while this handle has a round hook (kosk_set_round_hook) its calls never join a merged run: the hook
                                 * then always fires on the handle's OWN calling thread (thread-local state, blocking hooks); 0 / < 0: merged */
    int32_t reserved[6];        /* zero */
} kosk_options;
void kosk_options_init(kosk_options *opt);
int kosk_create_ex(kosk_ctx **ctx, int device, int kyber_k, int max_batch, const kosk_options *opt);
void kosk_destroy(kosk_ctx *ctx);
const char *kosk_last_error(const kosk_ctx *ctx); /* ctx may be NULL: error of the last failed kosk_create */
int kosk_set_randombytes(kosk_ctx *ctx, kosk_randombytes_fn fn, void *user); /* NULL: OS entropy */

/* void kyber_verifiable_keygen(kyber_keypair *keypair, uint8_t *pi)   kosk.hpp:20-21, kosk.cpp:72-86
 * n independent instances; pk/sk/pi are n consecutive records of kosk_*_bytes().
 * `tapes` == NULL draws randomness through the randombytes callback; otherwise
 * proof b consumes tapes[b*tape_stride ..] (kosk_tape_bytes() bytes each).
 * Error containment (all entry points): no C++ exception and no abort leaves the library -- a failed allocation, thread or
 * HIP call is rc -1 + kosk_last_error(); the only abort is the reference's own, an OS entropy failure
 * (kyber/randombytes.c:49-52).  A batch call creates no threads (kosk_create made them).
 * HIP error state: the library's launchers read hipGetLastError(), so every entry point first RESETS the calling thread's HIP
 * last-error state; an application that launches kernels of its own must check them before it calls in here.
 * The library never page-locks caller memory (KOSK_REGISTER=2 of rounds 2-4, which did so for the duration of a multi-chunk call,
 * was removed in round 5: the two process aborts on record both happened inside calls that had just page-locked Python heap
 * memory and were never explained; the value is now read as 1).
 * gen_matrix's rejection sampling (indcpa.c:124-145) loops without a bound in the reference; on the host this library does the
 * same, on the GPU it squeezes at most 32 SHAKE128 blocks per matrix entry (three suffice with probability 1 - 2^-40) and a
 * call that ever reached that limit returns -1 ("block limit") without results -- for key generation and for the verifier's
 * decoding of a public key alike. */
int kosk_verifiable_keygen_batch(kosk_ctx *ctx, int n, const uint8_t *tapes, size_t tape_stride,
                                 uint8_t *pk, uint8_t *sk, uint8_t *pi);

/* Page-locked host memory for the proof buffers of the two host-buffer calls (no reference counterpart: the reference's
 * caller owns plain arrays, main.cpp:71).  Proof images in such a buffer (or in any memory the caller page-locked itself with
 * hipHostMalloc / hipHostRegister) cross PCIe straight from / to it: no staging copy on the host and no per-call locking.
 * Plain (pageable) buffers keep working through the library's pinned staging buffers.  NULL on failure. */
void *kosk_host_alloc(size_t bytes);
void kosk_host_free(void *p);

/* bool kyber_kosk_verify(const uint8_t *pi, const uint8_t *pk)       kosk.hpp:23-24, kosk.cpp:88-117
 * ok[b] = 1 accept / 0 reject.  The reference prints a diagnostic and returns
 * false (mlwe_verifier.cpp:120 etc.); here kosk_verify_fail_masks() reports
 * which checks failed (bit i = check i of DESIGN.md's verifier table). */
int kosk_verify_batch(kosk_ctx *ctx, int n, const uint8_t *pi, const uint8_t *pk, uint8_t *ok);
int kosk_verify_fail_masks(const kosk_ctx *ctx, uint32_t *masks, int n); /* n <= proofs of the last completed verify call */

/* ---- Second-level entry points of the reference (mlwe_prover.hpp:77-99, mlwe_verifier.hpp:14-15; used directly by
 * main.cpp:21-47) on n instances at a time.  The buffers are arrays of the reference's structs for this kyber_k, byte for
 * byte (little-endian, no padding):
 *   share_vec          { uint64 len = 1454; u16 share_x[1454] (= 256 + party); u16 share_y[1454]; }      ss.hpp:33-37
 *   mpcith_randomness  { u16 f[M][256]; u16 NTT_f[M][256]; share_vec f_shares[M]; share_vec NTT_f_shares[M]; }   M = 71 + 2K
 *   mpcith_range_proof { share_vec s_eta_shares[K][2 eta1 + 1]; share_vec e_eta_shares[K][2 eta1 + 1]; }
 *   mlwe_inst          { i16 A[K][K][256] (NTT domain); i16 t[K][256]; i16 s[K][256]; i16 e[K][256]; }   = kosk_keygen's outputs
 * Randomness: `tapes` = per instance exactly the bytes the reference would draw in that call, in its order
 * (prepare_randomness: M x 32 then 2M x 302; prepare_range_proof: 2K(2 eta1 + 1) x 302; prove: (3K + 4K eta1) x 302),
 * tape_stride apart; NULL = the randombytes callback / OS entropy with the reference's call sequence. */
size_t kosk_randomness_bytes(int kyber_k);  /* sizeof(mpcith_randomness)  */
size_t kosk_range_proof_bytes(int kyber_k); /* sizeof(mpcith_range_proof) */
size_t kosk_mlwe_inst_bytes(int kyber_k);   /* sizeof(mlwe_inst)          */
/* prepare_randomness, mlwe_prover.cpp:4-39 */
int kosk_prepare_randomness(kosk_ctx *ctx, int n, const uint8_t *tapes, size_t tape_stride, uint8_t *rand_out);
/* prepare_range_proof, mlwe_prover.cpp:41-59 */
int kosk_prepare_range_proof(kosk_ctx *ctx, int n, const uint8_t *tapes, size_t tape_stride, uint8_t *range_out);
/* prove + encode_mpcith_proof, mlwe_prover.cpp:81-543: pi receives n proof images */
int kosk_prove_prepared(kosk_ctx *ctx, int n, const uint8_t *inst, const uint8_t *rand_in, const uint8_t *range_in,
                        const uint8_t *tapes, size_t tape_stride, uint8_t *pi);
/* decode_mpcith_proof + verify, mlwe_verifier.cpp:4-686, with A and t from the instance (s, e are not read) */
int kosk_verify_inst(kosk_ctx *ctx, int n, const uint8_t *pi, const uint8_t *inst, uint8_t *ok);

/* Device-resident split of the two calls above, for callers that keep inputs
 * and proofs in HBM (and for bench.py: the timed region is *_resident only). */
int kosk_stage_prover_inputs(kosk_ctx *ctx, int n, const uint8_t *tapes, size_t tape_stride, uint8_t *pk, uint8_t *sk);
int kosk_prove_resident(kosk_ctx *ctx, int n);
int kosk_fetch_proofs(kosk_ctx *ctx, int n, uint8_t *pi);
int kosk_stage_verifier_inputs(kosk_ctx *ctx, int n, const uint8_t *pi, const uint8_t *pk);
int kosk_verify_resident(kosk_ctx *ctx, int n, uint8_t *ok);
/* The two reference calls as ONE resident call each (what bench.py times):
 * kyber_verifiable_keygen (kosk.cpp:72-86) -- key generation (kosk.cpp:4-70) runs on the device at the head of the
 * prover's first segment, pk / sk are written to the HOST buffers, the proofs stay in HBM.  `tapes` may be host or DEVICE
 * memory (a device buffer whose base and tape_stride are multiples of 8 is read in place) or NULL (randombytes callback).
 * kyber_kosk_verify (kosk.cpp:88-117) on the resident proofs -- polyvec_frombytes(t) + gen_matrix (kosk.cpp:94-99) run at
 * the head of the verifier's first segment from `pk` (host or device memory), or, with pk == NULL, from the pk bytes the
 * key generation (or a verifier staging call) of at least n proofs left in HBM on this handle -- an error otherwise.
 * A DEVICE tape buffer that is read in place must stay valid and unmodified until the prove call that consumes it has
 * returned (kosk_verifiable_keygen_resident itself, or kosk_prove_resident after kosk_stage_prover_inputs). */
int kosk_verifiable_keygen_resident(kosk_ctx *ctx, int n, const uint8_t *tapes, size_t tape_stride, uint8_t *pk, uint8_t *sk);
int kosk_verify_resident_pk(kosk_ctx *ctx, int n, const uint8_t *pk, uint8_t *ok);
/* ---- Compact wire format (no reference counterpart: the reference ships the raw image of
 * mpcith_proof, mlwe_prover.cpp:540-543).  Same 24 fields in the same order, every u16 field packed two values into
 * three bytes (12 bits each, the bit order of Kyber's poly_tobytes, kyber/poly.c:128-147), the two digest fields raw,
 * every field on a 16-byte boundary: 664 340 / 680 980 / 744 148 -> 78 % of that.  Lossless for images whose u16
 * values are below 4096 (compression fails otherwise).  The resident variants pack / unpack on the GPU so that PCIe
 * carries the compact bytes. */
size_t kosk_compact_proof_bytes(int kyber_k);
int kosk_proof_compress(int kyber_k, const uint8_t *pi, uint8_t *out);   /* host codec; -1 if a value >= 4096 */
int kosk_proof_decompress(int kyber_k, const uint8_t *in, uint8_t *pi);
/* kosk_verifiable_keygen_batch / kosk_verify_batch with the proofs in the compact format: n records of
 * kosk_compact_proof_bytes(); any n (chunked like the image calls), host or page-locked host buffers */
int kosk_verifiable_keygen_batch_compact(kosk_ctx *ctx, int n, const uint8_t *tapes, size_t tape_stride,
                                         uint8_t *pk, uint8_t *sk, uint8_t *out);
int kosk_verify_batch_compact(kosk_ctx *ctx, int n, const uint8_t *in, const uint8_t *pk, uint8_t *ok);
int kosk_fetch_proofs_compact(kosk_ctx *ctx, int n, uint8_t *out);        /* like kosk_fetch_proofs */
int kosk_stage_verifier_inputs_compact(kosk_ctx *ctx, int n, const uint8_t *in, const uint8_t *pk); /* like kosk_stage_verifier_inputs */

/* Which kernel / copy paths ran on this handle since it was created (the tests of the fallbacks and of the Fiat-Shamir mode assert on
 * these).  ids: 0 commitment hash with LDS-DMA staging, 1 without (a layout the staged kernel cannot take: unaligned rows, a lane map),
 * 2 shared-table products on k_table_gemm / k_table_gemm_p, 3 products on the generic limb GEMM (grouped products, unaligned callers of the
 * kernel-level entry points), 4 proof images copied straight between HBM and page-locked caller memory (kosk_host_alloc or locked by the
 * caller), 5 through the pinned staging buffer (pageable caller memory), 6 hipGraph segment replays (KOSK_GRAPHS=1), 7 commitment rounds
 * whose digest table was copied to the host (host Fiat-Shamir mode), 8 small copies between HBM and the library's own page-locked buffers
 * made by a copy kernel, 9 Fiat-Shamir rounds hashed on the device (k_fs_chain), 10 on the host. */
int kosk_path_count(const kosk_ctx *ctx, int id, long *count);
/* host worker threads per sub-context (kosk_options::host_threads; else <= 8, <= CPUs of the process / streams; all created by kosk_create) */
int kosk_host_threads(const kosk_ctx *ctx);

/* wall seconds of the phases of the last prove / verify on this context (16 values: host_pre, gpu_commit,
 * fs_alpha, gpu_relation, fs_open, gpu_assemble, d2h, then the host time spent issuing the prover's three
 * segments, and the verifier's issue1, wait1, fs_alpha, issue2, wait2, fs_open) */
int kosk_phase_seconds(const kosk_ctx *ctx, double *out, int n);

/* HIP-event timing of the library's own launches on its stream.  ids: 0 prover Tcomm hash, 1 prover view
 * hash, 2 expansion GEMM #1, 3 expansion GEMM #2, 4 beta/gamma lincomb, 5 NTT(f), 6 wire image,
 * 7/8 verifier Tcomm/view hash, 9 interpolation operators, 10 interpolation GEMM, 11 verifier expansion,
 * 12 reconstruction GEMM, 13 verifier lincomb.  Reading returns the sum over launches since enable.
 * on = 1: only the view-commitment hashes (ids 1 and 8), which are plain launches; everything else keeps
 * running as captured hipGraph segments (the production path).  on = 2: every id, with plain stream launches
 * instead of graphs (diagnostic: changes the launch overhead being measured).  on = 0: off. */
int kosk_profile_enable(kosk_ctx *ctx, int on);
int kosk_profile_read(const kosk_ctx *ctx, int id, double *total_ms, long *launches);
/* the same plus the proofs those launches served: a merged run of a cohort (call combining, below) serves several callers'
 * batches per launch, and its launches are timed on the handle that led the run */
int kosk_profile_read_units(const kosk_ctx *ctx, int id, double *total_ms, long *launches, long *proofs);

/* ---- Call combining (no reference counterpart: the reference is one call, one proof, one thread -- kosk.hpp:18-24).
 * With kosk_options::combine = C (2..16, needs streams = 1), handles created with equal (device, kyber_k,
 * max_batch) are grouped into cohorts of C that share one workspace, and the resident calls kosk_verifiable_keygen_resident /
 * kosk_verify_resident_pk that neighbouring members of a cohort make at about the same time -- each from its own thread --
 * are served by ONE pipeline run over all their proofs: every launch then covers 2..C callers' batches, which is what the
 * chip-filling kernels need (a 46-proof launch leaves the commitment hash at one and a bit rounds of waves per SIMD).  Per
 * call nothing changes: same arguments, same results byte for byte, each caller gets its own pk / sk / verify bits / fail
 * masks / resident proofs (kosk_resident_proofs, kosk_resident_digests point at the member's own block).  A member's call
 * waits at most combine_wait_us (default 5000) for the other members, and only for those that are inside a call or left
 * one less than combine_idle_us (default 1000) ago: a lone caller is never delayed, callers that loop fall into step after
 * one or two calls (a request whose kind is in the minority of its window is held back once, so that callers alternating
 * keygen / verify in opposite phase meet).  The callers of a merged run sleep while it executes and are woken shortly before its
 * end (they then spin at most combine_prewake_us, default 400, for the return).  Calls that draw randomness through
 * the callback (tapes == NULL) and every other entry point run unmerged on the member's own block.  A member's round hook
 * (kosk_set_round_hook) fires from a merged run as well, with that member's block of the table, ON THE THREAD OF THE RUN'S LEADER while
 * the member's own caller sleeps inside its call: a hook that relies on thread-local state (a current device, a stream context, a
 * thread-affine communicator) or that blocks must opt out with kosk_options::hooks_unmerged = 1, which keeps the calls of a handle
 * with a hook out of merged runs.  A cohort is formed by handles whose options agree (combine, strict_encoding, fs_mode,
 * host_threads, blocking_sync).  All handles of a cohort must be destroyed before the process ends (the last one frees the workspace).
 * kosk_combine_stats: resident calls of THIS handle that went through the combiner, and the sum over those calls of the
 * members their run served (members / calls = mean callers per launch; both 0 for a handle outside a cohort). */
int kosk_combine_stats(const kosk_ctx *ctx, long *calls, long *members);

/* hipEvent pair on the ctx stream around caller-issued kernel-level calls (micro-benchmarks) */
int kosk_stream_timer_start(kosk_ctx *ctx);
int kosk_stream_timer_stop(kosk_ctx *ctx, double *ms);

/* ---- kernel-level entry points on DEVICE pointers (stream 0 of the ctx) ----
 * Used by the parity tests and by bench.py's roofline leg.  Every stream of the library is a NON-BLOCKING HIP stream: it is not
 * ordered against the legacy null stream (nor against any other stream of the caller).  Device buffers handed to these entry
 * points -- and device tapes / keys handed to the resident calls -- must be complete before the call (synchronise the stream
 * that produced them), and kosk_device_synchronize() must have returned before another stream reads the outputs. */

/* sha3_256(h, in, inlen) for n equal-length messages        kyber/fips202.c:745-754
 * message-major layout: message i at in + i*in_stride; digest i at out + 32*i */
int kosk_sha3_256_batch(kosk_ctx *ctx, const uint8_t *d_in, size_t in_stride, size_t inlen, uint8_t *d_out, int n);
/* the same function computed by the lane-pair ("warp-cooperative") sponge: one Keccak state spread over two adjacent lanes of a
 * wave, 64-bit rotations exchanged by DPP, 32 messages per wave (csrc/kosk_keccak_split_dev.hpp; BASELINE.json north_star names
 * this layout; the pipeline's hashes use one lane per state, which is faster at its wave counts -- DESIGN.md 8) */
int kosk_sha3_256_batch_pair(kosk_ctx *ctx, const uint8_t *d_in, size_t in_stride, size_t inlen, uint8_t *d_out, int n);
/* sha3_256 of n LONG messages, one WAVE per message: one Keccak state spread over the 64 lanes
 * of a wave, a 32-bit word of the bit-interleaved state per lane, theta / pi / chi exchanged through LDS (csrc/kosk_fs_dev.hpp) --
 * the layout for a strictly sequential chain: ~2 us per permutation where the one-state-per-lane sponge needs ~9 us when its wave
 * runs alone.  d_in and in_stride must be multiples of 8.        kyber/fips202.c:745-754 */
int kosk_sha3_256_batch_wave(kosk_ctx *ctx, const uint8_t *d_in, size_t in_stride, size_t inlen, uint8_t *d_out, int n);
/* kosk_fs_alpha / kosk_fs_opened (below) on n digest tables [1454][32] in HBM, table_stride bytes apart (multiples of 8): what a
 * handle in device Fiat-Shamir mode runs between its commitment kernels and their consumers (mlwe_prover.cpp:130-153, :445-474).
 * d_alpha: n x 80 u16 (70 + 2K derived, zeros behind); d_h1 / d_ch: optional n x 32 bytes, the tables' sha3_256.
 * d_sel: n rows of sel_stride u16 -- I in [0, 150), at 160 + w the number of unopened parties below 64 w (w = 0..23), at 192 the
 * opened parties ascending and at 352 their positions in I; d_rest: n rows of sel_stride u16, the ascending complement (1304). */
int kosk_fs_alpha_device(kosk_ctx *ctx, const uint8_t *d_tables, size_t table_stride, int n, uint16_t *d_alpha, uint8_t *d_h1);
int kosk_fs_opened_device(kosk_ctx *ctx, const uint8_t *d_tables, size_t table_stride, int n, uint16_t *d_sel, uint16_t *d_rest, int sel_stride, uint8_t *d_ch);
/* shake256(out, outlen, in, inlen)                           kyber/fips202.c:723-734 */
int kosk_shake256_batch(kosk_ctx *ctx, const uint8_t *d_in, size_t in_stride, size_t inlen,
                        uint8_t *d_out, size_t outlen, int n);
/* The view-commitment kernel on its native column layout: lane l hashes
 * [prefix32(l)] || rows[r][l], r < words, rows being u16 arrays `row_stride`
 * elements apart (mlwe_prover.cpp:116-127 when with_prefix == 0, :397-444 when 1).
 * words must equal the Tcomm / view word count of kyber_k. */
int kosk_commit_hash_lanes(kosk_ctx *ctx, const uint16_t *d_rows, size_t row_stride, int n_lanes,
                           const uint8_t *d_prefix, int with_prefix, uint8_t *d_out);
/* poly_ntt(r) on n polynomials of 256 int16                  kyber/poly.c:261-265 (ntt.c:80-95 + Barrett) */
int kosk_ntt256_batch(kosk_ctx *ctx, const int16_t *d_in, int16_t *d_out, int n);
/* share values of all 1454 parties from the 407 values at points 0..406
 * (recompute_share_secrets_ddeg, ss.cpp:76-99): in  n x 407 u16, out n x 1454 u16 */
int kosk_lagrange_expand(kosk_ctx *ctx, const uint16_t *d_y407, uint16_t *d_shares, int n);
/* recon_secrets_ddeg / recon_secrets_2ddeg (ss.cpp:37-73): in n x 1454 u16, out n x 256 u16 */
int kosk_recon_secrets(kosk_ctx *ctx, const uint16_t *d_shares, uint16_t *d_secrets, int n, int two_d);
int kosk_device_synchronize(kosk_ctx *ctx);
/* number of sub-batches a handle keeps in flight on separate HIP streams (kosk_options::streams, default 1) */
int kosk_streams(const kosk_ctx *ctx);
/* device pointer / stride of the resident proof images of sub-batch 0, for callers chaining work in HBM
 * (with streams = 1 this is the whole batch) */
int kosk_resident_proofs(kosk_ctx *ctx, void **d_proofs, size_t *stride);
/* The per-party commitment digests in HBM: round 0 = Tcomm[1454][32] of every proof of the last batch
 * (mlwe_prover.cpp:116-127, the input of sha3_256(Tcomm[0..N)) at :130-135), round 1 = the view commitments
 * (:397-444, input of :445-449); `stride` = bytes per proof (1454 * 32).  This is what a multi-GPU job all-gathers
 * (RCCL) after each commitment round (BASELINE.json configs[3]).  Needs streams = 1. */
int kosk_resident_digests(kosk_ctx *ctx, int round, void **d_digests, size_t *stride);
/* Called on the calling thread -- for a call served by a merged run of a cohort (call combining): on the thread of the caller that
 * leads the run, while this handle's own caller sleeps inside its call -- as soon as a round's table is complete in HBM (role 0
 * prover / 1 verifier; round as above; bytes = n * 1454 * 32; d_digests = this handle's own block); the context's stream may
 * already be running the kernels of the next segment, none of which writes the tables.  The place to start that all-gather so that it overlaps the host's Fiat-Shamir hashing.
 * fn == NULL removes the hook. */
typedef void (*kosk_round_fn)(void *user, int role, int round, const void *d_digests, size_t bytes);
int kosk_set_round_hook(kosk_ctx *ctx, kosk_round_fn fn, void *user);

/* ---- host-only pieces of the path (no device needed) ------------------------ */

/* void kyber_keygen(kyber_keypair *keypair, mlwe_inst *raw_key)      kosk.hpp:18-19, kosk.cpp:4-70
 * seed64 = the 64 bytes the reference draws with randombytes (kosk.cpp:12).
 * Raw key (mlwe_inst, mlwe_prover.hpp:34-37): A [K][K][256] NTT domain in [0,q),
 * s,e [K][256] small signed, t [K][256] NTT domain centred.  Any of A/s/e/t may be NULL. */
int kosk_keygen(int kyber_k, const uint8_t seed64[64], uint8_t *pk, uint8_t *sk,
                int16_t *A, int16_t *s, int16_t *e, int16_t *t);
/* alpha challenge from the 1454 Tcomm digests (mlwe_prover.cpp:130-142); alpha has 70+2K entries */
int kosk_fs_alpha(int kyber_k, const uint8_t *tcomm_all, uint16_t *alpha);
/* opened set I[150] and its ascending complement rest[1304] from the 1454 view digests
 * (mlwe_prover.cpp:445-490) */
int kosk_fs_opened(const uint8_t *digests_all, uint16_t *I, uint16_t *rest);
/* host sha3_256 / shake256 used by the two functions above (kyber/fips202.c:745-754, :723-734) */
void kosk_host_sha3_256(uint8_t out[32], const uint8_t *in, size_t inlen);
void kosk_host_shake256(uint8_t *out, size_t outlen, const uint8_t *in, size_t inlen);
/* sha3_256 of `count` equal-length messages with the SIMD multi-buffer code of the Fiat-Shamir rounds;
 * returns the SIMD width used (8 AVX-512, 4 AVX2, 1 scalar) */
int kosk_host_sha3_256_multi(uint8_t *out, const uint8_t *in, size_t in_stride, size_t inlen, int count, int nthreads);
/* Lagrange coefficient tables the reference reads through utils/precomputed_kyber.h:10-13:
 * which = 0: share_coeff_ddeg [1303][407], 1: recon_coeff_ddeg [256][407], 2: recon_coeff_2ddeg [256][813] */
int kosk_lagrange_table(int which, uint16_t *out);

#ifdef __cplusplus
}
#endif
#endif /* KOSK_MI355X_H */
