// Host-side pieces of the product path: the Fiat-Shamir hashing that stays on
// the CPU by design (BASELINE.json north_star), Kyber key generation
// (host for now), Lagrange table generation, and a small
// thread pool.  None of this touches oracle/.
#pragma once
#include <cstddef>
#include <cstdint>
#include <functional>
#include <vector>

#include "kosk_params.hpp"

namespace kosk {

// fips202.c:745-774, :723-734, symmetric-shake.c:43-51
void sha3_256(uint8_t out[32], const uint8_t *in, size_t inlen);
void sha3_512(uint8_t out[64], const uint8_t *in, size_t inlen);
void shake128(uint8_t *out, size_t outlen, const uint8_t *in, size_t inlen);
void shake256(uint8_t *out, size_t outlen, const uint8_t *in, size_t inlen);
void shake256_prf(uint8_t *out, size_t outlen, const uint8_t key[32], uint8_t nonce);

// kosk.cpp:4-70.  A is [K][K][256] canonical, s/e are [K][256] small signed,
// t is [K][256] centred NTT-domain; pk/sk in Kyber wire format.
struct HostKey {
    int16_t A[MAXK * MAXK * 256];
    int16_t se[2 * MAXK * 256]; // s polys then e polys
    int16_t t[MAXK * 256];
};
void host_keygen(const Params &P, const uint8_t seed64[64], uint8_t *pk, uint8_t *sk, HostKey &key);
// kosk.cpp:94-110: decode pk into A (canonical) and t (12-bit values as stored)
void host_decode_pk(const Params &P, const uint8_t *pk, HostKey &key);

// mlwe_prover.cpp:130-142: alpha_j = BE16(SHAKE256(sha3_256(Tcomm) || 1)) % q
void fs_alpha(const Params &P, const uint8_t *tcomm_all /* [1454][32] */, uint16_t *alpha /* [J] */);
// mlwe_prover.cpp:445-474: opened list I (with the linear-probing de-dup) and its complement
void fs_opened(const uint8_t *digests_all /* [1454][32] */, uint16_t I[NOPEN], uint16_t rest[NREST]);

// Lagrange basis over n consecutive integer nodes a..a+n-1 evaluated at t
// (utils/precomputed_kyber.h:10-13; values by the call sites ss.cpp:26-27,:47,:66)
void lagrange_row(uint16_t *row, int n, int a, int t);
uint16_t gf_inv_host(uint16_t a);

// pack A[m][k] (canonical) into the limb-matrix operand format of the MFMA GEMM (kosk_device.hpp)
void pack_limb_table(const std::vector<uint16_t> &A, int M, int Kdim, int Mpad, int KS, std::vector<uint8_t> &out);
// the same limbs with each 1 KiB tile in MFMA-fragment order (lane l's 16 bytes at offset 16 l): operands loaded from global memory
void pack_frag_table(const std::vector<uint16_t> &A, int M, int Kdim, int Mpad, int KS, std::vector<uint8_t> &out);

// Persistent worker pool.  One per library context: several contexts (pipeline slots) run their
// Fiat-Shamir rounds concurrently, each on its own few threads.
class Pool;
Pool *pool_create();
void pool_destroy(Pool *);
// create the workers for jobs of up to `nthreads` threads now; returns the threads such a job will really run on (>= 1:
// thread creation may fail under a pid / thread limit, the pool then works with what it has)
int pool_reserve(Pool *, int nthreads);
// CPUs this process may run on (sched_getaffinity), the cap for every host thread count in the library
int host_cpu_count();
// run fn(i) for i in [0,n) on up to nthreads threads of `pool` (nullptr: a process-wide default pool)
void parallel_for(Pool *pool, int n, int nthreads, const std::function<void(int)> &fn);
inline void parallel_for(int n, int nthreads, const std::function<void(int)> &fn) { parallel_for(nullptr, n, nthreads, fn); }

// sha3_256 of `count` equal-length messages in[i] -> out + 32*i, `width` (4 or 8) at a time in SIMD
// lanes when the CPU has AVX2 / AVX-512F (runtime dispatch, scalar otherwise).  Used for the
// 46 528-byte Fiat-Shamir hashes (mlwe_prover.cpp:135, :449), which are sequential per proof.
void sha3_256_multi(uint8_t *out, const uint8_t *const *in, size_t len, int count);
int sha3_multi_width(); // 8 (AVX-512F), 4 (AVX2) or 1

// One proof's digest table [1454][32] on the verifier's host, put together from what the proof itself carries -- the digests of
// the 1304 unopened parties, ascending (fields Tcomm / comm of mpcith_proof, mlwe_verifier.cpp:36-38, :645-647) -- and the 150
// digests the verifier recomputed for the opened parties, in the order of the list I (:22-35, :585-632).  Memory-safe for a
// malformed I (out-of-range or repeated entries: the table is then garbage, and the proof is rejected for the list anyway).
void assemble_digest_table(uint8_t *table, const uint16_t *I, const uint8_t *unopened, const uint8_t *opened);

// batch forms of fs_alpha / fs_opened over n proofs whose digest tables are dig_stride bytes apart
// prep (optional): called for proof b on the worker that hashes it, before the hashing (the verifier assembles the table there)
void fs_alpha_batch(const Params &P, int n, const uint8_t *digs, size_t dig_stride, uint16_t *alpha, size_t alpha_stride, int nthreads, Pool *pool = nullptr,
                    const std::function<void(int)> *prep = nullptr);
// windows: also write, behind each proof's list I (at I + SEL_WIN), the NWIN + 1 boundaries of the complement's aligned 64-party windows
void fs_opened_batch(int n, const uint8_t *digs, size_t dig_stride, uint16_t *I, uint16_t *rest, size_t sel_stride, int nthreads, Pool *pool = nullptr,
                     bool windows = false, const std::function<void(int)> *prep = nullptr);

// OS entropy (kyber/randombytes.c:44-57, Linux branch)
void os_randombytes(uint8_t *out, size_t len);

} // namespace kosk
