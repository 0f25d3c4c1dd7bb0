// Kyber key generation on the GPU: kyber_keygen, kosk.cpp:4-70, and the public-key
// decoding of kyber_kosk_verify, kosk.cpp:94-110.  One thread per sponge; squeezed blocks are parsed
// straight from the state registers.
//   (seed hash, gen_matrix and noise sampling of the PROVER are roles of k_prover_pre, kosk_kernels.hip; their device
//    functions live in kosk_keygen_dev.hpp)
//   k_gen_matrix_wave SHAKE128(seed || j || i) + rej_uniform (verifier), one entry per wave  indcpa.c:124-145, :168-193
//   k_keygen_pack    t = A o NTT(s) * R^-1 * R + NTT(e), Barrett; pk / sk bytes    kosk.cpp:39-69, poly.c:124-139
//   k_decode_pk      polyvec_frombytes + seed extraction                           kosk.cpp:94-97, poly.c:151-158
#include <hip/hip_runtime.h>

#include "kosk_device.hpp"
#include <cstdlib>
#include <utility>

#include "kosk_keccak_dev.hpp"
#include "kosk_keygen_dev.hpp"
#include "kosk_keygen_wave_dev.hpp"
#include "kosk_math.hpp"

namespace kosk {

__constant__ static const ZetaTable kZetasKg = ZetaTable();

// A[b][i][j][256] canonical from the 32-byte public seed found at seeds + b * seed_stride (the verifier's gen_matrix, kosk.cpp:98-99;
// the prover's runs as a role of k_prover_pre, kosk_kernels.hip), on the wave sponge (kw_gen_matrix): one entry per 64-thread block
// (rounds 5-6a: the lane-pair sponge, 32 entries per block: 56 us per 276 proofs, the length of one lane's chain)
__global__ __launch_bounds__(64) void k_gen_matrix_wave(const uint8_t *__restrict__ seeds, size_t seed_stride, int16_t *__restrict__ A,
                                                        size_t A_stride, int K, int n, XofGuard xof)
{
    __shared__ __attribute__((aligned(16))) uint32_t st[KW_ST_WORDS];
    __shared__ __attribute__((aligned(16))) uint8_t sq[KW_SQ_BYTES];
    const int t = blockIdx.x; // one wave per matrix entry (kosk_keygen_wave_dev.hpp); the grid is exactly n K K blocks
    const int b = t / (K * K), ij = t - b * K * K, i = ij / K, j = ij - i * K;
    __builtin_amdgcn_s_setprio(3);
    kw_gen_matrix(seeds + (size_t)b * seed_stride, false, K, i, j, A + (size_t)b * A_stride + (size_t)ij * 256, xof, st, sq);
}

// poly.c:124-139 on a pair of centred coefficients
__device__ __forceinline__ void tobytes3(uint8_t *r, int32_t c0, int32_t c1)
{
    const uint32_t t0 = gf_encode(c0), t1 = gf_encode(c1);
    r[0] = (uint8_t)t0;
    r[1] = (uint8_t)((t0 >> 8) | (t1 << 4));
    r[2] = (uint8_t)(t1 >> 4);
}

// blockIdx.x = polynomial i, thread = degree-1 factor.  sehat = NTT(s) then NTT(e), centred int16.
__global__ __launch_bounds__(128) void k_keygen_pack(const int16_t *__restrict__ A, size_t A_stride, const int16_t *__restrict__ sehat,
                                                     size_t sehat_stride, const uint8_t *__restrict__ seeds, size_t seed_stride,
                                                     uint16_t *__restrict__ t_out,
                                                     uint8_t *__restrict__ pk, size_t pk_stride, uint8_t *__restrict__ shat_bytes,
                                                     size_t sb_stride, int K)
{
    const int p = threadIdx.x, i = blockIdx.x, b = blockIdx.y;
    const int32_t zeta = (p & 1) ? -(int32_t)kZetasKg.z[64 + (p >> 1)] : (int32_t)kZetasKg.z[64 + (p >> 1)];
    const int16_t *Ai = A + (size_t)b * A_stride + (size_t)i * K * 256;
    const int16_t *sh = sehat + (size_t)b * sehat_stride;
    int32_t r0 = 0, r1 = 0;
    for (int l = 0; l < K; l++) {
        const int32_t a0 = Ai[l * 256 + 2 * p], a1 = Ai[l * 256 + 2 * p + 1];
        const int32_t b0 = sh[l * 256 + 2 * p], b1 = sh[l * 256 + 2 * p + 1];
        r0 += fqmul(fqmul(a1, b1), zeta) + fqmul(a0, b0);
        r1 += fqmul(a0, b1) + fqmul(a1, b0);
    }
    constexpr int32_t f = (int32_t)((1ULL << 32) % Q);
    const int32_t e0 = sh[(K + i) * 256 + 2 * p], e1 = sh[(K + i) * 256 + 2 * p + 1];
    r0 = barrett_reduce((int16_t)(montgomery_reduce(barrett_reduce((int16_t)r0) * f) + e0));
    r1 = barrett_reduce((int16_t)(montgomery_reduce(barrett_reduce((int16_t)r1) * f) + e1));
    t_out[((size_t)b * K + i) * 256 + 2 * p] = (uint16_t)gf_encode(r0);
    t_out[((size_t)b * K + i) * 256 + 2 * p + 1] = (uint16_t)gf_encode(r1);
    tobytes3(pk + (size_t)b * pk_stride + 384 * i + 3 * p, r0, r1);
    tobytes3(shat_bytes + (size_t)b * sb_stride + 384 * i + 3 * p, sh[i * 256 + 2 * p], sh[i * 256 + 2 * p + 1]);
    if (i == 0 && p < 32) pk[(size_t)b * pk_stride + 384 * K + p] = seeds[(size_t)b * seed_stride + p];
}

// verifier: t (12-bit values as stored) from the pk bytes
__global__ __launch_bounds__(128) void k_decode_pk(const uint8_t *__restrict__ pk, size_t pk_stride, uint16_t *__restrict__ t_out, int K)
{
    const int p = threadIdx.x, i = blockIdx.x, b = blockIdx.y;
    const uint8_t *a = pk + (size_t)b * pk_stride + 384 * i + 3 * p;
    t_out[((size_t)b * K + i) * 256 + 2 * p] = (uint16_t)(((uint32_t)a[0] | ((uint32_t)a[1] << 8)) & 0xFFF);
    t_out[((size_t)b * K + i) * 256 + 2 * p + 1] = (uint16_t)(((uint32_t)(a[1] >> 4) | ((uint32_t)a[2] << 4)) & 0xFFF);
}

hipError_t launch_keygen_pack(const int16_t *A, size_t A_stride, const int16_t *sehat, size_t sehat_stride, const uint8_t *seeds,
                              size_t seed_stride, uint16_t *t_out, uint8_t *pk, size_t pk_stride, uint8_t *shat_bytes, size_t sb_stride, int K, int n,
                              hipStream_t st)
{
    hipLaunchKernelGGL(k_keygen_pack, dim3(K, n), dim3(128), 0, st, A, A_stride, sehat, sehat_stride, seeds, seed_stride, t_out, pk, pk_stride,
                       shat_bytes, sb_stride, K);
    return hipGetLastError();
}
hipError_t launch_decode_pk(const uint8_t *pk, size_t pk_stride, uint16_t *t_out, int16_t *A, size_t A_stride, int K, int n, hipStream_t st,
                            XofGuard xof)
{
    hipLaunchKernelGGL(k_decode_pk, dim3(K, n), dim3(128), 0, st, pk, pk_stride, t_out, K);
    // gen_matrix from the seed stored behind the packed t (kosk.cpp:96-99)
    hipLaunchKernelGGL(k_gen_matrix_wave, dim3(n * K * K), dim3(64), 0, st, pk + 384 * K, pk_stride, A, A_stride, K, n, xof);
    return hipGetLastError();
}

} // namespace kosk
