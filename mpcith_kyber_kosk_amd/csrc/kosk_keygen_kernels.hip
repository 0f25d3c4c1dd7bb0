// Kyber key generation on the GPU: kyber_keygen, kosk.cpp:4-70, and the public-key
// decoding of kyber_kosk_verify, kosk.cpp:94-110.  One thread per sponge; squeezed blocks are parsed
// from a per-thread LDS byte buffer so that the rejection sampler / CBD read plain bytes.
//   k_keygen_seeds   sha3_512(d || K) -> public seed || noise seed                 kosk.cpp:12-14
//   k_gen_matrix     SHAKE128(seed || j || i) + rej_uniform                        indcpa.c:124-145, :168-193
//   k_noise          SHAKE256(noise seed || nonce) + cbd2 / cbd3                   poly.c:225-230, cbd.c:58-107
//   k_keygen_pack    t = A o NTT(s) * R^-1 * R + NTT(e), Barrett; pk / sk bytes    kosk.cpp:39-69, poly.c:124-139
//   k_decode_pk      polyvec_frombytes + seed extraction                           kosk.cpp:94-97, poly.c:151-158
#include <hip/hip_runtime.h>

#include "kosk_device.hpp"
#include "kosk_keccak_dev.hpp"
#include "kosk_math.hpp"

namespace kosk {

__constant__ static const ZetaTable kZetasKg = ZetaTable();

// byte i of the sponge state
__device__ __forceinline__ uint8_t kbyte(const KState &s, int lane, int byte)
{
    const uint32_t w = byte < 4 ? s.lo[lane] : s.hi[lane];
    return (uint8_t)(w >> (8 * (byte & 3)));
}

// absorb up to 40 bytes of (seed32 || extra bytes) into a fresh state, with padding for `rate` bytes
__device__ __forceinline__ void absorb_seed(KState &s, const uint8_t *seed32, const uint8_t *extra, int nextra, int rate, uint8_t dom)
{
    kstate_zero(s);
    uint8_t m[40];
#pragma unroll
    for (int i = 0; i < 40; i++) m[i] = 0;
    for (int i = 0; i < 32; i++) m[i] = seed32[i];
    for (int i = 0; i < nextra; i++) m[32 + i] = extra[i];
    m[32 + nextra] = dom;
#pragma unroll
    for (int l = 0; l < 5; l++) {
        uint32_t lo = 0, hi = 0;
#pragma unroll
        for (int b = 0; b < 4; b++) {
            lo |= (uint32_t)m[8 * l + b] << (8 * b);
            hi |= (uint32_t)m[8 * l + 4 + b] << (8 * b);
        }
        s.lo[l] = lo;
        s.hi[l] = hi;
    }
    const int last = rate / 8 - 1;
#pragma unroll
    for (int l = 0; l < 25; l++)
        if (l == last) s.hi[l] ^= 0x80000000u;
}

// dump the first nlanes lanes of the state as bytes
template <int NLANES>
__device__ __forceinline__ void dump_lanes(const KState &s, uint8_t *dst)
{
#pragma unroll
    for (int l = 0; l < NLANES; l++) {
        uint32_t *d = reinterpret_cast<uint32_t *>(dst + 8 * l);
        d[0] = s.lo[l];
        d[1] = s.hi[l];
    }
}

__global__ __launch_bounds__(64) void k_keygen_seeds(const uint8_t *__restrict__ tape, size_t tape_stride, uint8_t *__restrict__ seeds, int K, int n)
{
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= n) return;
    KState s;
    const uint8_t kk = (uint8_t)K;
    absorb_seed(s, tape + (size_t)b * tape_stride, &kk, 1, 72, 0x06);
    keccak_f1600_dev(s);
    uint32_t *o = reinterpret_cast<uint32_t *>(seeds + (size_t)b * 64);
#pragma unroll
    for (int l = 0; l < 8; l++) { o[2 * l] = s.lo[l]; o[2 * l + 1] = s.hi[l]; }
}

// A[b][i][j][256] canonical; the XOF input is seed || j || i (gen_matrix with transposed == 0)
__global__ __launch_bounds__(64) void k_gen_matrix(const uint8_t *__restrict__ seeds, size_t seed_stride, int16_t *__restrict__ A,
                                                   size_t A_stride, int K, int n)
{
    __shared__ __attribute__((aligned(8))) uint8_t buf[64][168];
    const int t = blockIdx.x * 64 + threadIdx.x;
    if (t >= n * K * K) return;
    const int b = t / (K * K), ij = t - b * K * K, i = ij / K, j = ij - i * K;
    KState s;
    const uint8_t xy[2] = {(uint8_t)j, (uint8_t)i};
    absorb_seed(s, seeds + (size_t)b * seed_stride, xy, 2, 168, 0x1F);
    int16_t *r = A + (size_t)b * A_stride + (size_t)ij * 256;
    uint8_t *my = buf[threadIdx.x];
    int ctr = 0;
    for (int blk = 0; blk < 32 && ctr < 256; blk++) { // 3 blocks suffice with probability 1 - 2^-40; bounded anyway
        keccak_f1600_dev(s);
        dump_lanes<21>(s, my);
        for (int pos = 0; pos + 3 <= 168 && ctr < 256; pos += 3) {
            const uint32_t v0 = ((uint32_t)my[pos] | ((uint32_t)my[pos + 1] << 8)) & 0xFFF;
            const uint32_t v1 = ((uint32_t)(my[pos + 1] >> 4) | ((uint32_t)my[pos + 2] << 4)) & 0xFFF;
            if (v0 < (uint32_t)Q) r[ctr++] = (int16_t)v0;
            if (ctr < 256 && v1 < (uint32_t)Q) r[ctr++] = (int16_t)v1;
        }
    }
}

// se[b][nonce][256]: nonce < K: s, else e   (kosk.cpp:17-20)
__global__ __launch_bounds__(64) void k_noise(const uint8_t *__restrict__ seeds, int16_t *__restrict__ se, size_t se_stride, int K, int eta1, int n)
{
    __shared__ __attribute__((aligned(8))) uint8_t buf[64][272];
    const int t = blockIdx.x * 64 + threadIdx.x;
    if (t >= n * 2 * K) return;
    const int b = t / (2 * K), nonce = t - b * 2 * K;
    KState s;
    const uint8_t nn = (uint8_t)nonce;
    absorb_seed(s, seeds + (size_t)b * 64 + 32, &nn, 1, 136, 0x1F);
    uint8_t *my = buf[threadIdx.x];
    keccak_f1600_dev(s);
    dump_lanes<17>(s, my);
    if (eta1 == 3) { // 192 bytes: a second block
        keccak_f1600_dev(s);
        dump_lanes<17>(s, my + 136);
    }
    int16_t *r = se + (size_t)b * se_stride + (size_t)nonce * 256;
    if (eta1 == 2) {
        for (int w = 0; w < 32; w++) {
            const uint32_t x = (uint32_t)my[4 * w] | ((uint32_t)my[4 * w + 1] << 8) | ((uint32_t)my[4 * w + 2] << 16) | ((uint32_t)my[4 * w + 3] << 24);
            const uint32_t d = (x & 0x55555555u) + ((x >> 1) & 0x55555555u);
#pragma unroll
            for (int q = 0; q < 8; q++) r[8 * w + q] = (int16_t)(((d >> (4 * q)) & 3) - ((d >> (4 * q + 2)) & 3));
        }
    } else {
        for (int w = 0; w < 64; w++) {
            const uint32_t x = (uint32_t)my[3 * w] | ((uint32_t)my[3 * w + 1] << 8) | ((uint32_t)my[3 * w + 2] << 16);
            const uint32_t d = (x & 0x00249249u) + ((x >> 1) & 0x00249249u) + ((x >> 2) & 0x00249249u);
#pragma unroll
            for (int q = 0; q < 4; q++) r[4 * w + q] = (int16_t)(((d >> (6 * q)) & 7) - ((d >> (6 * q + 3)) & 7));
        }
    }
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
                                                     size_t sehat_stride, const uint8_t *__restrict__ seeds, uint16_t *__restrict__ t_out,
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
    if (i == 0 && p < 32) pk[(size_t)b * pk_stride + 384 * K + p] = seeds[(size_t)b * 64 + p];
}

// verifier: t (12-bit values as stored) from the pk bytes
__global__ __launch_bounds__(128) void k_decode_pk(const uint8_t *__restrict__ pk, size_t pk_stride, uint16_t *__restrict__ t_out, int K)
{
    const int p = threadIdx.x, i = blockIdx.x, b = blockIdx.y;
    const uint8_t *a = pk + (size_t)b * pk_stride + 384 * i + 3 * p;
    t_out[((size_t)b * K + i) * 256 + 2 * p] = (uint16_t)(((uint32_t)a[0] | ((uint32_t)a[1] << 8)) & 0xFFF);
    t_out[((size_t)b * K + i) * 256 + 2 * p + 1] = (uint16_t)(((uint32_t)(a[1] >> 4) | ((uint32_t)a[2] << 4)) & 0xFFF);
}

hipError_t launch_keygen(const uint8_t *tape, size_t tape_stride, uint8_t *seeds, int16_t *A, size_t A_stride, int16_t *se,
                         size_t se_stride, int K, int eta1, int n, hipStream_t st)
{
    hipLaunchKernelGGL(k_keygen_seeds, dim3((n + 63) / 64), dim3(64), 0, st, tape, tape_stride, seeds, K, n);
    hipLaunchKernelGGL(k_gen_matrix, dim3((n * K * K + 63) / 64), dim3(64), 0, st, seeds, (size_t)64, A, A_stride, K, n);
    hipLaunchKernelGGL(k_noise, dim3((n * 2 * K + 63) / 64), dim3(64), 0, st, seeds, se, se_stride, K, eta1, n);
    return hipGetLastError();
}
hipError_t launch_keygen_pack(const int16_t *A, size_t A_stride, const int16_t *sehat, size_t sehat_stride, const uint8_t *seeds,
                              uint16_t *t_out, uint8_t *pk, size_t pk_stride, uint8_t *shat_bytes, size_t sb_stride, int K, int n,
                              hipStream_t st)
{
    hipLaunchKernelGGL(k_keygen_pack, dim3(K, n), dim3(128), 0, st, A, A_stride, sehat, sehat_stride, seeds, t_out, pk, pk_stride,
                       shat_bytes, sb_stride, K);
    return hipGetLastError();
}
hipError_t launch_decode_pk(const uint8_t *pk, size_t pk_stride, uint16_t *t_out, int16_t *A, size_t A_stride, int K, int n, hipStream_t st)
{
    hipLaunchKernelGGL(k_decode_pk, dim3(K, n), dim3(128), 0, st, pk, pk_stride, t_out, K);
    // gen_matrix from the seed stored behind the packed t (kosk.cpp:96-99)
    hipLaunchKernelGGL(k_gen_matrix, dim3((n * K * K + 63) / 64), dim3(64), 0, st, pk + 384 * K, pk_stride, A, A_stride, K, n);
    return hipGetLastError();
}

} // namespace kosk
