// Compact wire format of the proof: the reference ships the raw struct image of mpcith_proof
// (mlwe_prover.cpp:540-543, 0.65-0.73 MB), in which every share is a u16 below q = 3329.  The compact form keeps the 24
// fields in the same order and packs each u16 field two values into three bytes (12 bits each, value 2i in the low 12
// bits: byte0 = a & 0xFF, byte1 = (a >> 8) | ((b & 0xF) << 4), byte2 = b >> 4 -- the bit order of Kyber's poly_tobytes,
// poly.c:128-147); the two digest fields stay raw; every field starts on a 16-byte boundary.  Lossless for any image whose
// u16 values are below 4096 (every image an honest prover emits; a larger value makes compression fail, not wrap).
// K=3: 680 980 -> 531 760 bytes (78 %).  Packing runs on the GPU in front of the D2H copy of kosk_fetch_proofs_compact and
// unpacking behind the H2D copy of kosk_stage_verifier_inputs_compact, so PCIe carries the compact bytes.
#include <hip/hip_runtime.h>

#include <cstring>

#include "kosk_ctx.hpp"

namespace kosk {

#define HIPCHK(x) KOSK_HIPCHK(x)

CompactPlan make_compact_plan(const Params &P)
{
    CompactPlan cp{};
    size_t o = 0;
    for (int f = 0; f < NFIELDS; f++) {
        cp.f[f].src_off = (uint32_t)P.off[f];
        cp.f[f].dst_off = (uint32_t)o;
        cp.f[f].raw = (f == F_TCOMM || f == F_COMM);
        cp.f[f].n = (uint32_t)(cp.f[f].raw ? P.size[f] : P.size[f] / 2); // bytes (raw) or u16 values
        const size_t bytes = cp.f[f].raw ? P.size[f] : P.size[f] / 2 * 3 / 2;
        o += (bytes + 15) / 16 * 16;
    }
    cp.bytes = o;
    return cp;
}

// host codec (same layout), for callers that hold images in host memory
int compact_encode(const Params &P, const uint8_t *img, uint8_t *out)
{
    const CompactPlan cp = make_compact_plan(P);
    memset(out, 0, cp.bytes);
    for (int f = 0; f < NFIELDS; f++) {
        const CompactField &cf = cp.f[f];
        if (cf.raw) { memcpy(out + cf.dst_off, img + cf.src_off, cf.n); continue; }
        const uint8_t *s = img + cf.src_off;
        uint8_t *d = out + cf.dst_off;
        for (uint32_t i = 0; i + 1 < cf.n; i += 2) {
            const uint32_t a = s[2 * i] | (s[2 * i + 1] << 8), b = s[2 * i + 2] | (s[2 * i + 3] << 8);
            if (a >= 4096 || b >= 4096) return -1;
            d[0] = (uint8_t)a; d[1] = (uint8_t)((a >> 8) | ((b & 0xF) << 4)); d[2] = (uint8_t)(b >> 4);
            d += 3;
        }
    }
    return 0;
}
void compact_decode(const Params &P, const uint8_t *in, uint8_t *img)
{
    const CompactPlan cp = make_compact_plan(P);
    for (int f = 0; f < NFIELDS; f++) {
        const CompactField &cf = cp.f[f];
        if (cf.raw) { memcpy(img + cf.src_off, in + cf.dst_off, cf.n); continue; }
        const uint8_t *s = in + cf.dst_off;
        uint8_t *d = img + cf.src_off;
        for (uint32_t i = 0; i + 1 < cf.n; i += 2) {
            const uint32_t a = s[0] | ((s[1] & 0xF) << 8), b = (s[1] >> 4) | (s[2] << 4);
            d[2 * i] = (uint8_t)a; d[2 * i + 1] = (uint8_t)(a >> 8); d[2 * i + 2] = (uint8_t)b; d[2 * i + 3] = (uint8_t)(b >> 8);
            s += 3;
        }
    }
}

// one thread per 8 values (16 image bytes as four aligned u32 -> 12 compact bytes as three aligned u32), or per 16 raw bytes
__global__ __launch_bounds__(256) void k_pack_proofs(const uint8_t *__restrict__ img, size_t image_stride, uint8_t *__restrict__ out,
                                                     size_t out_stride, CompactPlan cp, uint32_t *__restrict__ bad)
{
    const CompactField cf = cp.f[blockIdx.y];
    const uint8_t *s = img + (size_t)blockIdx.z * image_stride + cf.src_off;
    uint8_t *d = out + (size_t)blockIdx.z * out_stride + cf.dst_off;
    const uint32_t units = cf.raw ? (cf.n + 15) / 16 : (cf.n + 7) / 8;
    bool overflow = false;
    for (uint32_t u = blockIdx.x * 256 + threadIdx.x; u < units; u += gridDim.x * 256) {
        if (cf.raw) { // field sizes are multiples of 32
            *reinterpret_cast<uint4 *>(d + 16 * u) = *reinterpret_cast<const uint4 *>(s + 16 * u);
            continue;
        }
        const uint32_t left = cf.n - 8 * u; // values from here on (even)
        if (left >= 8) {
            const uint32_t *sw = reinterpret_cast<const uint32_t *>(s + 16 * u);
            const uint32_t w0 = sw[0], w1 = sw[1], w2 = sw[2], w3 = sw[3];
            overflow |= ((w0 | w1 | w2 | w3) & 0xF000F000u) != 0;
            // 8 values a0..a7 of 12 bits -> 96 bits, little-endian bit stream
            const uint32_t a0 = w0 & 0xFFF, a1 = (w0 >> 16) & 0xFFF, a2 = w1 & 0xFFF, a3 = (w1 >> 16) & 0xFFF;
            const uint32_t a4 = w2 & 0xFFF, a5 = (w2 >> 16) & 0xFFF, a6 = w3 & 0xFFF, a7 = (w3 >> 16) & 0xFFF;
            uint32_t *dw = reinterpret_cast<uint32_t *>(d + 12 * u);
            dw[0] = a0 | (a1 << 12) | (a2 << 24);
            dw[1] = (a2 >> 8) | (a3 << 4) | (a4 << 16) | (a5 << 28);
            dw[2] = (a5 >> 4) | (a6 << 8) | (a7 << 20);
        } else {
            for (uint32_t i = 0; i + 1 < left; i += 2) {
                const uint16_t *sv = reinterpret_cast<const uint16_t *>(s + 16 * u) + i;
                const uint32_t a = sv[0], b = sv[1];
                overflow |= (a | b) >= 4096;
                uint8_t *db = d + 12 * u + 3 * (i / 2);
                db[0] = (uint8_t)a; db[1] = (uint8_t)((a >> 8) | ((b & 0xF) << 4)); db[2] = (uint8_t)(b >> 4);
            }
        }
    }
    if (overflow) atomicOr(&bad[blockIdx.z], 1u);
}

__global__ __launch_bounds__(256) void k_unpack_proofs(const uint8_t *__restrict__ in, size_t in_stride, uint8_t *__restrict__ img,
                                                       size_t image_stride, CompactPlan cp)
{
    const CompactField cf = cp.f[blockIdx.y];
    const uint8_t *s = in + (size_t)blockIdx.z * in_stride + cf.dst_off;
    uint8_t *d = img + (size_t)blockIdx.z * image_stride + cf.src_off;
    const uint32_t units = cf.raw ? (cf.n + 15) / 16 : (cf.n + 7) / 8;
    for (uint32_t u = blockIdx.x * 256 + threadIdx.x; u < units; u += gridDim.x * 256) {
        if (cf.raw) {
            *reinterpret_cast<uint4 *>(d + 16 * u) = *reinterpret_cast<const uint4 *>(s + 16 * u);
            continue;
        }
        const uint32_t left = cf.n - 8 * u;
        if (left >= 8) {
            const uint32_t *sw = reinterpret_cast<const uint32_t *>(s + 12 * u);
            const uint32_t x0 = sw[0], x1 = sw[1], x2 = sw[2];
            const uint32_t a0 = x0 & 0xFFF, a1 = (x0 >> 12) & 0xFFF, a2 = (x0 >> 24) | ((x1 & 0xF) << 8), a3 = (x1 >> 4) & 0xFFF;
            const uint32_t a4 = (x1 >> 16) & 0xFFF, a5 = (x1 >> 28) | ((x2 & 0xFF) << 4), a6 = (x2 >> 8) & 0xFFF, a7 = x2 >> 20;
            uint32_t *dw = reinterpret_cast<uint32_t *>(d + 16 * u);
            dw[0] = a0 | (a1 << 16); dw[1] = a2 | (a3 << 16); dw[2] = a4 | (a5 << 16); dw[3] = a6 | (a7 << 16);
        } else {
            for (uint32_t i = 0; i + 1 < left; i += 2) {
                const uint8_t *sb = s + 12 * u + 3 * (i / 2);
                uint16_t *dv = reinterpret_cast<uint16_t *>(d + 16 * u) + i;
                dv[0] = (uint16_t)(sb[0] | ((sb[1] & 0xF) << 8));
                dv[1] = (uint16_t)((sb[1] >> 4) | (sb[2] << 4));
            }
        }
    }
}

static int ensure_compact(Ctx &c)
{
    if (c.d_compact) return 0;
    c.cplan = make_compact_plan(c.P);
    c.compact_stride = (c.cplan.bytes + 63) / 64 * 64;
    // a view keeps its own staging, sized for its own callers' batches (the compact calls are never merged)
    HIPCHK(hipMalloc(reinterpret_cast<void **>(&c.d_compact), (size_t)c.own_batch * c.compact_stride));
    HIPCHK(hipMalloc(reinterpret_cast<void **>(&c.d_compact_bad), sizeof(uint32_t) * c.own_batch));
    HIPCHK(hipHostMalloc(reinterpret_cast<void **>(&c.h_compact), (size_t)c.own_batch * c.compact_stride, hipHostMallocDefault));
    HIPCHK(hipHostMalloc(reinterpret_cast<void **>(&c.h_compact_bad), sizeof(uint32_t) * c.own_batch, hipHostMallocDefault));
    return 0;
}

int fetch_proofs_compact(Ctx &c, int n, uint8_t *out, bool direct)
{
    if (n < 1 || n > c.own_batch) { c.err = "batch size out of range"; return -1; }
    HIPCHK(hipSetDevice(c.device));
    if (ensure_compact(c)) return -1;
    HIPCHK(hipMemsetAsync(c.d_compact, 0, (size_t)n * c.compact_stride, c.stream)); // padding bytes are zero
    HIPCHK(hipMemsetAsync(c.d_compact_bad, 0, sizeof(uint32_t) * n, c.stream));
    hipLaunchKernelGGL(k_pack_proofs, dim3(16, NFIELDS, n), dim3(256), 0, c.stream, c.d_proof, c.image_stride, c.d_compact, c.compact_stride,
                       c.cplan, c.d_compact_bad);
    HIPCHK(hipGetLastError());
    // direct: `out` is page-locked host memory (the caller's own, or locked for this call by kosk_capi.cpp): no staging copy
    if (direct) HIPCHK(hipMemcpy2DAsync(out, c.cplan.bytes, c.d_compact, c.compact_stride, c.cplan.bytes, n, hipMemcpyDeviceToHost, c.stream));
    else HIPCHK(hipMemcpyAsync(c.h_compact, c.d_compact, (size_t)n * c.compact_stride, hipMemcpyDeviceToHost, c.stream));
    HIPCHK(hipMemcpyAsync(c.h_compact_bad, c.d_compact_bad, sizeof(uint32_t) * n, hipMemcpyDeviceToHost, c.stream));
    HIPCHK(stream_sync(c));
    c.path_n[direct ? PATH_COPY_DIRECT : PATH_COPY_STAGED]++;
    for (int b = 0; b < n; b++)
        if (c.h_compact_bad[b]) { c.err = "a resident proof holds a value >= 4096: not representable in the compact format"; return -1; }
    if (!direct)
        parallel_for(c.pool, n, c.nthreads, [&](int b) { memcpy(out + (size_t)b * c.cplan.bytes, c.h_compact + (size_t)b * c.compact_stride, c.cplan.bytes); });
    return 0;
}

int stage_verifier_inputs_compact(Ctx &c, int n, const uint8_t *in, const uint8_t *pk, bool direct)
{
    if (n < 1 || n > c.own_batch) { c.err = "batch size out of range"; return -1; }
    HIPCHK(hipSetDevice(c.device));
    if (ensure_verify_workspace(c)) return -1;
    if (ensure_compact(c)) return -1;
    const Params &P = c.P;
    parallel_for(c.pool, n, c.nthreads, [&](int b) {
        memcpy(c.h_pk + (size_t)b * c.pk_stride, pk + (size_t)b * P.pk_bytes, P.pk_bytes);
        if (!direct) memcpy(c.h_compact + (size_t)b * c.compact_stride, in + (size_t)b * c.cplan.bytes, c.cplan.bytes);
    });
    c.path_n[direct ? PATH_COPY_DIRECT : PATH_COPY_STAGED]++;
    HIPCHK(hipMemcpyAsync(c.d_pk, c.h_pk, (size_t)n * c.pk_stride, hipMemcpyHostToDevice, c.stream));
    c.resident_pk_n = n;
    if (direct) HIPCHK(hipMemcpy2DAsync(c.d_compact, c.compact_stride, in, c.cplan.bytes, c.cplan.bytes, n, hipMemcpyHostToDevice, c.stream));
    else HIPCHK(hipMemcpyAsync(c.d_compact, c.h_compact, (size_t)n * c.compact_stride, hipMemcpyHostToDevice, c.stream));
    hipLaunchKernelGGL(k_unpack_proofs, dim3(16, NFIELDS, n), dim3(256), 0, c.stream, c.d_compact, c.compact_stride, c.d_proof, c.image_stride, c.cplan);
    HIPCHK(hipGetLastError());
    HIPCHK(launch_decode_pk(c.d_pk, c.pk_stride, c.d_t, c.d_A, c.key_stride, P.K, n, c.stream, c.xof_guard()));
    HIPCHK(stream_sync(c));
    if (device_error_check(c)) return -1;
    return 0;
}

} // namespace kosk
