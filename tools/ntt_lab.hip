// Probe (not product code): root-causing the packed-fp32 NTT that round 1 dropped ("a wrong polynomial about once per
// 1000 launches under concurrent load, cause not found").  The kernel below is that variant (git 7f6d171) with the
// arithmetic style as a template parameter; it runs on one stream while a chosen load generator runs on others, every
// output is compared with a host NTT, and every mismatch is logged with its position and value so that the error CLASS
// is visible (reduction off by q? stale LDS data of another polynomial? a whole wave? random bits?).
//   ARITH 0: packed fp32 as written in 7f6d171 (v2f operators; hipcc contracts mul+add into v_pk_fma_f32)
//   ARITH 1: the same on scalar floats (v_fma_f32 / v_mul_f32 / v_add_f32, no packed opcodes)
//   ARITH 2: packed, but every multiply-add kept apart (no FMA contraction: v_pk_mul_f32 + v_pk_add_f32)
//   ARITH 3: integer Montgomery butterflies in the same data flow (control)
//   ARITH 4: as 0, but every constant and zeta is forced into VGPRs (no SGPR / literal source operands on the packed ops)
// Build: hipcc -O3 -std=c++20 --offload-arch=gfx950 tools/ntt_lab.hip -o tools/ntt_lab
// Run:   tools/ntt_lab <seconds per cell>
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#define CK(x)                                                                              \
    do {                                                                                   \
        hipError_t e_ = (x);                                                               \
        if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } \
    } while (0)

constexpr int Q = 3329;
typedef float v2f __attribute__((ext_vector_type(2)));

struct Zetas {
    float zf[128];   // plain roots 17^bitrev7(k) mod q, centred
    int16_t zm[128]; // Montgomery form (kyber/ntt.c:39-56)
};
static Zetas make_zetas()
{
    Zetas z{};
    int pw[128];
    pw[0] = 1;
    for (int i = 1; i < 128; i++) pw[i] = pw[i - 1] * 17 % Q;
    for (int i = 0; i < 128; i++) {
        int br = 0;
        for (int b = 0; b < 7; b++) br |= ((i >> b) & 1) << (6 - b);
        int v = pw[br];
        if (v > Q / 2) v -= Q;
        z.zf[i] = (float)v;
        int m = pw[br] * 2285 % Q;
        if (m > Q / 2) m -= Q;
        z.zm[i] = (int16_t)m;
    }
    return z;
}
__constant__ Zetas kZ;

constexpr float QF = (float)Q, QINVF = 1.0f / (float)Q, MAGIC = 12582912.0f;
constexpr int PPB = 16, FSTRIDE = 16 * 20 + 16;

template <int ARITH>
__device__ __forceinline__ v2f red(v2f p)
{
    if constexpr (ARITH == 4) {
        v2f qi = {QINVF, QINVF}, mg = {MAGIC, MAGIC}, qq = {QF, QF};
        asm volatile("" : "+v"(qi), "+v"(mg), "+v"(qq));
        const v2f k = (p * qi + mg) - mg;
        return p - k * qq;
    } else if constexpr (ARITH == 1) {
        const float k0 = __fsub_rn(__fmaf_rn(p.x, QINVF, MAGIC), MAGIC), k1 = __fsub_rn(__fmaf_rn(p.y, QINVF, MAGIC), MAGIC);
        return (v2f){__fmaf_rn(-k0, QF, p.x), __fmaf_rn(-k1, QF, p.y)};
    } else if constexpr (ARITH == 2) {
        v2f t = p * QINVF;
        asm volatile("" : "+v"(t));
        v2f k = t + MAGIC;
        asm volatile("" : "+v"(k));
        k = k - MAGIC;
        v2f kq = k * QF;
        asm volatile("" : "+v"(kq));
        return p - kq;
    } else {
        const v2f k = (p * QINVF + MAGIC) - MAGIC;
        return p - k * QF;
    }
}
template <int ARITH>
__device__ __forceinline__ void bfly(v2f &lo, v2f &hi, v2f z)
{
    v2f m;
    if constexpr (ARITH == 4) asm volatile("" : "+v"(z));
    if constexpr (ARITH == 1) m = (v2f){__fmul_rn(hi.x, z.x), __fmul_rn(hi.y, z.y)};
    else m = hi * z;
    if constexpr (ARITH == 2) asm volatile("" : "+v"(m));
    const v2f t = red<ARITH>(m);
    hi = lo - t;
    lo = lo + t;
}
__device__ __forceinline__ int mont(int a)
{
    const int t = (int16_t)((int16_t)a * (int16_t)-3327);
    return (a - t * Q) >> 16;
}
__device__ __forceinline__ void bfly_i(int &lo, int &hi, int z)
{
    const int t = mont(z * hi);
    hi = lo - t;
    lo = lo + t;
}

// in / out: npoly x 256 int16, canonical output
template <int ARITH>
__global__ __launch_bounds__(256) void k_ntt(const int16_t *__restrict__ in, uint16_t *__restrict__ out, int npoly)
{
    __shared__ __attribute__((aligned(16))) float ldsf[PPB * FSTRIDE];
    int16_t *lds16 = reinterpret_cast<int16_t *>(ldsf);
    const int tid = threadIdx.x, p0 = blockIdx.x * PPB;
    for (int c = tid; c < PPB * 32; c += 256) {
        const int pl = c >> 5, ch = c & 31, p = p0 + pl;
        if (p < npoly) *reinterpret_cast<uint4 *>(lds16 + pl * 2 * FSTRIDE + ch * 8) = *reinterpret_cast<const uint4 *>(in + (size_t)p * 256 + ch * 8);
    }
    __syncthreads();
    const int pl = tid >> 4, l = tid & 15;
    const int16_t *mine16 = lds16 + pl * 2 * FSTRIDE;
    float *minef = ldsf + pl * FSTRIDE;
    const int p = p0 + pl;
    uint32_t w[8];
    if constexpr (ARITH == 3) {
        int r[16];
#pragma unroll
        for (int i = 0; i < 16; i++) r[i] = mine16[l + 16 * i];
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 8; i++) bfly_i(r[i], r[i + 8], kZ.zm[1]);
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int i = 0; i < 4; i++) bfly_i(r[8 * b + i], r[8 * b + i + 4], kZ.zm[2 + b]);
#pragma unroll
        for (int b = 0; b < 4; b++)
#pragma unroll
            for (int i = 0; i < 2; i++) bfly_i(r[4 * b + i], r[4 * b + i + 2], kZ.zm[4 + b]);
#pragma unroll
        for (int b = 0; b < 8; b++) bfly_i(r[2 * b], r[2 * b + 1], kZ.zm[8 + b]);
#pragma unroll
        for (int i = 0; i < 16; i++) minef[l + 20 * i] = (float)r[i];
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 16; i++) r[i] = (int)minef[20 * l + i];
        const int z8 = kZ.zm[16 + l];
#pragma unroll
        for (int c = 0; c < 8; c++) bfly_i(r[c], r[c + 8], z8);
        const int z4a = kZ.zm[32 + 2 * l], z4b = kZ.zm[33 + 2 * l];
#pragma unroll
        for (int c = 0; c < 4; c++) { bfly_i(r[c], r[c + 4], z4a); bfly_i(r[8 + c], r[12 + c], z4b); }
#pragma unroll
        for (int q = 0; q < 4; q++) { const int z2 = kZ.zm[64 + 4 * l + q]; bfly_i(r[4 * q], r[4 * q + 2], z2); bfly_i(r[4 * q + 1], r[4 * q + 3], z2); }
#pragma unroll
        for (int q = 0; q < 8; q++) {
            int x0 = r[2 * q] % Q, x1 = r[2 * q + 1] % Q;
            x0 += x0 < 0 ? Q : 0; x1 += x1 < 0 ? Q : 0;
            w[q] = (uint32_t)x0 | ((uint32_t)x1 << 16);
        }
    } else {
        v2f P[8];
#pragma unroll
        for (int q = 0; q < 8; q++) P[q] = (v2f){(float)mine16[l + 32 * q], (float)mine16[l + 32 * q + 16]};
        __syncthreads();
        {
            const v2f z = {kZ.zf[1], kZ.zf[1]};
#pragma unroll
            for (int q = 0; q < 4; q++) bfly<ARITH>(P[q], P[q + 4], z);
        }
#pragma unroll
        for (int b = 0; b < 2; b++) {
            const v2f z = {kZ.zf[2 + b], kZ.zf[2 + b]};
#pragma unroll
            for (int q = 0; q < 2; q++) bfly<ARITH>(P[4 * b + q], P[4 * b + q + 2], z);
        }
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const v2f z = {kZ.zf[4 + b], kZ.zf[4 + b]};
            bfly<ARITH>(P[2 * b], P[2 * b + 1], z);
        }
#pragma unroll
        for (int m = 0; m < 4; m++) {
            v2f lo = {P[2 * m].x, P[2 * m + 1].x}, hi = {P[2 * m].y, P[2 * m + 1].y};
            const v2f z = {kZ.zf[8 + 2 * m], kZ.zf[8 + 2 * m + 1]};
            bfly<ARITH>(lo, hi, z);
            lo = red<ARITH>(lo);
            hi = red<ARITH>(hi);
            minef[l + 20 * (4 * m)] = lo.x;
            minef[l + 20 * (4 * m + 1)] = hi.x;
            minef[l + 20 * (4 * m + 2)] = lo.y;
            minef[l + 20 * (4 * m + 3)] = hi.y;
        }
        __syncthreads();
        {
            const float4 *src = reinterpret_cast<const float4 *>(minef + 20 * l);
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const float4 v = src[q];
                P[2 * q] = (v2f){v.x, v.y};
                P[2 * q + 1] = (v2f){v.z, v.w};
            }
        }
        {
            const v2f z8 = {kZ.zf[16 + l], kZ.zf[16 + l]};
#pragma unroll
            for (int q = 0; q < 4; q++) bfly<ARITH>(P[q], P[q + 4], z8);
            const v2f z4a = {kZ.zf[32 + 2 * l], kZ.zf[32 + 2 * l]}, z4b = {kZ.zf[33 + 2 * l], kZ.zf[33 + 2 * l]};
#pragma unroll
            for (int q = 0; q < 2; q++) { bfly<ARITH>(P[q], P[q + 2], z4a); bfly<ARITH>(P[4 + q], P[6 + q], z4b); }
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const v2f z2 = {kZ.zf[64 + 4 * l + q], kZ.zf[64 + 4 * l + q]};
                bfly<ARITH>(P[2 * q], P[2 * q + 1], z2);
            }
        }
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const v2f rr = red<ARITH>(P[q]);
            int x0 = (int)rr.x, x1 = (int)rr.y;
            x0 = ((x0 % Q) + Q) % Q; x1 = ((x1 % Q) + Q) % Q; // integer canonicalisation: the float path ends at the residue
            w[q] = (uint32_t)x0 | ((uint32_t)x1 << 16);
        }
    }
    if (p < npoly) {
        uint4 *o = reinterpret_cast<uint4 *>(out + (size_t)p * 256 + 16 * l);
        o[0] = make_uint4(w[0], w[1], w[2], w[3]);
        o[1] = make_uint4(w[4], w[5], w[6], w[7]);
    }
}

// ---- load generators (each launch runs ~100-200 us on the whole chip) ----
__global__ void g_valu(uint32_t *out, int iters)
{
    uint32_t x = threadIdx.x * 2654435761u + blockIdx.x, y = x ^ 0x9e3779b9u, z = x + 77;
    for (int i = 0; i < iters; i++) {
        x = __builtin_amdgcn_bitop3_b32(x, y, z, 0x96); y = __builtin_amdgcn_alignbit(y, x, 7); z = __builtin_amdgcn_bitop3_b32(z, x, y, 0xD2);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x ^ y ^ z;
}
typedef int v4i __attribute__((ext_vector_type(4)));
__global__ void g_mfma(int *out, int iters)
{
    v4i a = {(int)threadIdx.x, 2, 3, 4}, b = {5, 6, 7, (int)blockIdx.x}, c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    for (int i = 0; i < iters; i++) {
        c0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(b, a, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, a, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_i32_16x16x64_i8(b, b, c3, 0, 0, 0);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}
__global__ void g_pkf32(float *out, int iters)
{
    v2f x = {(float)threadIdx.x, 1.5f}, y = {0.999f, 1.0001f}, z = {0.25f, -0.125f};
    for (int i = 0; i < iters; i++) { x = x * y + z; z = z * y - x; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x.x + x.y + z.x + z.y;
}
__global__ void g_stream(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

static void host_ntt(const int16_t *in, uint16_t *out, const Zetas &z)
{
    int r[256];
    for (int i = 0; i < 256; i++) r[i] = in[i];
    int k = 1;
    for (int len = 128; len >= 2; len >>= 1)
        for (int start = 0; start < 256; start += 2 * len) {
            const long zeta = (long)z.zf[k++];
            for (int j = start; j < start + len; j++) {
                long t = zeta * r[j + len] % Q;
                r[j + len] = (int)(((long)r[j] - t) % Q);
                r[j] = (int)(((long)r[j] + t) % Q);
            }
        }
    for (int i = 0; i < 256; i++) out[i] = (uint16_t)(((r[i] % Q) + Q) % Q);
}

int main(int argc, char **argv)
{
    const double secs = argc > 1 ? atof(argv[1]) : 4.0;
    const int npoly = 4096;
    Zetas z = make_zetas();
    CK(hipMemcpyToSymbol(HIP_SYMBOL(kZ), &z, sizeof z));
    std::vector<int16_t> hin((size_t)npoly * 256);
    uint32_t s = 12345;
    for (auto &v : hin) { s = s * 1664525u + 1013904223u; v = (int16_t)((s >> 8) % Q); }
    for (int i = 0; i < 256; i++) hin[i] = (int16_t)(i & 1 ? 3328 : 0), hin[256 + i] = 3328; // edge polynomials
    std::vector<uint16_t> ref((size_t)npoly * 256), hout(ref.size());
    for (int p = 0; p < npoly; p++) host_ntt(&hin[(size_t)p * 256], &ref[(size_t)p * 256], z);

    int16_t *din;
    uint16_t *dout;
    CK(hipMalloc(&din, hin.size() * 2));
    CK(hipMalloc(&dout, ref.size() * 2));
    CK(hipMemcpy(din, hin.data(), hin.size() * 2, hipMemcpyHostToDevice));
    uint16_t *hpin;
    CK(hipHostMalloc(&hpin, ref.size() * 2));
    uint32_t *lout;
    CK(hipMalloc(&lout, 2048 * 256 * 4));
    uint4 *sa, *sb;
    const size_t sn = (size_t)64 << 20; // 1 GiB each way
    CK(hipMalloc(&sa, sn * 16));
    CK(hipMalloc(&sb, sn * 16));
    CK(hipMemset(sa, 1, sn * 16));
    hipStream_t st, ls[3];
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    for (auto &x : ls) CK(hipStreamCreateWithFlags(&x, hipStreamNonBlocking));

    const char *loads[] = {"none", "valu-int", "mfma-i8", "packed-fp32", "hbm-stream", "all"};
    const char *ariths[] = {"packed fp32 (as 7f6d171)", "scalar fp32", "packed, no FMA contraction", "integer Montgomery", "packed fp32, VGPR operands only"};
    const int only_load = argc > 2 ? atoi(argv[2]) : -1;
    printf("%-28s %-12s %10s %10s %12s\n", "arithmetic", "load", "launches", "bad", "bad coeffs");
    for (int ar = 0; ar < 5; ar++)
        for (int ld = 0; ld < 6; ld++) {
            if (only_load >= 0 && ld != only_load) continue;
            std::atomic<bool> stop{false};
            std::thread gen([&] {
                CK(hipSetDevice(0));
                while (!stop.load()) {
                    if (ld == 1 || ld == 5) hipLaunchKernelGGL(g_valu, dim3(2048), dim3(256), 0, ls[0], lout, 20000);
                    if (ld == 2 || ld == 5) hipLaunchKernelGGL(g_mfma, dim3(1024), dim3(256), 0, ls[1], (int *)lout, 4000);
                    if (ld == 3) hipLaunchKernelGGL(g_pkf32, dim3(2048), dim3(256), 0, ls[0], (float *)lout, 20000);
                    if (ld == 4 || ld == 5) hipLaunchKernelGGL(g_stream, dim3(2048), dim3(256), 0, ls[2], sa, sb, sn / 8);
                    for (auto &x : ls) (void)hipStreamSynchronize(x);
                    if (ld == 0) std::this_thread::sleep_for(std::chrono::milliseconds(1));
                }
            });
            long launches = 0, bad = 0, badc = 0;
            int shown = 0;
            const auto t0 = std::chrono::steady_clock::now();
            while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < secs) {
                CK(hipMemsetAsync(dout, 0xEE, ref.size() * 2, st));
                const dim3 grid((npoly + PPB - 1) / PPB);
                if (ar == 0) hipLaunchKernelGGL(k_ntt<0>, grid, dim3(256), 0, st, din, dout, npoly);
                else if (ar == 1) hipLaunchKernelGGL(k_ntt<1>, grid, dim3(256), 0, st, din, dout, npoly);
                else if (ar == 2) hipLaunchKernelGGL(k_ntt<2>, grid, dim3(256), 0, st, din, dout, npoly);
                else if (ar == 3) hipLaunchKernelGGL(k_ntt<3>, grid, dim3(256), 0, st, din, dout, npoly);
                else hipLaunchKernelGGL(k_ntt<4>, grid, dim3(256), 0, st, din, dout, npoly);
                CK(hipMemcpyAsync(hpin, dout, ref.size() * 2, hipMemcpyDeviceToHost, st));
                CK(hipStreamSynchronize(st));
                launches++;
                if (memcmp(hpin, ref.data(), ref.size() * 2) != 0) {
                    bad++;
                    for (size_t i = 0; i < ref.size(); i++)
                        if (hpin[i] != ref[i]) {
                            badc++;
                            if (shown < 12) {
                                printf("#   mismatch: arith %d load %s launch %ld poly %zu (block %zu, wave %zu) coeff %zu got %u want %u (diff %d)\n", ar, loads[ld],
                                       launches, i / 256, i / 256 / PPB, (i / 256 % PPB) / 4, i % 256, hpin[i], ref[i], (int)hpin[i] - (int)ref[i]);
                                shown++;
                            }
                        }
                }
            }
            stop.store(true);
            gen.join();
            printf("%-28s %-12s %10ld %10ld %12ld\n", ariths[ar], loads[ld], launches, bad, badc);
            fflush(stdout);
        }
    return 0;
}
