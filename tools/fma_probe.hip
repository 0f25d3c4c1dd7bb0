// Probe (not product code): does a floating-point FMA return wrong values while matrix-core waves share the chip?
// tools/ntt_lab.hip narrowed the packed-fp32 NTT's rare wrong polynomials down to "fp32 FMA instructions + a concurrent
// MFMA kernel" (mul and add kept apart: clean; integer: clean; other load generators: clean).  This probe removes the
// NTT: every lane evaluates ONE instruction form on fixed operands many times and compares each result bitwise with the
// first one, on one stream, while a load generator runs on another.
// Build: hipcc -O3 -std=c++20 --offload-arch=gfx950 tools/fma_probe.hip -o tools/fma_probe ; run: tools/fma_probe <seconds per cell>
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <thread>

#define CK(x)                                                                              \
    do {                                                                                   \
        hipError_t e_ = (x);                                                               \
        if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } \
    } while (0)

typedef float v2f __attribute__((ext_vector_type(2)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));

// FORM 0: v_fma_f32 d = a*b + c, operands as in the NTT's rint step (|a| < 2^24, b = 1/3329, c = 1.5 * 2^23)
// FORM 1: v_fma_f32 on generic operands (same exponent range)
// FORM 2: v_pk_fma_f32 (both halves)
// FORM 3: v_mul_f32 then v_add_f32 (control)
// FORM 4: v_mad_u32_u24 (integer control)
// FORM 5: v_fmac_f32 (VOP2 form, accumulator = destination)
// FORM 6: the NTT's reduction as a DEPENDENT chain in asm: k1 = fma(p, 1/q, magic); k = k1 - magic; r = fma(-k, q, p)
// FORM 7: the same chain written in C (hipcc schedules and contracts it), 4 chains interleaved
// FORM 8: the same on packed v2f values (v_pk_fma_f32 chains)
// FORM 9: v_fma_f32 with the multiplier in an SGPR (how hipcc emits the NTT's constants)
// FORM 10: v_pk_fma_f32 with an SGPR-pair multiplier broadcast by op_sel_hi:[1,0,1] and a negated first source
// FORM 11: v_pk_mul_f32 by an SGPR pair, then v_pk_fma_f32 with an SGPR pair, then v_pk_add_f32 with an SGPR pair (one NTT butterfly's opening)
template <int FORM>
__global__ __launch_bounds__(256) void k_probe(unsigned long long *bad, uint32_t *first_bad, int iters)
{
    const int gid = blockIdx.x * 256 + threadIdx.x;
    float a = (float)((gid * 2654435761u) >> 9) * ((gid & 1) ? 1.0f : -1.0f), b = 1.0f / 3329.0f, c = 12582912.0f;
    if (FORM == 1 || FORM == 2 || FORM == 5) { b = 1.0f + (float)(gid % 977) / 1024.0f; c = (float)(gid % 4093) * 0.37f - 700.0f; }
    uint32_t ia = (gid * 40503u) & 0xFFFFFF, ib = 0x3A5A5 ^ (gid & 0xFFFF), ic = gid;
    unsigned long long nbad = 0;
    uint32_t want0 = 0, want1 = 0, got_bad = 0;
    for (int it = 0; it < iters; it++) {
        uint32_t r0 = 0, r1 = 0;
        if constexpr (FORM == 0 || FORM == 1) {
            float r;
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
            r0 = __float_as_uint(r);
        } else if constexpr (FORM == 2) {
            v2f r, va = {a, -a * 0.5f}, vb = {b, b * 1.25f}, vc = {c, c + 1.0f};
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(va), "v"(vb), "v"(vc));
            r0 = __float_as_uint(r.x); r1 = __float_as_uint(r.y);
        } else if constexpr (FORM == 3) {
            float m, r;
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(m) : "v"(a), "v"(b));
            asm volatile("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(m), "v"(c));
            r0 = __float_as_uint(r);
        } else if constexpr (FORM == 4) {
            asm volatile("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r0) : "v"(ia), "v"(ib), "v"(ic));
        } else if constexpr (FORM == 5) {
            float r = c;
            asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(r) : "v"(a), "v"(b));
            r0 = __float_as_uint(r);
        } else if constexpr (FORM == 6) {
            float k1, k, r;
            const float qinv = 1.0f / 3329.0f, magic = 12582912.0f, q = 3329.0f;
            asm volatile("v_fma_f32 %0, %3, %4, %5\n\tv_sub_f32 %1, %0, %5\n\tv_fma_f32 %2, -%1, %6, %3"
                         : "=&v"(k1), "=&v"(k), "=&v"(r) : "v"(a), "v"(qinv), "v"(magic), "v"(q));
            r0 = __float_as_uint(r);
        } else if constexpr (FORM == 7) {
            float acc = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const float p = a + (float)(j * 4099);
                const float k = __fmaf_rn(p, 1.0f / 3329.0f, 12582912.0f) - 12582912.0f;
                acc += __fmaf_rn(-k, 3329.0f, p) * (float)(j + 1);
            }
            asm volatile("" : "+v"(acc));
            r0 = __float_as_uint(acc);
        } else if constexpr (FORM == 9) {
            float r;
            const float qinv = 1.0f / 3329.0f;
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(qinv), "v"(c));
            r0 = __float_as_uint(r);
        } else if constexpr (FORM == 10) {
            v2f r, vk = {a, a * 0.5f}, vp = {c, c + 3.0f};
            const v2f sq = {3329.0f, 3329.0f};
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1] neg_lo:[1,0,0] neg_hi:[1,0,0]" : "=v"(r) : "v"(vk), "s"(sq), "v"(vp));
            r0 = __float_as_uint(r.x); r1 = __float_as_uint(r.y);
        } else if constexpr (FORM == 11) {
            v2f m, k1, k, r, vh = {a, a * 0.5f + 3.0f}, vmagic = {12582912.0f, 12582912.0f};
            const v2f sz = {1729.0f, 1729.0f}, sqinv = {1.0f / 3329.0f, 1.0f / 3329.0f}, snm = {-12582912.0f, -12582912.0f}, sq = {3329.0f, 3329.0f};
            asm volatile("v_pk_mul_f32 %0, %4, %5 op_sel_hi:[1,0]\n\ts_nop 0\n\tv_pk_fma_f32 %1, %0, %6, %7 op_sel_hi:[1,0,0]\n\ts_nop 0\n\t"
                         "v_pk_add_f32 %2, %1, %8 op_sel_hi:[1,0]\n\ts_nop 0\n\tv_pk_fma_f32 %3, %2, %9, %0 op_sel_hi:[1,0,1] neg_lo:[1,0,0] neg_hi:[1,0,0]"
                         : "=&v"(m), "=&v"(k1), "=&v"(k), "=&v"(r) : "v"(vh), "s"(sz), "s"(sqinv), "v"(vmagic), "s"(snm), "s"(sq));
            r0 = __float_as_uint(r.x); r1 = __float_as_uint(r.y);
        } else {
            v2f p = {a, a * 0.75f + 11.0f};
            asm volatile("" : "+v"(p));
            v2f t = p;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const v2f k = (t * (1.0f / 3329.0f) + 12582912.0f) - 12582912.0f;
                t = (t - k * 3329.0f) * 1777.0f + p;
            }
            asm volatile("" : "+v"(t));
            r0 = __float_as_uint(t.x); r1 = __float_as_uint(t.y);
        }
        if (it == 0) { want0 = r0; want1 = r1; }
        else if (r0 != want0 || r1 != want1) { nbad++; got_bad = r0 != want0 ? r0 : r1; }
    }
    if (nbad) {
        atomicAdd(bad, nbad);
        if (atomicCAS(&first_bad[0], 0u, 1u) == 0u) { first_bad[1] = want0; first_bad[2] = got_bad; first_bad[3] = gid; first_bad[4] = __float_as_uint(a); }
    }
}

__global__ void g_mfma_i8(int *out, int iters)
{
    v4i a = {(int)threadIdx.x, 2, 3, 4}, b = {5, 6, 7, (int)blockIdx.x}, c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    for (int i = 0; i < iters; i++) {
        c0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(b, a, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, a, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_i32_16x16x64_i8(b, b, c3, 0, 0, 0);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}
__global__ void g_mfma_bf16(float *out, int iters)
{
    v8bf a, b;
    for (int i = 0; i < 8; i++) { a[i] = (__bf16)(0.01f * (threadIdx.x + i)); b[i] = (__bf16)(0.02f * (blockIdx.x % 7 + i)); }
    v4f c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    for (int i = 0; i < iters; i++) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, a, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, a, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, b, c3, 0, 0, 0);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}
__global__ void g_valu(uint32_t *out, int iters)
{
    uint32_t x = threadIdx.x * 2654435761u + blockIdx.x, y = x ^ 0x9e3779b9u, z = x + 77;
    for (int i = 0; i < iters; i++) { x = __builtin_amdgcn_bitop3_b32(x, y, z, 0x96); y = __builtin_amdgcn_alignbit(y, x, 7); z = __builtin_amdgcn_bitop3_b32(z, x, y, 0xD2); }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x ^ y ^ z;
}

int main(int argc, char **argv)
{
    const double secs = argc > 1 ? atof(argv[1]) : 3.0;
    unsigned long long *bad;
    uint32_t *fb, *lout;
    CK(hipMalloc(&bad, 8));
    CK(hipMalloc(&fb, 32));
    CK(hipMalloc(&lout, 4096 * 256 * 4));
    hipStream_t st, ls;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&ls, hipStreamNonBlocking));
    const char *forms[] = {"v_fma_f32 (rint operands)", "v_fma_f32 (generic)", "v_pk_fma_f32", "v_mul_f32 + v_add_f32", "v_mad_u32_u24", "v_fmac_f32",
                           "reduction chain (asm)", "reduction chains (C)", "packed reduction chains (C)",
                           "v_fma_f32, SGPR multiplier", "v_pk_fma_f32, SGPR pair", "butterfly opening, SGPR pairs"};
    const int first_form = argc > 2 ? atoi(argv[2]) : 0;
    const char *loads[] = {"none", "mfma-i8, 4 waves/CU", "mfma-i8, 16 waves/CU", "mfma-bf16, 16 waves/CU", "valu-int, 16 waves/CU"};
    printf("%-28s %-24s %12s %16s %14s  first mismatch\n", "instruction", "concurrent load", "launches", "results", "wrong");
    for (int f = first_form; f < 12; f++)
        for (int ld = 0; ld < 5; ld++) {
            std::atomic<bool> stop{false};
            std::thread gen([&] {
                CK(hipSetDevice(0));
                while (!stop.load()) {
                    if (ld == 1) hipLaunchKernelGGL(g_mfma_i8, dim3(256), dim3(256), 0, ls, (int *)lout, 20000);
                    if (ld == 2) hipLaunchKernelGGL(g_mfma_i8, dim3(1024), dim3(256), 0, ls, (int *)lout, 5000);
                    if (ld == 3) hipLaunchKernelGGL(g_mfma_bf16, dim3(1024), dim3(256), 0, ls, (float *)lout, 5000);
                    if (ld == 4) hipLaunchKernelGGL(g_valu, dim3(1024), dim3(256), 0, ls, lout, 40000);
                    (void)hipStreamSynchronize(ls);
                    if (ld == 0) std::this_thread::sleep_for(std::chrono::milliseconds(1));
                }
            });
            CK(hipMemsetAsync(bad, 0, 8, st));
            CK(hipMemsetAsync(fb, 0, 32, st));
            long launches = 0;
            const int blocks = 1024, iters = 4000;
            const auto t0 = std::chrono::steady_clock::now();
            while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < secs) {
                if (f == 0) hipLaunchKernelGGL(k_probe<0>, dim3(blocks), dim3(256), 0, st, bad, fb, iters);
                if (f == 1) hipLaunchKernelGGL(k_probe<1>, dim3(blocks), dim3(256), 0, st, bad, fb, iters);
                if (f == 2) hipLaunchKernelGGL(k_probe<2>, dim3(blocks), dim3(256), 0, st, bad, fb, iters);
                if (f == 3) hipLaunchKernelGGL(k_probe<3>, dim3(blocks), dim3(256), 0, st, bad, fb, iters);
                if (f == 4) hipLaunchKernelGGL(k_probe<4>, dim3(blocks), dim3(256), 0, st, bad, fb, iters);
                if (f == 5) hipLaunchKernelGGL(k_probe<5>, dim3(blocks), dim3(256), 0, st, bad, fb, iters);
                if (f == 6) hipLaunchKernelGGL(k_probe<6>, dim3(blocks), dim3(256), 0, st, bad, fb, iters);
                if (f == 7) hipLaunchKernelGGL(k_probe<7>, dim3(blocks), dim3(256), 0, st, bad, fb, iters);
                if (f == 8) hipLaunchKernelGGL(k_probe<8>, dim3(blocks), dim3(256), 0, st, bad, fb, iters);
                if (f == 9) hipLaunchKernelGGL(k_probe<9>, dim3(blocks), dim3(256), 0, st, bad, fb, iters);
                if (f == 10) hipLaunchKernelGGL(k_probe<10>, dim3(blocks), dim3(256), 0, st, bad, fb, iters);
                if (f == 11) hipLaunchKernelGGL(k_probe<11>, dim3(blocks), dim3(256), 0, st, bad, fb, iters);
                CK(hipStreamSynchronize(st));
                launches++;
            }
            stop.store(true);
            gen.join();
            unsigned long long hb = 0;
            uint32_t hf[8] = {0};
            CK(hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost));
            CK(hipMemcpy(hf, fb, 32, hipMemcpyDeviceToHost));
            printf("%-28s %-24s %12ld %16.3e %14llu", forms[f], loads[ld], launches, (double)launches * blocks * 256 * (iters - 1), hb);
            if (hf[0]) printf("  thread %u a=0x%08x want 0x%08x got 0x%08x", hf[3], hf[4], hf[1], hf[2]);
            printf("\n");
            fflush(stdout);
        }
    return 0;
}
