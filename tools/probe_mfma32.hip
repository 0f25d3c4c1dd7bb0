// Probe (not product code): v_mfma_i32_32x32x32_i8 operand / result lane maps with exact integer data, its rate, and how many
// vector-ALU instructions fit beside the matrix instructions of either shape before the loop gets longer (the table product is
// bound by exactly that: profiles/r04_gemm_stamps.txt).   hipcc --offload-arch=gfx950 -O3 tools/probe_mfma32.hip -o tools/probe_mfma32
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

__global__ void k_mfma32(const int8_t *A /*32x32 row-major [m][k]*/, const int8_t *B /*[k][n]*/, int *D /*32x32 [m][n]*/)
{
    const int l = threadIdx.x;
    v4i a, b;
    v16i c = {0};
    int8_t ab[16], bb[16];
    for (int j = 0; j < 16; j++) {
        const int k = 16 * (l >> 5) + j;
        ab[j] = A[(l & 31) * 32 + k];
        bb[j] = B[k * 32 + (l & 31)];
    }
    __builtin_memcpy(&a, ab, 16);
    __builtin_memcpy(&b, bb, 16);
    c = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c, 0, 0, 0);
    for (int v = 0; v < 16; v++) D[(8 * (v / 4) + 4 * (l >> 5) + (v % 4)) * 32 + (l & 31)] = c[v];
}

// SHAPE 0: four 16x16x64 (16 384 MAC each), SHAPE 1: four 32x32x32 (32 768 MAC each) per trip, NV independent vector instructions beside them
template <int SHAPE, int NV>
__global__ __launch_bounds__(512) void k_mix(int *out, int iters)
{
    v4i a = {(int)threadIdx.x, 1, 2, 3}, b = {4, 5, 6, (int)threadIdx.x};
    v4i c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    v16i d0 = {0}, d1 = {0}, d2 = {0}, d3 = {0};
    int x[8];
    for (int i = 0; i < 8; i++) x[i] = threadIdx.x * (i + 3);
    for (int i = 0; i < iters; i++) {
        if (SHAPE == 0) {
            c0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c3, 0, 0, 0);
        } else {
            d0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, d0, 0, 0, 0);
            d1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, d1, 0, 0, 0);
            d2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, d2, 0, 0, 0);
            d3 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, d3, 0, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < NV; q++) asm volatile("v_mad_i32_i24 %0, %0, %1, %2" : "+v"(x[q & 7]) : "v"(i), "v"(q));
    }
    int s = 0;
    for (int i = 0; i < 8; i++) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + c0[0] + c1[1] + c2[2] + c3[3] + d0[0] + d1[5] + d2[9] + d3[15];
}

template <int SHAPE, int NV>
static void run_mix(int *out, hipEvent_t e0, hipEvent_t e1)
{
    const int blocks = 256, threads = 512, iters = 4000; // 8 waves per CU = 2 per SIMD, as the table product
    hipLaunchKernelGGL((k_mix<SHAPE, NV>), dim3(blocks), dim3(threads), 0, 0, out, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_mix<SHAPE, NV>), dim3(blocks), dim3(threads), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double mac = (double)blocks * (threads / 64) * iters * 4 * (SHAPE ? 32768 : 16384);
    printf("%s  + %2d VALU per 4 MFMA: %.3f ms, %.2f P MAC/s, %.1f ns per trip and wave pair\n", SHAPE ? "32x32x32" : "16x16x64", NV, ms, mac / (ms * 1e-3) / 1e15,
           ms * 1e6 / iters);
}

int main()
{
    std::vector<int8_t> A(32 * 32), B(32 * 32);
    for (int i = 0; i < 32; i++) for (int k = 0; k < 32; k++) A[i * 32 + k] = (int8_t)((i * 7 + k * 3 + 1) % 61 - 30);
    for (int k = 0; k < 32; k++) for (int n = 0; n < 32; n++) B[k * 32 + n] = (int8_t)((k * 5 + n * 11 + 2) % 53 - 26);
    int8_t *dA, *dB; int *dD;
    hipMalloc(&dA, A.size()); hipMalloc(&dB, B.size()); hipMalloc(&dD, 1024 * 4);
    hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size(), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_mfma32, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    std::vector<int> D(1024);
    hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 32; i++) for (int n = 0; n < 32; n++) {
        int s = 0;
        for (int k = 0; k < 32; k++) s += (int)A[i * 32 + k] * (int)B[k * 32 + n];
        bad += s != D[i * 32 + n];
    }
    printf("mfma_i32_32x32x32_i8 layout (A[l&31][16(l>>5)+j], B[16(l>>5)+j][l&31], D[8(v/4)+4(l>>5)+v%%4][l&31]): %s (%d mismatches)\n", bad ? "WRONG" : "OK", bad);
    int *out; hipMalloc(&out, 256 * 512 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    run_mix<0, 0>(out, e0, e1); run_mix<0, 4>(out, e0, e1); run_mix<0, 8>(out, e0, e1); run_mix<0, 16>(out, e0, e1); run_mix<0, 32>(out, e0, e1);
    run_mix<1, 0>(out, e0, e1); run_mix<1, 8>(out, e0, e1); run_mix<1, 16>(out, e0, e1); run_mix<1, 32>(out, e0, e1); run_mix<1, 64>(out, e0, e1); run_mix<1, 96>(out, e0, e1);
    return 0;
}
