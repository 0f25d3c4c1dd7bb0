// Probe (not product code): issue rate of single VALU opcodes via inline asm (the compiler cannot fold these).
#include <hip/hip_runtime.h>
#include <cstdio>
#define OPS(X) X(0, "v_xor_b32 %0, %0, %1\n v_xor_b32 %0, %0, %2") X(1, "v_bfi_b32 %0, %0, %1, %2") X(2, "v_alignbit_b32 %0, %0, %1, 7") \
    X(3, "v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96") X(4, "v_xor_b32 %0, %0, %1") X(5, "v_and_or_b32 %0, %0, %1, %2") \
    X(6, "v_perm_b32 %0, %0, %1, %2") X(7, "v_mad_i32_i24 %0, %0, %1, %2") X(8, "v_lshl_or_b32 %0, %0, 3, %1") \
    X(9, "v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf") X(10, "v_pk_add_i16 %0, %0, %1") \
    X(11, "v_pk_mul_lo_u16 %0, %0, %1") X(12, "v_mul_lo_u32 %0, %0, %1") X(13, "v_mul_hi_u32 %0, %0, %1") X(14, "v_pk_mad_i16 %0, %0, %1, %2")
template <int OP>
__global__ void k(int *out, int iters)
{
    int x[8], a = threadIdx.x * 3 + 1, b = threadIdx.x ^ 0x5a5a;
    for (int i = 0; i < 8; i++) x[i] = threadIdx.x + i;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int i = 0; i < 8; i++) {
#define X(id, s) if (OP == id) asm volatile(s : "+v"(x[i]) : "v"(a), "v"(b));
                OPS(X)
#undef X
            }
    }
    int s = 0;
    for (int i = 0; i < 8; i++) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int OP>
float run(int blocks, int iters, int *out)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, iters); hipDeviceSynchronize();
    hipEventRecord(e0); hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main()
{
    int *out; hipMalloc(&out, 256 * 8 * 256 * 4);
    const char *names[] = {"2x v_xor_b32", "v_bfi_b32", "v_alignbit_b32", "v_bitop3_b32", "v_xor_b32", "v_and_or_b32", "v_perm_b32", "v_mad_i32_i24",
                           "v_lshl_or_b32", "v_mov_b32_dpp", "v_pk_add_i16", "v_pk_mul_lo_u16", "v_mul_lo_u32", "v_mul_hi_u32", "v_pk_mad_i16"};
    for (int wps : {1, 2, 8}) {
        const int blocks = 256 * wps, iters = 2000;
        float ms[15];
#define X(id, s) ms[id] = run<id>(blocks, iters, out);
        OPS(X)
#undef X
        for (int op = 0; op < 15; op++) {
            const double cyc = 256.0 * 4 * 2.4e9 * (ms[op] * 1e-3) / ((double)blocks * 4 * iters * 32); // SIMD-cycles per wave-instruction at 2.4 GHz
            printf("%d waves/SIMD %-16s %.2f cycles/wave-instr (%.1f T lane-op/s)\n", wps, names[op], cyc * wps, (double)blocks * 256 * iters * 32 / (ms[op] * 1e-3) / 1e12);
        }
    }
}
