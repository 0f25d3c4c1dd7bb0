// Probe (not product code): v_mfma_i32_16x16x64_i8 operand / result lane maps with exact integer data,
// and issue rates of the VALU instructions the kernels lean on.   hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));

__global__ void k_mfma(const int8_t *A /*16x64 row-major*/, const int8_t *B /*64x16: B[k][n]*/, int *D /*16x16*/)
{
    const int l = threadIdx.x;
    v4i a, b, c = {0, 0, 0, 0};
    int8_t ab[16], bb[16];
    for (int j = 0; j < 16; j++) {
        const int k = 16 * (l >> 4) + j;
        ab[j] = A[(l & 15) * 64 + k];
        bb[j] = B[k * 16 + (l & 15)];
    }
    __builtin_memcpy(&a, ab, 16);
    __builtin_memcpy(&b, bb, 16);
    c = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; r++) D[((l >> 4) * 4 + r) * 16 + (l & 15)] = c[r];
}

template <int OP>
__global__ void k_rate(int *out, int iters)
{
    int x0 = threadIdx.x, x1 = x0 * 3 + 1, x2 = x0 ^ 0x55, x3 = x0 + 7, x4 = x0 * 5, x5 = x0 - 3, x6 = x0 | 8, x7 = ~x0;
    typedef short s2 __attribute__((ext_vector_type(2)));
    for (int i = 0; i < iters; i++) {
#define R8(F) F(x0) F(x1) F(x2) F(x3) F(x4) F(x5) F(x6) F(x7)
        if (OP == 0) {
#define F0(v) v = __builtin_amdgcn_sdot2(__builtin_bit_cast(s2, v), __builtin_bit_cast(s2, i), v, false);
            R8(F0) R8(F0) R8(F0) R8(F0)
        } else if (OP == 1) {
#define F1(v) v = __builtin_amdgcn_bitop3_b32(v, i, x0, 0x96);
            R8(F1) R8(F1) R8(F1) R8(F1)
        } else if (OP == 2) {
#define F2(v) v = __builtin_amdgcn_alignbit(v, i, 7);
            R8(F2) R8(F2) R8(F2) R8(F2)
        } else {
#define F3(v) v = ((v << 8) >> 8) * ((i << 8) >> 8) + v;
            R8(F3) R8(F3) R8(F3) R8(F3)
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}

__global__ void k_mfma_rate(int *out, int iters)
{
    v4i a = {(int)threadIdx.x, 1, 2, 3}, b = {4, 5, 6, (int)threadIdx.x};
    v4i c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    for (int i = 0; i < iters; i++) {
        c0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c3, 0, 0, 0);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}

int main()
{
    std::vector<int8_t> A(16 * 64), B(64 * 16);
    for (int i = 0; i < 16; i++) for (int k = 0; k < 64; k++) A[i * 64 + k] = (int8_t)((i * 7 + k * 3 + 1) % 61 - 30);
    for (int k = 0; k < 64; k++) for (int n = 0; n < 16; n++) B[k * 16 + n] = (int8_t)((k * 5 + n * 11 + 2) % 53 - 26);
    int8_t *dA, *dB; int *dD;
    hipMalloc(&dA, A.size()); hipMalloc(&dB, B.size()); hipMalloc(&dD, 256 * 4);
    hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size(), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_mfma, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    std::vector<int> D(256);
    hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 16; i++) for (int n = 0; n < 16; n++) {
        int s = 0;
        for (int k = 0; k < 64; k++) s += (int)A[i * 64 + k] * (int)B[k * 16 + n];
        bad += s != D[i * 16 + n];
    }
    printf("mfma_i32_16x16x64_i8 layout (A[l&15][16(l>>4)+j], B[16(l>>4)+j][l&15], D[4(l>>4)+r][l&15]): %s (%d mismatches)\n", bad ? "WRONG" : "OK", bad);

    int *out; hipMalloc(&out, 256 * 8 * 1024 * 4 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char *names[] = {"v_dot2c_i32_i16", "v_bitop3_b32", "v_alignbit_b32", "v_mul_i24+add"};
    for (int wps = 1; wps <= 8; wps *= 2) { // waves per SIMD
        const int blocks = 256 * wps, threads = 256, iters = 4000;
        float ms[4];
        for (int op = 0; op < 4; op++) {
            auto launch = [&]() {
                if (op == 0) hipLaunchKernelGGL(k_rate<0>, dim3(blocks), dim3(threads), 0, 0, out, iters);
                if (op == 1) hipLaunchKernelGGL(k_rate<1>, dim3(blocks), dim3(threads), 0, 0, out, iters);
                if (op == 2) hipLaunchKernelGGL(k_rate<2>, dim3(blocks), dim3(threads), 0, 0, out, iters);
                if (op == 3) hipLaunchKernelGGL(k_rate<3>, dim3(blocks), dim3(threads), 0, 0, out, iters);
            };
            launch(); hipDeviceSynchronize();
            hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms[op], e0, e1);
        }
        for (int op = 0; op < 4; op++) {
            const double inst = (double)blocks * threads * iters * 32 * (op == 3 ? 2 : 1);
            printf("%d waves/SIMD  %-16s %.1f T lane-op/s\n", wps, names[op], inst / (ms[op] * 1e-3) / 1e12);
        }
        float m;
        hipLaunchKernelGGL(k_mfma_rate, dim3(blocks), dim3(threads), 0, 0, out, 2000); hipDeviceSynchronize();
        hipEventRecord(e0); hipLaunchKernelGGL(k_mfma_rate, dim3(blocks), dim3(threads), 0, 0, out, 2000); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&m, e0, e1);
        printf("%d waves/SIMD  mfma_i32_16x16x64_i8 %.1f T MAC/s\n", wps, (double)blocks * 4 * 2000 * 4 * 16384 / (m * 1e-3) / 1e12);
    }
    return 0;
}
