// Probe (not product code): does a workgroup keep its LDS to itself when its allocation sits high in the 160 KiB
// (next to a 112 KiB workgroup of another kernel), for 16-bit, 32-bit and 128-bit DS accesses?
#include <hip/hip_runtime.h>
#include <cstdio>
template <int BYTES, typename T>
__global__ __launch_bounds__(256) void k_hold(unsigned tag, int spins, unsigned *errs)
{
    __shared__ __attribute__((aligned(16))) unsigned char raw[BYTES];
    T *lds = reinterpret_cast<T *>(raw);
    constexpr int N = BYTES / sizeof(T);
    auto val = [&](int i) { T v; unsigned x = tag ^ (unsigned)i * 2654435761u; if constexpr (sizeof(T) == 16) v = T{x, x + 1, x + 2, x + 3}; else v = (T)x; return v; };
    auto neq = [&](T a, T b) { if constexpr (sizeof(T) == 16) return a.x != b.x || a.y != b.y || a.z != b.z || a.w != b.w; else return a != b; };
    for (int i = threadIdx.x; i < N; i += 256) lds[i] = val(i);
    __syncthreads();
    unsigned bad = 0;
    for (int s = 0; s < spins; s++) {
        for (int i = threadIdx.x; i < N; i += 256) bad += neq(lds[i], val(i));
        __syncthreads();
        for (int i = threadIdx.x; i < N; i += 256) lds[i] = val(i); // keep writing too
        __syncthreads();
    }
    if (bad) atomicAdd(errs, bad);
}
int main()
{
    unsigned *e; hipMalloc(&e, 64); hipMemset(e, 0, 64);
    hipStream_t a, b; hipStreamCreateWithFlags(&a, hipStreamNonBlocking); hipStreamCreateWithFlags(&b, hipStreamNonBlocking);
    for (int rep = 0; rep < 60; rep++) {
        hipLaunchKernelGGL((k_hold<114688, uint4>), dim3(220), dim3(256), 0, a, 0xA5A50000u, 10, e);
        hipLaunchKernelGGL((k_hold<21504, unsigned short>), dim3(512), dim3(256), 0, b, 0x5A5A0000u, 40, e + 1);
        hipLaunchKernelGGL((k_hold<21504, uint4>), dim3(512), dim3(256), 0, b, 0x3C3C0000u, 40, e + 2);
        hipLaunchKernelGGL((k_hold<21504, float>), dim3(512), dim3(256), 0, b, 0x0F0F0000u, 40, e + 3);
    }
    hipDeviceSynchronize();
    unsigned h[4]; hipMemcpy(h, e, 16, hipMemcpyDeviceToHost);
    printf("mismatches: 112 KiB/b128 %u | 21 KiB next to it: u16 %u, b128 %u, f32 %u\n", h[0], h[1], h[2], h[3]);
}
