// Probe (not product code): does a stream of tiny kernels on stream B slow an L2-resident kernel on stream A
// (kernel-boundary cache maintenance is per device, not per stream)?
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <thread>
__global__ void k_reader(const uint4 *x, size_t n, int reps, unsigned *out)
{
    unsigned acc = 0;
    for (int r = 0; r < reps; r++)
        for (size_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
            const uint4 v = x[i];
            acc += v.x ^ v.y ^ v.z ^ v.w;
        }
    if (acc == 0x12345678u) *out = acc;
}
__global__ void k_tiny(unsigned *p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[1] += 1; }
int main()
{
    for (size_t mb : {8, 64, 512}) {
        const size_t n = mb * 1024 * 1024 / 16;
        uint4 *x; unsigned *out, *p;
        hipMalloc(&x, n * 16); hipMemset(x, 1, n * 16); hipMalloc(&out, 64); hipMalloc(&p, 64); hipMemset(p, 0, 64);
        hipStream_t a, b; hipStreamCreateWithFlags(&a, hipStreamNonBlocking); hipStreamCreateWithFlags(&b, hipStreamNonBlocking);
        const int reps = (int)(2048 / mb) + 1;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int with = 0; with < 2; with++) {
            std::atomic<bool> stop{false};
            std::thread th;
            if (with) th = std::thread([&] { hipSetDevice(0); while (!stop) { for (int i = 0; i < 64; i++) hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, b, p); hipStreamSynchronize(b); } });
            hipLaunchKernelGGL(k_reader, dim3(2048), dim3(256), 0, a, x, n, reps, out); hipStreamSynchronize(a);
            hipEventRecord(e0, a);
            for (int i = 0; i < 5; i++) hipLaunchKernelGGL(k_reader, dim3(2048), dim3(256), 0, a, x, n, reps, out);
            hipEventRecord(e1, a); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("%4zu MB x %4d passes: %8.1f us per kernel, %.0f GB/s %s\n", mb, reps, ms / 5 * 1e3, (double)n * 16 * reps / (ms / 5 * 1e-3) / 1e9, with ? "(tiny kernels streaming on another stream)" : "(alone)");
            if (with) { stop = true; th.join(); }
        }
        hipFree(x);
    }
}
