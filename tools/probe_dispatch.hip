// Probe (not product code): aggregate dispatch rate of small dependent kernels over S streams / host threads.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
__global__ void k_small(const unsigned short *x, unsigned *fail)
{
    const size_t o = (size_t)blockIdx.y * 6144 + blockIdx.x * 256 + threadIdx.x;
    if (x[o] == 0xFFFF) atomicOr(fail, 1u);
}
int main()
{
    unsigned short *x; unsigned *f;
    hipMalloc(&x, 46 * 6144 * 2 * 8); hipMemset(x, 0, 46 * 6144 * 2 * 8); hipMalloc(&f, 64);
    for (int blocks : {24, 1}) for (int S : {1, 2, 3, 6, 12}) {
        const int N = 4000;
        std::vector<hipStream_t> st(S);
        for (auto &s : st) hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
        auto run = [&](int i) { for (int k = 0; k < N; k++) hipLaunchKernelGGL(k_small, dim3(blocks, blocks == 1 ? 1 : 46), dim3(256), 0, st[i], x, f); hipStreamSynchronize(st[i]); };
        for (int i = 0; i < S; i++) run(i);
        auto t0 = std::chrono::steady_clock::now();
        std::vector<std::thread> th;
        for (int i = 0; i < S; i++) th.emplace_back(run, i);
        for (auto &t : th) t.join();
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        printf("grid %4d WGs  streams %2d: %.2f us per kernel per stream, %.0f k kernels/s aggregate\n", blocks == 1 ? 1 : blocks * 46, S, dt / N * 1e6, S * N / dt / 1e3);
        for (auto &s : st) hipStreamDestroy(s);
    }
}
