// Probe for DESIGN 14.9 "what is next (1)": can a stream be made to wait for a word the HOST writes (hipStreamWaitValue32), and how long after the
// write does the first queued kernel finish, against launching that kernel only after the host is ready?  Not product code.
//   hipcc --offload-arch=gfx950 -O2 tools/probe_waitvalue.hip -o tools/probe_waitvalue && tools/probe_waitvalue
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void k_touch(unsigned *p) { if (threadIdx.x == 0) atomicAdd(p, 1u); }
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    unsigned *d_cnt;
    CK(hipMalloc(&d_cnt, 4));
    CK(hipMemset(d_cnt, 0, 4));
    hipEvent_t ev;
    CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    // (a) launch after the host is ready: launch + event, spin on the event
    std::vector<double> a, b;
    for (int it = 0; it < 200; it++) {
        std::this_thread::sleep_for(std::chrono::microseconds(300));
        const double t0 = now_us();
        hipLaunchKernelGGL(k_touch, dim3(1), dim3(64), 0, st, d_cnt);
        CK(hipEventRecord(ev, st));
        while (hipEventQuery(ev) == hipErrorNotReady) {}
        a.push_back(now_us() - t0);
    }
    // (b) kernel queued behind a wait on a host-written word
    unsigned *sig = nullptr;
    hipError_t e = hipExtMallocWithFlags(reinterpret_cast<void **>(&sig), 8, hipMallocSignalMemory);
    if (e != hipSuccess) { printf("hipExtMallocWithFlags(hipMallocSignalMemory): %s\n", hipGetErrorString(e)); (void)hipGetLastError(); CK(hipHostMalloc(reinterpret_cast<void **>(&sig), 8, hipHostMallocDefault)); printf("falling back to hipHostMalloc memory for the word\n"); }
    *reinterpret_cast<volatile unsigned *>(sig) = 0;
    int ok = 0;
    for (int it = 0; it < 200; it++) {
        const unsigned want = (unsigned)it + 1;
        e = hipStreamWaitValue32(st, sig, want, hipStreamWaitValueEq, 0xFFFFFFFFu);
        if (e != hipSuccess) { printf("hipStreamWaitValue32: %s\n", hipGetErrorString(e)); break; }
        hipLaunchKernelGGL(k_touch, dim3(1), dim3(64), 0, st, d_cnt);
        CK(hipEventRecord(ev, st));
        std::this_thread::sleep_for(std::chrono::microseconds(300)); // the host's "Fiat-Shamir round"
        if (hipEventQuery(ev) != hipErrorNotReady) { printf("the stream did not wait (iteration %d)\n", it); break; }
        const double t0 = now_us();
        *reinterpret_cast<volatile unsigned *>(sig) = want;
        __sync_synchronize();
        while (hipEventQuery(ev) == hipErrorNotReady) { if (now_us() - t0 > 2e6) { printf("no release after 2 s\n"); return 2; } }
        b.push_back(now_us() - t0);
        ok++;
    }
    unsigned cnt = 0;
    CK(hipStreamSynchronize(st));
    CK(hipMemcpy(&cnt, d_cnt, 4, hipMemcpyDeviceToHost));
    auto med = [](std::vector<double> v) { if (v.empty()) return -1.0; std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    printf("launch when ready -> kernel done: median %.1f us (200 runs)\n", med(a));
    printf("queued behind hipStreamWaitValue32, host writes the word -> kernel done: median %.1f us (%d runs); kernels run %u of %d\n", med(b), ok, cnt, 200 + ok);
    return 0;
}
