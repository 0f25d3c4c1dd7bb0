// How long a Fiat-Shamir phase of the host takes against its 343-permutation chain, and what the worker pool's wake-up costs:
//   g++ -O2 -std=c++17 -pthread -Impcith_kyber_kosk_amd/csrc tools/pool_wake.cpp mpcith_kyber_kosk_amd/csrc/kosk_host.cpp -o /tmp/pool_wake
//   /tmp/pool_wake <threads> <proofs> <idle_us>
#include "kosk_host.hpp"
#include <chrono>
#include <thread>
#include <vector>
#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
using clk = std::chrono::steady_clock;
static double us(clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); }
int main(int argc, char **argv)
{
    const int nt = argc > 1 ? atoi(argv[1]) : 8, n = argc > 2 ? atoi(argv[2]) : 64, idle = argc > 3 ? atoi(argv[3]) : 500;
    kosk::Params P;
    kosk::make_params(3, P);
    kosk::Pool *pool = kosk::pool_create();
    kosk::pool_reserve(pool, nt);
    std::vector<uint8_t> digs((size_t)n * 1454 * 32);
    for (size_t i = 0; i < digs.size(); i++) digs[i] = (uint8_t)(i * 2654435761u >> 13);
    std::vector<uint16_t> al((size_t)n * 80);
    auto med = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    // (1) wake-up: time until the LAST of nt workers has started, job published after `idle` us of nothing
    std::vector<double> last_start, whole, fs;
    for (int it = 0; it < 200; it++) {
        std::this_thread::sleep_for(std::chrono::microseconds(idle));
        std::atomic<int64_t> latest{0};
        const auto t0 = clk::now();
        kosk::parallel_for(pool, nt, nt, [&](int) {
            const int64_t t = (int64_t)(us(t0, clk::now()) * 1000);
            int64_t cur = latest.load();
            while (t > cur && !latest.compare_exchange_weak(cur, t)) {}
            const auto s = clk::now();
            while (us(s, clk::now()) < 60) {} // a job long enough that nobody takes two indices
        });
        whole.push_back(us(t0, clk::now()));
        last_start.push_back(latest.load() / 1000.0);
    }
    printf("threads %d idle %d us: last worker starts after %.1f us (median), a 60 us job per thread takes %.1f us\n", nt, idle, med(last_start), med(whole));
    // (2) one Fiat-Shamir phase
    for (int it = 0; it < 100; it++) {
        std::this_thread::sleep_for(std::chrono::microseconds(idle));
        const auto t0 = clk::now();
        kosk::fs_alpha_batch(P, n, digs.data(), (size_t)1454 * 32, al.data(), 80, nt, pool);
        fs.push_back(us(t0, clk::now()));
    }
    printf("fs_alpha_batch of %d proofs on %d threads after %d us idle: %.1f us (median)\n", n, nt, idle, med(fs));
    const auto t0 = clk::now();
    for (int it = 0; it < 20; it++) kosk::fs_alpha_batch(P, 8, digs.data(), (size_t)1454 * 32, al.data(), 80, 1, pool);
    printf("one group of 8 on the calling thread: %.1f us\n", us(t0, clk::now()) / 20);
    kosk::pool_destroy(pool);
    return 0;
}
