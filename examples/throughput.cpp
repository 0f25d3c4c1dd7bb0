// Native caller harness: what a C++ host (the reference is one: main.cpp:66-94 calls kyber_verifiable_keygen and kyber_kosk_verify
// from plain C++) runs against the C ABI -- `callers` std::threads, each with its own handle, its own resident tape bank and its own
// 46-proof calls, the two reference calls as resident library calls per step:
//     kosk_verifiable_keygen_resident   kyber_verifiable_keygen  (kosk.cpp:72-86)
//     kosk_verify_resident_pk           kyber_kosk_verify        (kosk.cpp:88-117), every verify bit asserted
// bench.py times the same calls from Python threads (its line of record); this program is the same arrangement without an
// interpreter: no GIL queue behind a merged run, no ctypes.  bench.py spawns it as a child process and reports it as
// `native_callers` next to `value`, never instead of it.
//   hipcc -std=c++17 -O2 -Iinclude examples/throughput.cpp -Lmpcith_kyber_kosk_amd -lkosk_mi355x -Wl,-rpath,'$ORIGIN/../mpcith_kyber_kosk_amd' -lpthread -o examples/throughput
//   examples/throughput [--k 3] [--batch 46] [--callers 18] [--combine 6] [--fs host|device] [--threads 3] [--steps 3600] [--warmup 180]
//                       [--tape-sets 4] [--device 0] [--blocking 0|1]
// Prints ONE JSON line.  Tapes: SHAKE256("kosk-tape-v1:<index>") as in bench.py (tapes_for), resident in HBM before the timed run.
#include <hip/hip_runtime.h>
#include <sys/resource.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "kosk_mi355x.h"

static double now_s()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
static double cpu_s()
{
    rusage ru{};
    getrusage(RUSAGE_SELF, &ru);
    return ru.ru_utime.tv_sec + ru.ru_utime.tv_usec * 1e-6 + ru.ru_stime.tv_sec + ru.ru_stime.tv_usec * 1e-6;
}
#define DIE(...) do { fprintf(stderr, __VA_ARGS__); fprintf(stderr, "\n"); exit(1); } while (0)
#define HIPOK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) DIE("%s: %s", #x, hipGetErrorString(e_)); } while (0)

struct Caller {
    kosk_ctx *h = nullptr;
    uint8_t *bank = nullptr; // device: [nsets][B][stride]
    std::vector<uint8_t> pk, sk, ok;
    long steps = 0;
    double t_prove = 0, t_verify = 0;
};

int main(int argc, char **argv)
{
    int k = 3, B = 46, S = 18, CMB = 6, threads = 3, steps = 3600, warmup = 180, nsets = 4, device = 0, blocking = -1;
    std::string fs = "host";
    for (int i = 1; i + 1 < argc; i += 2) {
        const std::string a = argv[i];
        const char *v = argv[i + 1];
        if (a == "--k") k = atoi(v);
        else if (a == "--batch") B = atoi(v);
        else if (a == "--callers") S = atoi(v);
        else if (a == "--combine") CMB = atoi(v);
        else if (a == "--threads") threads = atoi(v);
        else if (a == "--steps") steps = atoi(v);
        else if (a == "--warmup") warmup = atoi(v);
        else if (a == "--tape-sets") nsets = atoi(v);
        else if (a == "--device") device = atoi(v);
        else if (a == "--blocking") blocking = atoi(v);
        else if (a == "--fs") fs = v;
        else DIE("unknown argument %s", a.c_str());
    }
    if (k < 2 || k > 4 || B < 1 || S < 1 || CMB < 1 || nsets < 1 || steps < 1) DIE("bad arguments");
    HIPOK(hipSetDevice(device));
    const size_t tape_bytes = kosk_tape_bytes(k), stride = (tape_bytes + 63) / 64 * 64;
    const size_t pkb = kosk_pk_bytes(k), skb = kosk_sk_bytes(k);

    std::vector<Caller> cs((size_t)S);
    for (int s = 0; s < S; s++) {
        kosk_options o;
        kosk_options_init(&o);
        o.combine = CMB;
        o.combine_idle_us = 20000;                       // closed loops that pause between runs (bench.py: Slot.__init__)
        if (CMB >= 5) o.combine_prewake_us = 0;
        o.fs_mode = fs == "device" ? KOSK_FS_DEVICE : KOSK_FS_HOST;
        o.host_threads = threads;
        o.blocking_sync = blocking;
        if (kosk_create_ex(&cs[s].h, device, k, B, &o)) DIE("kosk_create_ex: %s", kosk_last_error(nullptr));
        std::vector<uint8_t> host((size_t)nsets * B * stride, 0);
        for (int t = 0; t < nsets; t++)
            for (int b = 0; b < B; b++) {
                char seed[64];
                const int n = snprintf(seed, sizeof seed, "kosk-tape-v1:%ld", ((long)s * nsets + t) * B + b);
                kosk_host_shake256(&host[((size_t)t * B + b) * stride], tape_bytes, reinterpret_cast<const uint8_t *>(seed), (size_t)n);
            }
        HIPOK(hipMalloc(reinterpret_cast<void **>(&cs[s].bank), host.size()));
        HIPOK(hipMemcpy(cs[s].bank, host.data(), host.size(), hipMemcpyHostToDevice));
        cs[s].pk.resize(pkb * B); cs[s].sk.resize(skb * B); cs[s].ok.assign((size_t)B, 0);
    }
    HIPOK(hipDeviceSynchronize());

    auto step = [&](Caller &c, long index) {
        const double t0 = now_s();
        if (kosk_verifiable_keygen_resident(c.h, B, c.bank + (size_t)(index % nsets) * B * stride, stride, c.pk.data(), c.sk.data()))
            DIE("kosk_verifiable_keygen_resident: %s", kosk_last_error(c.h));
        const double t1 = now_s();
        if (kosk_verify_resident_pk(c.h, B, nullptr, c.ok.data())) DIE("kosk_verify_resident_pk: %s", kosk_last_error(c.h));
        const double t2 = now_s();
        for (int b = 0; b < B; b++)
            if (c.ok[(size_t)b] != 1) DIE("the verifier rejected an honest proof (caller step %ld, proof %d)", index, b);
        c.t_prove += t1 - t0; c.t_verify += t2 - t1; c.steps++;
    };
    for (auto &c : cs) step(c, 0); // setup, not a benchmark step: first use allocates the verifier's workspace

    // every caller makes the same number of calls (steps dealt statically): the cohorts stay whole to the last step
    std::mutex mu;
    std::vector<double> done, lat;
    auto run = [&](long nsteps) {
        const long per = (nsteps + S - 1) / S;
        done.clear(); lat.clear();
        std::vector<std::thread> th;
        const double t_s = now_s();
        for (int s = 0; s < S; s++)
            th.emplace_back([&, s] {
                std::vector<double> d, l;
                d.reserve((size_t)per); l.reserve((size_t)per);
                for (long i = 0; i < per; i++) {
                    const double a = now_s();
                    step(cs[(size_t)s], s + i * S);
                    const double b = now_s();
                    d.push_back(b); l.push_back(b - a);
                }
                std::lock_guard<std::mutex> lk(mu);
                done.insert(done.end(), d.begin(), d.end());
                lat.insert(lat.end(), l.begin(), l.end());
            });
        for (auto &t : th) t.join();
        std::sort(done.begin(), done.end());
        return t_s;
    };
    { // conditioning: clocks, runtime pools and worker threads of a fresh process ramp for a few hundred milliseconds
        const double t0 = now_s();
        while (now_s() - t0 < 1.0) run(2L * S);
    }
    for (auto &c : cs) { c.steps = 0; c.t_prove = c.t_verify = 0; }
    const long W = std::max(0, warmup), K = steps;
    long total = W + K + S;
    total = (total + S - 1) / S * S;
    HIPOK(hipDeviceSynchronize());
    const double cpu0 = cpu_s();
    const double t_s = run(total);
    HIPOK(hipDeviceSynchronize());
    const double t_e = now_s(), cpu1 = cpu_s();
    // the timed window: exactly K steps from the completion of step W, mean over the S adjacent window positions (bench.py)
    const long D = total - W - K, nwin = std::max(1L, std::min<long>(S, D));
    double dt = 0;
    for (long j = W; j < W + nwin; j++) dt += done[(size_t)(j + K - 1)] - (j > 0 ? done[(size_t)(j - 1)] : t_s);
    dt /= (double)nwin;
    std::sort(lat.begin(), lat.end());
    long calls = 0, members = 0;
    for (auto &c : cs) {
        long a = 0, b = 0;
        kosk_combine_stats(c.h, &a, &b);
        calls += a; members += b;
    }
    long fs_dev = 0, fs_host = 0;
    for (auto &c : cs) {
        long v = 0;
        if (!kosk_path_count(c.h, 9, &v)) fs_dev += v;
        if (!kosk_path_count(c.h, 10, &v)) fs_host += v;
    }
    double tp = 0, tv = 0;
    long st = 0;
    for (auto &c : cs) { tp += c.t_prove; tv += c.t_verify; st += c.steps; }
    printf("{\"harness\": \"examples/throughput.cpp: %d std::thread callers on the C ABI, no interpreter\", \"kyber_k\": %d, \"proofs_per_call\": %d, "
           "\"callers\": %d, \"handles_per_cohort\": %d, \"fiat_shamir\": \"%s\", \"host_threads_per_caller\": %d, \"steps\": %ld, \"warmup\": %ld, "
           "\"proofs_per_s\": %.1f, \"ms_per_step\": %.5f, \"drained_proofs_per_s\": %.1f, \"step_latency_ms\": {\"median\": %.3f, \"p99\": %.3f, \"max\": %.3f, "
           "\"mean_in_keygen_call\": %.3f, \"mean_in_verify_call\": %.3f}, \"host_cpu_cores_busy\": %.2f, \"mean_callers_per_run\": %.2f, "
           "\"fs_rounds_on_device\": %ld, \"fs_rounds_on_host\": %ld, \"every_verify_bit_asserted\": true}\n",
           S, k, B, S, CMB, fs.c_str(), threads, K, W, (double)K * B / dt, dt / (double)K * 1e3, (double)total * B / (t_e - t_s),
           lat[lat.size() / 2] * 1e3, lat[(size_t)((double)lat.size() * 0.99)] * 1e3, lat.back() * 1e3, tp / (double)std::max(1L, st) * 1e3,
           tv / (double)std::max(1L, st) * 1e3, (cpu1 - cpu0) / std::max(t_e - t_s, 1e-9), calls ? (double)members / (double)calls : 0.0, fs_dev, fs_host);
    fflush(stdout);
    for (auto &c : cs) { kosk_destroy(c.h); (void)hipFree(c.bank); }
    return 0;
}
