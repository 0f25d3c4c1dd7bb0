// san_driver.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// CPU sanitizer target (the reference's CMakeLists.txt:6-7 has none; its probe build ran clean
// under ASan+UBSan).  Built by `make -C oracle asan` (-fsanitize=address,undefined) and `make -C oracle tsan`
// (-fsanitize=thread) from the oracle (kosk_oracle.c) and the product's host file (csrc/kosk_host.cpp: sponge,
// multi-buffer SHA3, keygen, Fiat-Shamir batches, Lagrange rows, thread pool); run by tests/test_sanitizers.py.
//
//   san_driver full  : K = 2,3,4: oracle keygen + prove + verify (+ one tampered proof), host code against the oracle
//   san_driver pool  : back-to-back small parallel_for calls (the hand-off race of ADVICE r1) + batched Fiat-Shamir
//   san_driver lanes : the handle's lane threads and chunk dealing (csrc/kosk_lanes.hpp: what kosk_capi.cpp's run_chunks /
//                      kosk_ctx::run execute) on fake sub-contexts: every unit exactly once, throwing and failing jobs,
//                      a pool job inside each lane (the nesting of a real batch call), many back-to-back calls
//   san_driver combine : the call combiner of a cohort (csrc/kosk_combine.hpp: what kosk_capi.cpp's merged resident calls go
//                      through) with fake executors: every call served exactly once by a run of neighbouring members with its
//                      own kind, short batches only at the end of a run, a throwing run contained, callers that alternate two
//                      kinds of call in OPPOSITE phase end up in step, a lone caller is not delayed, members that come and go
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "kosk_oracle.h"
#include "../mpcith_kyber_kosk_amd/csrc/kosk_host.hpp"
#include "../mpcith_kyber_kosk_amd/csrc/kosk_combine.hpp"
#include "../mpcith_kyber_kosk_amd/csrc/kosk_lanes.hpp"
#include <stdexcept>
#include <thread>

static int fails = 0;
#define CHECK(cond, ...)                   \
    do {                                   \
        if (!(cond)) {                     \
            fails++;                       \
            fprintf(stderr, "FAIL: ");     \
            fprintf(stderr, __VA_ARGS__);  \
            fprintf(stderr, "\n");         \
        }                                  \
    } while (0)

static std::vector<uint8_t> tape_for(int K, int idx)
{
    ko_params P;
    ko_get_params(K, &P);
    char seed[64];
    const int n = snprintf(seed, sizeof seed, "kosk-tape-v1:%d", idx);
    std::vector<uint8_t> t(P.tape_bytes);
    ko_shake256(t.data(), t.size(), (const uint8_t *)seed, (size_t)n);
    return t;
}

static void pool_hammer(int rounds)
{
    kosk::Pool *pool = kosk::pool_create();
    std::vector<int> hit(64);
    long total = 0;
    for (int r = 0; r < rounds; r++) {
        const int n = 2 + r % 13, nt = 2 + r % 5;
        std::fill(hit.begin(), hit.end(), 0);
        kosk::parallel_for(pool, n, nt, [&](int i) { hit[i]++; });
        for (int i = 0; i < n; i++) {
            CHECK(hit[i] == 1, "pool: index %d ran %d times in round %d", i, hit[i], r);
            total += hit[i];
        }
    }
    kosk::pool_destroy(pool);
    printf("pool: %d back-to-back jobs, %ld indices\n", rounds, total);
}

// fake sub-context: a capacity, its own worker pool, a record of the units it processed
struct FakeSub {
    int per;
    kosk::Pool *pool;
    std::vector<int> seen;
    long calls = 0;
};

static void lanes_hammer(int rounds)
{
    for (int S = 1; S <= 4; S++) {
        kosk::LaneSet lanes;
        lanes.create(S - 1);
        CHECK(lanes.size() == S, "lane count");
        std::vector<FakeSub> sub(S);
        for (auto &f : sub) { f.per = 2 + S % 3; f.pool = kosk::pool_create(); kosk::pool_reserve(f.pool, 3); }
        std::vector<int> hits;
        for (int r = 0; r < rounds; r++) {
            const int n = 1 + (r * 7 + S) % 23;          // ragged: n % per != 0 most of the time
            const int fail_unit = r % 5 == 3 ? (r % n) : -1;   // a failing chunk (rc -1)
            const int throw_unit = r % 7 == 5 ? (r % n) : -1;  // a throwing chunk (std::runtime_error)
            hits.assign(n, 0);
            auto fn = [&](int lane, int first, int count) -> int {
                FakeSub &f = sub[lane];
                f.calls++;
                if (count < 1 || count > f.per || first < 0 || first + count > n) return -7;
                // the nesting of a real call: a parallel_for on the sub-context's own pool inside the lane job
                kosk::parallel_for(f.pool, count, 3, [&](int i) { hits[first + i]++; });
                if (fail_unit >= first && fail_unit < first + count) return -1;
                if (throw_unit >= first && throw_unit < first + count) throw std::runtime_error("injected");
                return 0;
            };
            std::vector<std::function<int()>> jobs;
            std::vector<int> rc;
            std::vector<std::string> what;
            if (r & 1) kosk::deal_chunks(S, sub[0].per, n, fn, jobs);
            else if (n <= S * sub[0].per) kosk::deal_split(S, n, fn, jobs);
            else kosk::deal_chunks(S, sub[0].per, n, fn, jobs);
            lanes.run(jobs, rc, what);
            bool any = false;
            for (int i = 0; i < S; i++) {
                any |= rc[i] != 0;
                CHECK(rc[i] == 0 || rc[i] == -1 || rc[i] == -2, "lane rc %d", rc[i]);
                CHECK(rc[i] != -2 || what[i] == "injected", "exception text '%s'", what[i].c_str());
                CHECK(rc[i] != -7, "chunk bounds");
            }
            CHECK(any == (fail_unit >= 0 || throw_unit >= 0), "round %d: failure reporting (S=%d n=%d)", r, S, n);
            for (int u = 0; u < n; u++) CHECK(hits[u] <= 1, "unit %d processed %d times", u, hits[u]);
            if (!any)
                for (int u = 0; u < n; u++) CHECK(hits[u] == 1, "unit %d of %d not processed (S=%d round %d)", u, n, S, r);
        }
        for (auto &f : sub) kosk::pool_destroy(f.pool);
    }
    printf("lanes: %d back-to-back batch calls on 1..4 lanes, injected failures and exceptions contained\n", rounds);
}

static void host_vs_oracle(int K)
{
    ko_params OP;
    ko_get_params(K, &OP);
    kosk::Params P;
    kosk::make_params(K, P);
    CHECK(P.proof_bytes == OP.proof_bytes && P.tape_bytes == OP.tape_bytes, "params K=%d", K);

    std::vector<uint8_t> tape = tape_for(K, 0);
    std::vector<uint8_t> pk(OP.pk_bytes), sk(OP.sk_bytes), pi(OP.proof_bytes);
    ko_tape t;
    ko_tape_init(&t, tape.data(), tape.size());
    ko_trace *tr = new ko_trace();
    ko_verifiable_keygen(K, &t, pk.data(), sk.data(), pi.data(), tr);
    CHECK(!t.overrun && t.pos == OP.tape_bytes, "tape consumption K=%d", K);
    char why[128] = {0};
    CHECK(ko_kosk_verify(K, pi.data(), pk.data(), why, sizeof why) == 1, "oracle rejects its own proof K=%d: %s", K, why);
    std::vector<uint8_t> bad = pi;
    bad[OP.off[KO_F_SR] + 3] ^= 1;
    CHECK(ko_kosk_verify(K, bad.data(), pk.data(), why, sizeof why) == 0, "oracle accepts a tampered proof K=%d", K);

    // host keygen (kosk.cpp:4-70) against the oracle's
    std::vector<uint8_t> pk2(P.pk_bytes), sk2(P.sk_bytes);
    kosk::HostKey *key = new kosk::HostKey();
    kosk::host_keygen(P, tape.data(), pk2.data(), sk2.data(), *key);
    CHECK(pk2 == pk && sk2 == sk, "host_keygen differs from the oracle K=%d", K);
    delete key;

    // Fiat-Shamir rounds (mlwe_prover.cpp:130-153, :445-474), single and batched (multi-buffer SHA3 + pool)
    uint16_t alpha[kosk::MAXJ] = {0};
    kosk::fs_alpha(P, &tr->tcomm[0][0], alpha);
    CHECK(memcmp(alpha, tr->alpha, sizeof(uint16_t) * P.J) == 0, "fs_alpha K=%d", K);
    uint16_t I[kosk::NOPEN], rest[kosk::NREST];
    kosk::fs_opened(&tr->view_digest[0][0], I, rest);
    CHECK(memcmp(I, pi.data() + OP.off[KO_F_I], sizeof I) == 0, "fs_opened K=%d", K);
    const int nb = 11;
    std::vector<uint8_t> digs((size_t)nb * KO_PARTIES * 32);
    for (int b = 0; b < nb; b++) {
        memcpy(&digs[(size_t)b * KO_PARTIES * 32], &tr->view_digest[0][0], (size_t)KO_PARTIES * 32);
        digs[(size_t)b * KO_PARTIES * 32 + 5] ^= (uint8_t)b; // b = 0 stays the real table
    }
    kosk::Pool *pool = kosk::pool_create();
    std::vector<uint16_t> Ib((size_t)nb * 1312), rb((size_t)nb * 1312), al((size_t)nb * 80);
    kosk::fs_opened_batch(nb, digs.data(), (size_t)KO_PARTIES * 32, Ib.data(), rb.data(), 1312, 3, pool);
    kosk::fs_alpha_batch(P, nb, digs.data(), (size_t)KO_PARTIES * 32, al.data(), 80, 3, pool);
    CHECK(memcmp(Ib.data(), I, sizeof I) == 0, "fs_opened_batch K=%d", K);
    for (int b = 0; b < nb; b++) {
        uint16_t I1[kosk::NOPEN], r1[kosk::NREST], a1[kosk::MAXJ];
        kosk::fs_opened(&digs[(size_t)b * KO_PARTIES * 32], I1, r1);
        kosk::fs_alpha(P, &digs[(size_t)b * KO_PARTIES * 32], a1);
        CHECK(memcmp(&Ib[(size_t)b * 1312], I1, sizeof I1) == 0 && memcmp(&rb[(size_t)b * 1312], r1, sizeof r1) == 0, "fs_opened_batch[%d] K=%d", b, K);
        CHECK(memcmp(&al[(size_t)b * 80], a1, sizeof(uint16_t) * P.J) == 0, "fs_alpha_batch[%d] K=%d", b, K);
    }
    kosk::pool_destroy(pool);
    delete tr;
    printf("K=%d: oracle prove/verify/tamper, host keygen and Fiat-Shamir rounds ok\n", K);
}

static void primitives()
{
    // sponge against the oracle's on ragged lengths around the rate boundaries
    std::vector<uint8_t> msg(700);
    for (size_t i = 0; i < msg.size(); i++) msg[i] = (uint8_t)(i * 131 + 7);
    for (size_t len : {0u, 1u, 135u, 136u, 137u, 271u, 272u, 273u, 320u, 472u, 524u, 700u}) {
        uint8_t a[64], b[64];
        kosk::sha3_256(a, msg.data(), len);
        ko_sha3_256(b, msg.data(), len);
        CHECK(memcmp(a, b, 32) == 0, "sha3_256 len %zu", len);
        kosk::sha3_512(a, msg.data(), len);
        ko_sha3_512(b, msg.data(), len);
        CHECK(memcmp(a, b, 64) == 0, "sha3_512 len %zu", len);
        uint8_t x[300], y[300];
        kosk::shake256(x, sizeof x, msg.data(), len);
        ko_shake256(y, sizeof y, msg.data(), len);
        CHECK(memcmp(x, y, sizeof x) == 0, "shake256 len %zu", len);
        kosk::shake128(x, sizeof x, msg.data(), len);
        ko_shake128(y, sizeof y, msg.data(), len);
        CHECK(memcmp(x, y, sizeof x) == 0, "shake128 len %zu", len);
    }
    // multi-buffer SHA3 on 1..9 messages (exercises the partial last group)
    for (int count = 1; count <= 9; count++) {
        std::vector<const uint8_t *> in(count);
        for (int i = 0; i < count; i++) in[i] = msg.data() + i;
        std::vector<uint8_t> out((size_t)count * 32);
        kosk::sha3_256_multi(out.data(), in.data(), 472, count);
        for (int i = 0; i < count; i++) {
            uint8_t b[32];
            ko_sha3_256(b, in[i], 472);
            CHECK(memcmp(&out[(size_t)i * 32], b, 32) == 0, "sha3_256_multi %d/%d", i, count);
        }
    }
    // Lagrange rows against the oracle's tables (first, middle, last row of each)
    const uint16_t *ts = ko_table_share_ddeg(), *tr = ko_table_recon_ddeg(), *t2 = ko_table_recon_2ddeg();
    std::vector<uint16_t> row(KO_DEG2 + 1);
    for (int x : {0, 651, 1302}) {
        kosk::lagrange_row(row.data(), KO_DEG + 1, 0, KO_DEG + 1 + x);
        CHECK(memcmp(row.data(), ts + (size_t)x * (KO_DEG + 1), 2 * (KO_DEG + 1)) == 0, "share table row %d", x);
    }
    for (int i : {0, 100, 255}) {
        kosk::lagrange_row(row.data(), KO_DEG + 1, KO_NSEC, i);
        CHECK(memcmp(row.data(), tr + (size_t)i * (KO_DEG + 1), 2 * (KO_DEG + 1)) == 0, "recon table row %d", i);
        kosk::lagrange_row(row.data(), KO_DEG2 + 1, KO_NSEC, i);
        CHECK(memcmp(row.data(), t2 + (size_t)i * (KO_DEG2 + 1), 2 * (KO_DEG2 + 1)) == 0, "recon 2d table row %d", i);
    }
    printf("primitives ok\n");
}

// ---- call combiner ----
struct FakeCall { int member, seq, kind, n; int served_by = -1, run_first = -1, run_count = 0; };

static void combine_hammer(int rounds)
{
    using namespace kosk;
    const int C = 3, per = 4;
    // (1) closed loops in opposite phase: members 0,1 start with kind 0, member 2 with kind 1; each alternates 0,1,0,1...
    {
        Combiner comb(C, 200000, 100000, 300); // members pre-woken by near_end spin up to 300 us for the end of their run
        for (int i = 0; i < C; i++) CHECK(comb.join() == i, "join order");
        CHECK(comb.join() == -1, "a full cohort must refuse a fourth member");
        std::vector<std::vector<FakeCall>> calls(C);
        std::atomic<long> execs{0}, merged3{0};
        auto worker = [&](int i) {
            for (int r = 0; r < rounds; r++) {
                FakeCall fc{i, r, (r + (i == 2 ? 1 : 0)) & 1, per};
                CombineReq q;
                q.kind = fc.kind; q.n = fc.n; q.full = true; q.args = &fc;
                int rc_count = 0;
                const int rc = comb.call(i, q, [&](int first, int count, const CombineReq *const *reqs) -> int {
                    execs++;
                    if (count == 3) merged3++;
                    for (int k = 0; k < count; k++) {
                        FakeCall *f = static_cast<FakeCall *>(reqs[k]->args);
                        CHECK(f->member == first + k, "run member order");
                        CHECK(reqs[k]->kind == reqs[0]->kind, "kinds mixed in one run");
                        CHECK(f->served_by < 0, "a call served twice");
                        f->served_by = first; f->run_first = first; f->run_count = count;
                    }
                    std::this_thread::sleep_for(std::chrono::microseconds(100));
                    comb.near_end(first); // the run announces its end: its other members wake up and wait awake
                    std::this_thread::sleep_for(std::chrono::microseconds(100));
                    return 0;
                }, nullptr, &rc_count);
                CHECK(rc == 0 && fc.served_by >= 0 && fc.run_count == rc_count, "call %d/%d not served (rc %d)", i, r, rc);
                calls[i].push_back(fc);
            }
        };
        std::vector<std::thread> th;
        for (int i = 0; i < C; i++) th.emplace_back(worker, i);
        for (auto &t : th) t.join();
        // after the first few calls everything runs as runs of three
        long late3 = 0;
        for (int i = 0; i < C; i++)
            for (int r = rounds / 2; r < rounds; r++) late3 += calls[i][r].run_count == 3;
        CHECK(late3 >= (long)C * (rounds - rounds / 2) * 9 / 10, "opposite-phase callers did not fall into step: %ld of %d late calls in runs of 3",
              late3, C * (rounds - rounds / 2));
        long runs = 0, served = 0;
        comb.stats(&runs, &served);
        CHECK(served == (long)C * rounds && runs == execs.load(), "stats: %ld served, %ld runs, %ld execs", served, runs, execs.load());
    }
    // (2) ragged batches, unmergeable kinds, a throwing run, members that leave and join
    {
        Combiner comb(C, 20000, 5000, 40); // a pre-wake window SHORTER than the tails below: pre-woken members go back to sleep
        for (int i = 0; i < C; i++) comb.join();
        std::atomic<int> bad{0};
        auto worker = [&](int i) {
            for (int r = 0; r < rounds; r++) {
                FakeCall fc{i, r, (r % 7 == 3 && i == 1) ? -1 : 0, (r % 5 == 1 && i == 0) ? per - 1 : per};
                CombineReq q;
                q.kind = fc.kind; q.n = fc.n; q.full = fc.n == per; q.args = &fc;
                std::string what;
                const bool thrower = r % 11 == 6;
                const int rc = comb.call(i, q, [&](int first, int count, const CombineReq *const *reqs) -> int {
                    for (int k = 0; k < count; k++) {
                        FakeCall *f = static_cast<FakeCall *>(reqs[k]->args);
                        if (k + 1 < count && f->n != per) bad++;          // a short batch in the middle of a run
                        if (count > 1 && reqs[k]->kind < 0) bad++;        // an unmergeable request merged
                        f->served_by = first;
                    }
                    if (static_cast<FakeCall *>(reqs[0]->args)->seq % 3 != 0) comb.near_end(first); // some runs announce their end, twice even
                    if (static_cast<FakeCall *>(reqs[0]->args)->seq % 6 == 1) comb.near_end(first);
                    if (static_cast<FakeCall *>(reqs[0]->args)->seq % 4 == 2) std::this_thread::sleep_for(std::chrono::microseconds(120)); // longer than the window
                    if (static_cast<FakeCall *>(reqs[0]->args)->seq % 11 == 6) throw std::runtime_error("boom");
                    return 0;
                }, &what);
                if (rc == -2) CHECK(what == "boom", "exception text lost");
                else CHECK(rc == 0, "rc %d", rc);
                (void)thrower;
                CHECK(fc.served_by >= 0, "call not served");
                if (i == 2 && r % 50 == 49) { // member 2 leaves and comes back
                    comb.leave(2);
                    CHECK(comb.join() == 2, "rejoin");
                }
            }
        };
        std::vector<std::thread> th;
        for (int i = 0; i < C; i++) th.emplace_back(worker, i);
        for (auto &t : th) t.join();
        CHECK(bad.load() == 0, "%d malformed runs", bad.load());
    }
    // (3) a lone caller of a cohort whose other members are idle is not delayed
    {
        Combiner comb(C, 2000000, 1000);
        for (int i = 0; i < C; i++) comb.join();
        const auto t0 = std::chrono::steady_clock::now();
        for (int r = 0; r < 20; r++) {
            CombineReq q;
            q.kind = 0; q.n = per; q.full = true;
            CHECK(comb.call(1, q, [&](int first, int count, const CombineReq *const *) { return (first == 1 && count == 1) ? 0 : -1; }) == 0, "lone call");
        }
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        CHECK(ms < 1000.0, "20 lone calls took %.1f ms", ms);
    }
}

int main(int argc, char **argv)
{
    const char *mode = argc > 1 ? argv[1] : "full";
    if (!strcmp(mode, "pool")) {
        pool_hammer(argc > 2 ? atoi(argv[2]) : 20000);
        host_vs_oracle(2);
    } else if (!strcmp(mode, "lanes")) {
        lanes_hammer(argc > 2 ? atoi(argv[2]) : 3000);
    } else if (!strcmp(mode, "combine")) {
        combine_hammer(argc > 2 ? atoi(argv[2]) : 400);
    } else {
        primitives();
        for (int K = 2; K <= 4; K++) host_vs_oracle(K);
        pool_hammer(2000);
    }
    if (fails) {
        fprintf(stderr, "%d check(s) failed\n", fails);
        return 1;
    }
    printf("san_driver %s: ok\n", mode);
    return 0;
}
