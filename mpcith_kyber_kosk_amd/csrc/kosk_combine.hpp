// Call combiner of a cohort of library handles (KOSK_COMBINE = C).  No HIP in here: oracle/san_driver.cpp runs this file under
// ThreadSanitizer / AddressSanitizer with fake executors (tests/test_sanitizers.py).
//
// Why: at 46 proofs per launch the chip-filling kernels of the path (commitment hashes, table products, lincomb, image
// transposes) sit far below their saturated rates -- a launch of 1 058 waves on 1 024 SIMDs costs two rounds of the SIMDs, a
// product of 9 982 rows leaves 48 CUs idle.  A service that holds one handle per worker thread and sends
// 46-proof calls from each cannot make its calls larger, but the library can serve several of them with ONE pipeline run:
//
//   * the C handles of a cohort are VIEWS of one arena context (kosk_ctx.hpp): member i owns proofs [i B, (i + 1) B) of every
//     per-proof buffer, so members i .. j together are simply a batch of (j - i + 1) B proofs starting at member i's pointers;
//   * a member entering a mergeable call posts a request.  The first one to arrive opens a WINDOW and waits -- at most
//     wait_us -- for the members that can be expected to call too (those inside a call right now, and those that left one
//     less than idle_us ago).  When the window closes, neighbouring members with the same request kind form a RUN; the run's
//     first member executes the merged call on its own view (stream, events, host workers) for all of them while the others
//     sleep, then every member returns with its own results.  Requests of a minority kind are deferred by one window (once),
//     which is what brings callers that alternate two kinds of call in opposite phase into step;
//   * a member nobody joins runs alone, exactly as an unmerged handle would (plus the wait when other members are active).
//
// Callers that loop (the bench's slots, a service under load) fall into step after one iteration: members of a run return
// together, call again together, and the window closes as soon as the last of them has posted.
#pragma once
#include <atomic>
#include <chrono>
#include <cstdio>
#include <condition_variable>
#include <exception>
#include <functional>
#include <mutex>
#include <string>
#include <vector>

namespace kosk {

struct CombineReq {
    int kind = 0;      // requests of equal kind merge (>= 0); the executor knows what the kinds mean
    int n = 0;         // units of this member's call
    bool full = false; // the member's block is full (n == its capacity): only then may a neighbour's block follow it in a run
    void *args = nullptr;
};

class Combiner {
public:
    // exec(first, count, reqs): run the merged call of members first .. first + count - 1 (reqs[k] = member first + k's request)
    // on the calling thread -- member `first`'s thread; the return value is handed to every member of the run
    using Exec = std::function<int(int first, int count, const CombineReq *const *reqs)>;

    static constexpr int MAX_WIDTH = 16;
    Combiner(int width, int wait_us, int idle_us, int prewake_us = 0)
        : C_(width < 1 ? 1 : (width > MAX_WIDTH ? MAX_WIDTH : width)), wait_us_(wait_us), idle_us_(idle_us), prewake_us_(prewake_us), m_((size_t)C_) {}
    int width() const { return C_; }

    // a free member index (-1: the cohort is full); leave() frees it again
    int join()
    {
        std::lock_guard<std::mutex> lk(mu_);
        for (int i = 0; i < C_; i++)
            if (m_[i].st == ABSENT) {
                m_[i] = Member{};
                m_[i].st = IDLE;
                m_[i].last_exit = clock::now() - std::chrono::seconds(3600);
                present_++;
                return i;
            }
        return -1;
    }
    // returns the members left
    int leave(int i)
    {
        std::lock_guard<std::mutex> lk(mu_);
        if (i >= 0 && i < C_ && m_[i].st != ABSENT) { m_[i].st = ABSENT; present_--; }
        cv_.notify_all();
        return present_;
    }
    int members() const
    {
        std::lock_guard<std::mutex> lk(mu_);
        return present_;
    }

    // Member i's call.  Returns exec's value for the run the call ended up in (-2 if exec threw; *what then holds the text).
    // *run_count: members served by that run.
    int call(int i, const CombineReq &r, const Exec &exec, std::string *what = nullptr, int *run_count = nullptr)
    {
        std::unique_lock<std::mutex> lk(mu_);
        Member &me = m_[i];
        me.req = r;
        me.st = POSTED;
        me.deferred = false;
        me.prewake = false;
        __atomic_store_n(&me.done_flag, 0, __ATOMIC_RELAXED);
        me.posted_at = clock::now();
        if (leader_ < 0) leader_ = i; // nobody is collecting requests: this member opens a window
        cv_.notify_all();
        while (me.st == POSTED) {
            if (leader_ == i) lead(lk, i);
            else cv_.wait(lk);
        }
        const auto assigned_at = clock::now();
        me.t_window += std::chrono::duration<double>(assigned_at - me.posted_at).count();
        me.calls++;
        me.size_hist[me.run_count < 9 ? me.run_count : 8]++;
        if (me.leader == i) {
            const int cnt = me.run_count;
            const CombineReq *reqs[MAX_WIDTH]; // no allocation between "assigned" and "running": nothing here can throw
            for (int k = 0; k < cnt; k++) {
                reqs[k] = &m_[i + k].req;
                m_[i + k].st = RUNNING;
            }
            lk.unlock();
            int rc;
            std::string text;
            try {
                rc = exec(i, cnt, reqs);
            } catch (const std::exception &e) {
                rc = -2;
                try { text = e.what(); } catch (...) {}
            } catch (...) {
                rc = -2;
                try { text = "unknown exception"; } catch (...) {}
            }
            lk.lock();
            for (int k = 0; k < cnt; k++) {
                m_[i + k].rc = rc;
                try { m_[i + k].what = text; } catch (...) {}
                m_[i + k].st = DONE;
                __atomic_store_n(&m_[i + k].done_flag, 1, __ATOMIC_RELEASE); // a pre-woken member spins on this outside the lock
            }
            runs_++;
            served_ += cnt;
            cv_.notify_all();
        } else {
            for (;;) {
                cv_.wait(lk, [&] { return me.st == DONE || me.prewake; });
                if (me.st == DONE) break;
                // the run's executor has announced its end (near_end): stay awake for it -- a sleeping thread needs 30-50 us to get
                // back on a core, and the members' next calls can only merge once the LAST of them is back
                me.prewake = false;
                lk.unlock();
                const auto t0 = clock::now();
                while (!__atomic_load_n(&me.done_flag, __ATOMIC_ACQUIRE) && clock::now() - t0 < std::chrono::microseconds(prewake_us_)) cpu_relax();
                lk.lock();
                if (me.st == DONE) break; // (otherwise: the end took longer than announced; back to sleep)
            }
        }
        const int rc = me.rc;
        if (what && rc == -2) { try { *what = me.what; } catch (...) {} }
        if (run_count) *run_count = me.run_count;
        me.st = IDLE;
        me.last_exit = clock::now();
        me.t_run += std::chrono::duration<double>(me.last_exit - assigned_at).count();
        return rc;
    }

    // Called by the executor of member `leader`'s run when only a short tail of the run is left (the last kernel has been queued, or
    // the last host round begins): the run's sleeping members wake up now and spin until the run is over.  No-op with prewake_us 0.
    void near_end(int leader)
    {
        if (prewake_us_ <= 0) return;
        std::lock_guard<std::mutex> lk(mu_);
        if (leader < 0 || leader >= C_ || m_[leader].st != RUNNING) return;
        bool any = false;
        for (int k = 1; k < m_[leader].run_count && leader + k < C_; k++)
            if (m_[leader + k].st == RUNNING) { m_[leader + k].prewake = true; any = true; }
        if (any) cv_.notify_all();
    }

    // diagnostic (KOSK_COMBINE_TRACE=1 prints it when a member leaves): calls, seconds between posting and being assigned to a
    // run, seconds from there to the return, deferrals, windows this member closed by time-out, histogram of run sizes
    std::string trace(int i) const
    {
        std::lock_guard<std::mutex> lk(mu_);
        const Member &m = m_[i];
        char buf[256];
        snprintf(buf, sizeof buf, "member %d: calls %ld, window wait %.3f ms/call, run %.3f ms/call, deferred %ld, window timeouts %ld, run sizes 1:%ld 2:%ld 3:%ld 4:%ld 5+:%ld",
                 i, m.calls, m.calls ? m.t_window / m.calls * 1e3 : 0.0, m.calls ? m.t_run / m.calls * 1e3 : 0.0, m.n_deferred, m.n_timeouts,
                 m.size_hist[1], m.size_hist[2], m.size_hist[3], m.size_hist[4], m.size_hist[5] + m.size_hist[6] + m.size_hist[7] + m.size_hist[8]);
        return buf;
    }

    // runs executed and member calls served by them (served / runs = mean members per run)
    void stats(long *runs, long *served) const
    {
        std::lock_guard<std::mutex> lk(mu_);
        if (runs) *runs = runs_;
        if (served) *served = served_;
    }

private:
    using clock = std::chrono::steady_clock;
    enum State { ABSENT, IDLE, POSTED, ASSIGNED, RUNNING, DONE };
    struct Member {
        State st = ABSENT;
        CombineReq req;
        int leader = -1, run_count = 0, rc = 0;
        bool deferred = false, prewake = false;
        int done_flag = 0; // written under the lock, read by a pre-woken member outside it: __atomic accesses (keeps Member assignable)
        std::string what;
        clock::time_point last_exit, posted_at;
        double t_window = 0, t_run = 0;
        long calls = 0, n_deferred = 0, n_timeouts = 0, size_hist[9] = {0};
    };

    bool all_expected_posted(clock::time_point now) const
    {
        for (const Member &m : m_) {
            switch (m.st) {
            case ABSENT: case POSTED: break;
            case IDLE:
                if (now - m.last_exit < std::chrono::microseconds(idle_us_)) return false; // it has just left a call: it may be back
                break;
            default: return false; // inside a call: it can only post once that call is over
            }
        }
        return true;
    }
    // The window of member i (the one collecting requests right now): wait until every expected member has posted -- or until
    // i has waited wait_us since it posted -- then turn the posted requests into runs.  Members whose request kind is in the
    // minority are DEFERRED once: they stay posted while the majority's run executes, and the first of them leads the next
    // window.  (Two callers that alternate keygen / verify in opposite phase would otherwise never meet: the deferred one
    // skips a beat and they are in phase from then on.)
    void lead(std::unique_lock<std::mutex> &lk, int i)
    {
        const auto deadline = m_[i].posted_at + std::chrono::microseconds(wait_us_);
        bool timed_out = false;
        while (present_ > 1 && !all_expected_posted(clock::now())) {
            const auto now = clock::now();
            if (now >= deadline) { timed_out = true; m_[i].n_timeouts++; break; }
            // idle members stop being expected as time passes: poll, nobody notifies for that.  Under ThreadSanitizer the nap is
            // on the SYSTEM clock: pthread_cond_timedwait, which the sanitizer intercepts (a steady-clock wait is
            // pthread_cond_clockwait, which gcc 11's libtsan does not see: it then reports the mutex as locked twice);
            // the deadline itself is kept on the steady clock above
            const auto left = std::chrono::duration_cast<std::chrono::microseconds>(deadline - now);
            const auto nap = left < std::chrono::microseconds(50) ? left : std::chrono::microseconds(50);
#if defined(__SANITIZE_THREAD__)
            cv_.wait_until(lk, std::chrono::system_clock::now() + nap);
#else
            cv_.wait_for(lk, nap); // steady clock: a wall-clock step (NTP, a manual set) must not stretch a 50 us nap into the size of the step
#endif
        }
        form_runs(!timed_out);
        leader_ = -1;
        for (int j = 0; j < C_; j++)
            if (m_[j].st == POSTED) { leader_ = j; break; } // a deferred member takes over
        cv_.notify_all();
    }
    // neighbouring POSTED members with the same kind, every block but the last one full
    void form_runs(bool may_defer)
    {
        int keep = -1; // the request kind that runs now when several kinds are posted
        if (may_defer) {
            int votes[8] = {0};
            clock::time_point first[8];
            bool mixed = false;
            int seen = -1;
            for (const Member &m : m_)
                if (m.st == POSTED && m.req.kind >= 0 && m.req.kind < 8) {
                    if (!votes[m.req.kind] || m.posted_at < first[m.req.kind]) first[m.req.kind] = m.posted_at;
                    votes[m.req.kind]++;
                    if (seen >= 0 && seen != m.req.kind) mixed = true;
                    seen = m.req.kind;
                }
            if (mixed)
                for (int k = 0; k < 8; k++)
                    if (votes[k] && (keep < 0 || votes[k] > votes[keep] || (votes[k] == votes[keep] && first[k] < first[keep]))) keep = k;
        }
        for (int i = 0; i < C_;) {
            Member &a = m_[i];
            if (a.st != POSTED) { i++; continue; }
            if (keep >= 0 && a.req.kind >= 0 && a.req.kind < 8 && a.req.kind != keep && !a.deferred) {
                a.deferred = true; // stays POSTED
                a.n_deferred++;
                i++;
                continue;
            }
            int j = i;
            while (j + 1 < C_ && m_[j + 1].st == POSTED && m_[j].req.full && a.req.kind >= 0 && m_[j + 1].req.kind == a.req.kind) j++;
            for (int k = i; k <= j; k++) {
                m_[k].st = ASSIGNED;
                m_[k].leader = i;
                m_[k].run_count = j - i + 1;
            }
            i = j + 1;
        }
    }

    static void cpu_relax()
    {
#if defined(__x86_64__) || defined(__i386__)
        __builtin_ia32_pause();
#endif
    }
    const int C_, wait_us_, idle_us_, prewake_us_;
    mutable std::mutex mu_;
    std::condition_variable cv_;
    std::vector<Member> m_;
    int leader_ = -1; // the member collecting requests (its window is open), -1: none
    int present_ = 0;
    long runs_ = 0, served_ = 0;
};

} // namespace kosk
