// Persistent lane threads of a library handle (KOSK_STREAMS = S sub-contexts: S - 1 lane threads + the caller's thread) and the
// dealing of a batch call's chunks to them.  No HIP in here: oracle/san_driver.cpp runs this file under ThreadSanitizer /
// AddressSanitizer with fake sub-contexts (tests/test_sanitizers.py).
//
// Rules: a batch call never creates a thread (the lanes are created with the handle; a failure there is an ordinary error of
// kosk_create); a job's exception ends as rc -2 + text, never as std::terminate; LaneSet::run returns only when every lane
// is idle again, whatever was thrown where -- the jobs reference the caller's stack.
#pragma once
#include <condition_variable>
#include <exception>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace kosk {

class Lane {
public:
    Lane() : th_([this] { loop(); }) {} // std::thread may throw std::system_error: the creator reports it
    Lane(const Lane &) = delete;
    Lane &operator=(const Lane &) = delete;
    ~Lane()
    {
        {
            std::lock_guard<std::mutex> lk(mu_);
            stop_ = true;
        }
        cv_.notify_all();
        th_.join();
    }
    void start(std::function<int()> job)
    {
        {
            std::lock_guard<std::mutex> lk(mu_);
            job_ = std::move(job);
            what_.clear();
            state_ = RUNNING;
        }
        cv_.notify_all();
    }
    int wait(std::string *what)
    {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return state_ != RUNNING; });
        state_ = IDLE;
        if (what && rc_ == -2) *what = what_;
        return rc_;
    }

private:
    enum State { IDLE, RUNNING, DONE };
    void loop()
    {
        std::unique_lock<std::mutex> lk(mu_);
        for (;;) {
            cv_.wait(lk, [&] { return stop_ || state_ == RUNNING; });
            if (stop_) return;
            std::function<int()> job = std::move(job_);
            job_ = nullptr;
            lk.unlock();
            int rc;
            std::string what;
            try {
                rc = job();
            } catch (const std::exception &e) {
                rc = -2;
                try { what = e.what(); } catch (...) {}
            } catch (...) {
                rc = -2;
                try { what = "unknown exception"; } catch (...) {}
            }
            job = nullptr;
            lk.lock();
            rc_ = rc;
            what_.swap(what);
            state_ = DONE;
            cv_.notify_all();
        }
    }
    std::mutex mu_;
    std::condition_variable cv_;
    std::function<int()> job_;
    std::string what_;
    State state_ = IDLE;
    int rc_ = 0;
    bool stop_ = false;
    std::thread th_; // last member: the thread starts when everything above exists
};

struct LaneSet {
    std::vector<std::unique_ptr<Lane>> lanes; // lanes[i - 1] runs job i; job 0 runs on the calling thread

    void create(int extra) // may throw (std::system_error, std::bad_alloc): handle creation fails, nothing else
    {
        for (int i = 0; i < extra; i++) lanes.emplace_back(new Lane());
    }
    int size() const { return (int)lanes.size() + 1; }

    // jobs.size() == size(); an empty job is skipped.  rc[i] = the job's return value, -2 + what[i] if it threw.
    void run(std::vector<std::function<int()>> &jobs, std::vector<int> &rc, std::vector<std::string> &what) noexcept
    {
        const int S = size();
        std::vector<char> started;
        try {
            rc.assign(S, 0);
            what.assign(S, std::string());
            started.assign(S, 0);
        } catch (...) {
            // no memory for the bookkeeping: nothing has been started
            for (auto &r : rc) r = -2;
            return;
        }
        for (int i = 1; i < S; i++)
            if (jobs[i]) {
                try {
                    lanes[i - 1]->start(jobs[i]); // copies the job (may throw): the original stays valid for the report below
                    started[i] = 1;
                } catch (...) {
                    rc[i] = -2;
                    try { what[i] = "could not hand the job to its lane thread"; } catch (...) {}
                }
            }
        if (jobs[0]) {
            try {
                rc[0] = jobs[0]();
            } catch (const std::exception &e) {
                rc[0] = -2;
                try { what[0] = e.what(); } catch (...) {}
            } catch (...) {
                rc[0] = -2;
                try { what[0] = "unknown exception"; } catch (...) {}
            }
        }
        for (int i = 1; i < S; i++)
            if (started[i]) rc[i] = lanes[i - 1]->wait(&what[i]);
    }
};

// jobs for an n-unit call on S lanes whose sub-context takes `per` units at a time: chunk j = [j * per, min(n, (j + 1) * per))
// goes to lane j % S; a lane stops at its first failing chunk.  fn(lane, first, count) -> rc
template <typename F>
void deal_chunks(int S, int per, int n, F &fn, std::vector<std::function<int()>> &jobs)
{
    const int nchunks = (n + per - 1) / per;
    jobs.assign(S, std::function<int()>());
    for (int i = 0; i < S && i < nchunks; i++)
        jobs[i] = [&fn, i, S, per, nchunks, n] {
            for (int j = i; j < nchunks; j += S) {
                const int first = j * per, count = (n - first) < per ? (n - first) : per;
                if (const int rc = fn(i, first, count)) return rc;
            }
            return 0;
        };
}
// jobs for an n-unit call (n <= capacity) as S contiguous sub-batches: sub-batch i = base + (i < rem) units
template <typename F>
void deal_split(int S, int n, F &fn, std::vector<std::function<int()>> &jobs)
{
    jobs.assign(S, std::function<int()>());
    const int base = n / S, rem = n % S;
    for (int i = 0; i < S; i++) {
        const int count = base + (i < rem ? 1 : 0), first = i * base + (i < rem ? i : rem);
        if (count > 0) jobs[i] = [&fn, i, first, count] { return fn(i, first, count); };
    }
}

} // namespace kosk
