// extern "C" boundary (include/kosk_mi355x.h) over kosk::Ctx.
//
// Containment rules of this file (the reference's API is void / bool and never kills its caller except on RNG failure,
// kosk.hpp:18-24, kyber/randombytes.c:49-52):
//   * no C++ exception leaves an extern "C" function: every entry point runs inside guard(), which turns std::exception /
//     anything else into rc -1 + kosk_last_error() text;
//   * no thread is created inside a batch call: the S - 1 lane threads of a handle (KOSK_STREAMS = S) and every
//     sub-context's host workers are created by kosk_create(), whose failure is an ordinary -1;
//   * the library never page-locks caller memory (round 5: KOSK_REGISTER=2, which did for the duration of a multi-chunk call, is
//     gone -- both process aborts on record happened inside calls that had just done so and neither was ever explained);
//     buffers the caller page-locked itself (kosk_host_alloc, hipHostMalloc, hipHostRegister) are copied to / from directly,
//     everything else goes through the library's own pinned staging buffers.
#include "../../include/kosk_mi355x.h"

#include <unistd.h>

#include <algorithm>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <exception>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "kosk_combine.hpp"
#include "kosk_ctx.hpp"
#include "kosk_lanes.hpp"

using namespace kosk;

struct kosk_ctx;
// A cohort (KOSK_COMBINE = C): up to C handles of equal (device, kyber_k, max_batch) are views of ONE arena context, and their
// resident calls are served by merged pipeline runs (kosk_combine.hpp)
struct Cohort {
    int device = 0, k = 0, per = 0;
    CtxOpts opts; // what the members' options asked of the shared context: a handle only joins a cohort created with the same
    Ctx *arena = nullptr;
    std::unique_ptr<Combiner> comb;
    std::vector<kosk_ctx *> member;
};
static std::mutex g_cohort_mu; // creation / destruction of cohorts and their members
static std::vector<Cohort *> g_cohorts;
enum CombineKind { CK_ALONE = -1, CK_KEYGEN = 0, CK_VERIFY_PK_GIVEN = 1, CK_VERIFY_PK_RESIDENT = 2 };

// A handle owns S sub-contexts (own stream + HBM workspace each).  A batch call splits its proofs into S
// contiguous sub-batches that run concurrently on S host threads: while one sub-batch sits in a host
// Fiat-Shamir round trip the GPU works on the other, and kernels of different streams share the CUs.
struct kosk_ctx {
    std::vector<Ctx *> sub;
    LaneSet lanes; // lane i runs sub[i]; lane 0 is the caller's thread (kosk_lanes.hpp)
    Ctx *c = nullptr; // sub[0]: kernel-level entry points, error text, sizes
    int max_batch = 0;
    std::string err;
    std::vector<uint32_t> masks; // fail masks of the last verify call, in the caller's proof order
    int masks_n = 0;             // proofs of the last verify call (kosk_verify_fail_masks serves exactly these)
    Cohort *cohort = nullptr;    // KOSK_COMBINE: this handle is member `member_i` of a cohort, sub[0] a view of its arena
    int member_i = -1;
    long merged_calls = 0, merged_members = 0; // resident calls of this handle served by a run, and the members those runs served
    bool hooks_unmerged = false; // kosk_options::hooks_unmerged: with a round hook set, this handle's resident calls run on their own

    ~kosk_ctx()
    {
        lanes.lanes.clear(); // join the lane threads before their sub-contexts go
        for (Ctx *x : sub) delete x;
        if (cohort) {
            std::lock_guard<std::mutex> lk(g_cohort_mu);
            cohort->member[member_i] = nullptr;
            if (const char *e = getenv("KOSK_COMBINE_TRACE"))
                if (atoi(e)) fprintf(stderr, "[kosk combine] cohort %p %s\n", (void *)cohort, cohort->comb->trace(member_i).c_str());
            if (cohort->comb->leave(member_i) == 0) { // the last member frees the arena
                delete cohort->arena;
                g_cohorts.erase(std::remove(g_cohorts.begin(), g_cohorts.end(), cohort), g_cohorts.end());
                delete cohort;
            }
        }
    }
    // every entry point starts from a clean error state (kosk_last_error never reports a stale message)
    void clear_err()
    {
        err.clear();
        for (Ctx *x : sub) x->err.clear();
    }
    int run_lanes(std::vector<std::function<int()>> &jobs)
    {
        std::vector<int> rc;
        std::vector<std::string> what;
        lanes.run(jobs, rc, what);
        if (rc.size() < sub.size() || what.size() < sub.size()) { // no memory even for the bookkeeping: nothing was started
            err = "out of memory while dealing a batch call to the handle's lanes";
            c->err = err;
            return -1;
        }
        for (int i = 0; i < (int)sub.size(); i++)
            if (rc[i]) {
                if (rc[i] == -2) sub[i]->err = "exception in a batch job: " + what[i];
                err = sub[i]->err;
                c->err = err;
                return -1;
            }
        return 0;
    }
    // an n-proof call (n <= max_batch) as S contiguous sub-batches: fn(sub-context, first, count)
    template <typename F>
    int run(int n, F &&fn)
    {
        clear_err();
        auto on_lane = [&](int i, int first, int count) { return fn(*sub[i], first, count); };
        std::vector<std::function<int()>> jobs;
        deal_split((int)sub.size(), n, on_lane, jobs);
        return run_lanes(jobs);
    }
};

// Every extern "C" entry point's body runs in here: nothing thrown below (std::bad_alloc, std::system_error, an exception
// out of a host worker job) reaches the caller's frames.
template <typename F>
static int guard(kosk_ctx *ctx, const char *fn, F &&body) noexcept
{
    try {
        // a stale per-thread HIP error (an earlier failed call of ours, or of the host application) must not be taken for
        // the result of this call's first kernel launch: the launchers report hipGetLastError().  Documented in the header:
        // every entry point RESETS the calling thread's HIP last-error state, so the application has to look at the outcome of
        // its own launches before it calls in here
        (void)hipGetLastError();
        return body();
    } catch (const std::exception &e) {
        try {
            if (ctx) { ctx->err = std::string(fn) + ": " + e.what(); if (ctx->c) ctx->c->err = ctx->err; }
        } catch (...) {}
    } catch (...) {
        try {
            if (ctx) { ctx->err = std::string(fn) + ": unknown exception"; if (ctx->c) ctx->c->err = ctx->err; }
        } catch (...) {}
    }
    return -1;
}
template <typename F>
static int guard(const kosk_ctx *, const char *, F &&body) noexcept // read-only entry points leave the error string alone
{
    try {
        return body();
    } catch (...) {
    }
    return -1;
}
#define GUARD(ctx) return guard(ctx, __func__, [&]() -> int {
#define GUARD_END });

// argument validation failed: the call has done nothing; kosk_last_error(ctx) says which entry point refused
static int bad_args(const kosk_ctx *, const char *) { return -1; } // read-only entry points leave the error string alone
static int bad_args(kosk_ctx *ctx, const char *fn)
{
    if (ctx) {
        ctx->clear_err();
        ctx->err = std::string(fn) + ": invalid argument (null pointer, batch size out of range, or unsupported for this context)";
        ctx->c->err = ctx->err;
    }
    return -1;
}

// Batches larger than a sub-context: the chunks (of a sub-context's capacity each) are dealt round-robin to the S
// sub-contexts, which work through theirs concurrently on the handle's lane threads.  With KOSK_STREAMS >= 2 one chunk's
// PCIe transfers (tapes in, 0.68 MB of proof image out per proof; proofs and keys in for the verifier) and host hashing run
// under another chunk's kernels: the streaming write-back of SURVEY.md 8 f4 for a single caller thread.
template <typename F>
static int run_chunks(kosk_ctx *h, int n, F &&fn)
{
    h->clear_err();
    auto on_lane = [&](int i, int first, int count) { return fn(*h->sub[i], first, count); };
    std::vector<std::function<int()>> jobs;
    deal_chunks((int)h->sub.size(), h->sub[0]->own_batch, n, on_lane, jobs); // own_batch: a cohort member's max_batch spans its neighbours' blocks
    return h->run_lanes(jobs);
}

// [p, p + bytes) is page-locked host memory already (kosk_host_alloc, hipHostMalloc, hipHostRegister by the caller): the
// copies can go straight to / from it, no staging and no per-call locking
static bool span_is_pinned(const void *p, size_t bytes)
{
    if (!p || !bytes) return false;
    hipPointerAttribute_t a0{}, a1{};
    if (hipPointerGetAttributes(&a0, p) != hipSuccess ||
        hipPointerGetAttributes(&a1, static_cast<const uint8_t *>(p) + bytes - 1) != hipSuccess) {
        (void)hipGetLastError(); // plain pageable memory: not an error
        return false;
    }
    if (a0.type != hipMemoryTypeHost || a1.type != hipMemoryTypeHost) return false;
    // both ends page-locked is not enough: they must belong to ONE mapping (two registrations with a pageable gap between them,
    // or a neighbour's registration that ends inside the buffer, give unrelated device addresses)
    if (!a0.devicePointer || !a1.devicePointer) return false;
    return static_cast<const uint8_t *>(a1.devicePointer) - static_cast<const uint8_t *>(a0.devicePointer) == (ptrdiff_t)(bytes - 1);
}

// draw n proofs' worth of randomness through the (stateful) callback / OS entropy, sequentially in proof order and in the
// reference's call order and lengths (kosk.cpp:12, mlwe_prover.cpp:9, ss.cpp:5)
static void draw_tapes(const kosk_ctx *ctx, int n, std::vector<uint8_t> &drawn)
{
    const Params &P = ctx->c->P;
    drawn.resize((size_t)n * P.tape_bytes);
    const Ctx &c0 = *ctx->c;
    uint8_t *tp = drawn.data();
    auto draw = [&](size_t len) { if (c0.rb) c0.rb(c0.rb_user, tp, len); else os_randombytes(tp, len); tp += len; };
    for (int b = 0; b < n; b++) {
        draw(64);
        for (int i = 0; i < P.M; i++) draw(32);
        for (int i = 0; i < P.nfresh; i++) draw(302);
    }
}

static thread_local std::string g_create_err; // error text of the last failed kosk_create on this thread

#define HIPCHK_C(x) KOSK_HIPCHK(x)

extern "C" {

size_t kosk_pk_bytes(int k) { Params p; return make_params(k, p) ? p.pk_bytes : 0; }
size_t kosk_sk_bytes(int k) { Params p; return make_params(k, p) ? p.sk_bytes : 0; }
size_t kosk_proof_bytes(int k) { Params p; return make_params(k, p) ? p.proof_bytes : 0; }
size_t kosk_tape_bytes(int k) { Params p; return make_params(k, p) ? p.tape_bytes : 0; }
int kosk_proof_field(int k, int idx, size_t *offset, size_t *size)
{
    Params p;
    if (!make_params(k, p) || idx < 0 || idx >= NFIELDS) return -1;
    if (offset) *offset = p.off[idx];
    if (size) *size = p.size[idx];
    return 0;
}

void kosk_options_init(kosk_options *opt)
{
    if (!opt) return;
    memset(opt, 0, sizeof *opt);
    opt->size = (uint32_t)sizeof *opt;
    opt->combine_wait_us = opt->combine_idle_us = opt->combine_prewake_us = -1;
    opt->strict_encoding = opt->fs_mode = opt->blocking_sync = opt->hooks_unmerged = -1;
}
int kosk_create(kosk_ctx **ctx, int device, int kyber_k, int max_batch) { return kosk_create_ex(ctx, device, kyber_k, max_batch, nullptr); }

int kosk_create_ex(kosk_ctx **ctx, int device, int kyber_k, int max_batch, const kosk_options *user_opt)
{
    if (!ctx) return -1;
    *ctx = nullptr;
    kosk_ctx *h = nullptr;
    try {
        g_create_err.clear();
        kosk_options o;
        kosk_options_init(&o);
        if (user_opt) { // a caller compiled against a shorter (older) struct leaves the tail at "not given"
            if (user_opt->size < 8 || user_opt->size > 4096) { g_create_err = "kosk_create_ex: options.size is not set (call kosk_options_init first)"; return -1; }
            memcpy(&o, user_opt, std::min<size_t>(user_opt->size, sizeof o));
            o.size = (uint32_t)sizeof o;
        }
        CtxOpts co_;
        co_.host_threads = o.host_threads > 0 ? o.host_threads : 0;
        co_.blocking_sync = o.blocking_sync;
        co_.strict_encoding = o.strict_encoding;
        co_.fs_device = o.fs_mode < 0 ? -1 : (o.fs_mode == KOSK_FS_DEVICE ? 1 : 0);
        if (const char *e = getenv("AMD_DIRECT_DISPATCH"))
            if (atoi(e) == 0 && e[0] != '\0') {
                // measured on ROCm 7.2 (tools/stress.py): with direct dispatch off, hipStreamSynchronize returned before
                // device-to-host copies into pinned memory had landed; the host then hashed stale digests
                g_create_err = "AMD_DIRECT_DISPATCH=0 is not supported: stream synchronisation does not cover D2H copies in that mode";
                return -1;
            }
        int S = 1; // streams: sub-batches in flight per handle (1 measured best at 46 proofs: kernels sit on latency floors)
        if (o.streams > 0) S = o.streams;
        if (S > max_batch) S = max_batch > 0 ? max_batch : 1;
        if (S > 8) S = 8;
        h = new kosk_ctx();
        h->max_batch = max_batch;
        h->hooks_unmerged = o.hooks_unmerged > 0;
        int W = 1; // combine: handles per cohort (needs streams = 1)
        if (o.combine > 0) W = o.combine > Combiner::MAX_WIDTH ? Combiner::MAX_WIDTH : o.combine;
        if (W > 1 && S == 1 && max_batch >= 1) {
            std::lock_guard<std::mutex> lk(g_cohort_mu);
            Cohort *co = nullptr;
            int idx = -1;
            for (Cohort *x : g_cohorts)
                if (x->device == device && x->k == kyber_k && x->per == max_batch && x->comb->width() == W && x->opts.host_threads == co_.host_threads &&
                    x->opts.blocking_sync == co_.blocking_sync && x->opts.strict_encoding == co_.strict_encoding && x->opts.fs_device == co_.fs_device &&
                    (idx = x->comb->join()) >= 0) { co = x; break; }
            if (!co) {
                std::unique_ptr<Cohort> fresh(new Cohort());
                fresh->device = device; fresh->k = kyber_k; fresh->per = max_batch; fresh->opts = co_;
                int wait_us = 5000, idle_us = 1000;
                if (o.combine_wait_us >= 0) wait_us = o.combine_wait_us;
                if (o.combine_idle_us >= 0) idle_us = o.combine_idle_us;
                int prewake_us = 400; // how long a member whose run has announced its end may spin for it (0: members sleep to the end)
                if (o.combine_prewake_us >= 0) prewake_us = o.combine_prewake_us;
                fresh->comb.reset(new Combiner(W, wait_us, idle_us, prewake_us));
                fresh->member.assign((size_t)W, nullptr);
                if ((long)W * max_batch > 1 << 20) { g_create_err = "combine x max_batch too large"; delete h; return -1; }
                if (ctx_create(&fresh->arena, device, kyber_k, W * max_batch, g_create_err, 1, co_)) { delete h; return -1; }
                if (ensure_verify_workspace(*fresh->arena)) { // views share it: allocated with the arena, not on first use
                    g_create_err = fresh->arena->err;
                    delete fresh->arena;
                    delete h;
                    return -1;
                }
                idx = fresh->comb->join();
                co = fresh.release();
                g_cohorts.push_back(co);
            }
            Ctx *v = nullptr;
            // Host workers: a full run is led by member 0 and hashes W callers' tables, so member 0 gets W callers' worth; the
            // other members only ever lead partial runs (start-up, stragglers) and make do with one caller's worth -- a
            // process with many cohorts (one per rank on an 8-GPU node) stays far below any thread limit
            if (ctx_make_view(*co->arena, idx * max_batch, max_batch, co->arena->base_threads * (idx == 0 ? W : 1), &v, g_create_err)) {
                if (co->comb->leave(idx) == 0) {
                    delete co->arena;
                    g_cohorts.erase(std::remove(g_cohorts.begin(), g_cohorts.end(), co), g_cohorts.end());
                    delete co;
                }
                delete h;
                return -1;
            }
            h->sub.push_back(v);
            h->c = v;
            h->cohort = co;
            h->member_i = idx;
            co->member[idx] = h;
            *ctx = h;
            return 0;
        }
        const int per = (max_batch + S - 1) / S;
        for (int i = 0; i < S; i++) {
            Ctx *c = nullptr;
            if (ctx_create(&c, device, kyber_k, per, g_create_err, S, co_)) {
                delete h;
                return -1;
            }
            h->sub.push_back(c);
        }
        h->c = h->sub[0];
        // the lane threads live as long as the handle: a batch call never creates a thread
        h->lanes.create(S - 1);
        *ctx = h;
        return 0;
    } catch (const std::exception &e) {
        try { g_create_err = std::string("kosk_create: ") + e.what(); } catch (...) {}
    } catch (...) {
        try { g_create_err = "kosk_create: unknown exception"; } catch (...) {}
    }
    try { delete h; } catch (...) {}
    return -1;
}
void kosk_destroy(kosk_ctx *ctx)
{
    if (!ctx) return;
    try { delete ctx; } catch (...) {}
}
const char *kosk_last_error(const kosk_ctx *ctx)
{
    if (!ctx) return g_create_err.c_str();
    return ctx->err.empty() ? ctx->c->err.c_str() : ctx->err.c_str();
}
int kosk_set_randombytes(kosk_ctx *ctx, kosk_randombytes_fn fn, void *user)
{
    if (!ctx) return -1;
    for (Ctx *c : ctx->sub) { c->rb = fn; c->rb_user = user; }
    return 0;
}

int kosk_stage_prover_inputs(kosk_ctx *ctx, int n, const uint8_t *tapes, size_t tape_stride, uint8_t *pk, uint8_t *sk)
{
    if (!ctx || n < 1 || n > ctx->max_batch) return bad_args(ctx, __func__);
    GUARD(ctx)
    const Params &P = ctx->c->P;
    std::vector<uint8_t> drawn;
    if (!tapes) { // the randombytes callback is stateful: draw every tape sequentially, then stage in parallel
        draw_tapes(ctx, n, drawn);
        tapes = drawn.data();
        tape_stride = P.tape_bytes;
    }
    return ctx->run(n, [&](Ctx &c, int first, int count) {
        return stage_prover_inputs(c, count, tapes + (size_t)first * tape_stride, tape_stride,
                                   pk + (size_t)first * P.pk_bytes, sk + (size_t)first * P.sk_bytes);
    });
    GUARD_END
}
int kosk_prove_resident(kosk_ctx *ctx, int n)
{
    if (!ctx || n < 1 || n > ctx->max_batch) return bad_args(ctx, __func__);
    GUARD(ctx)
    return ctx->run(n, [&](Ctx &c, int, int count) { return prove_resident(c, count); });
    GUARD_END
}
int kosk_fetch_proofs(kosk_ctx *ctx, int n, uint8_t *pi)
{
    if (!ctx || n < 1 || n > ctx->max_batch || !pi) return bad_args(ctx, __func__);
    GUARD(ctx)
    const Params &P = ctx->c->P;
    return ctx->run(n, [&](Ctx &c, int first, int count) { return fetch_proofs(c, count, pi + (size_t)first * P.proof_bytes); });
    GUARD_END
}
int kosk_stage_verifier_inputs(kosk_ctx *ctx, int n, const uint8_t *pi, const uint8_t *pk)
{
    if (!ctx || n < 1 || n > ctx->max_batch || !pi || !pk) return bad_args(ctx, __func__);
    GUARD(ctx)
    const Params &P = ctx->c->P;
    return ctx->run(n, [&](Ctx &c, int first, int count) {
        return stage_verifier_inputs(c, count, pi + (size_t)first * P.proof_bytes, pk + (size_t)first * P.pk_bytes);
    });
    GUARD_END
}

// verify `count` resident proofs of sub-context c and file their fail masks at the caller's proof index `first`
static int verify_into(kosk_ctx *ctx, Ctx &c, int first, int count, uint8_t *ok, int pk_mode, const uint8_t *pk)
{
    if (verify_resident(c, count, ok + first, pk_mode, pk)) return -1;
    memcpy(ctx->masks.data() + first, c.h_fail, sizeof(uint32_t) * (size_t)count);
    return 0;
}
// every verify entry point: the masks of an earlier call are gone, whatever happens next
static void reset_masks(kosk_ctx *ctx, int n)
{
    ctx->masks_n = 0;
    ctx->masks.assign((size_t)n, 0);
}

int kosk_verify_resident(kosk_ctx *ctx, int n, uint8_t *ok)
{
    if (!ctx || n < 1 || n > ctx->max_batch || !ok) return bad_args(ctx, __func__);
    GUARD(ctx)
    reset_masks(ctx, n);
    if (ctx->run(n, [&](Ctx &c, int first, int count) { return verify_into(ctx, c, first, count, ok, 0, nullptr); })) return -1;
    ctx->masks_n = n;
    return 0;
    GUARD_END
}

// ---- merged resident calls of a cohort (KOSK_COMBINE) ----
struct KeygenCall { const uint8_t *tapes; size_t tape_stride; uint8_t *pk, *sk; };
struct VerifyCall { const uint8_t *pk; uint8_t *ok; };

// what every member of a finished run takes over from the run's leader
static void run_epilogue(Cohort &co, int first, int count, int rc, const Ctx &lead)
{
    for (int k = 0; k < count; k++) {
        kosk_ctx *m = co.member[first + k];
        m->merged_calls++;
        m->merged_members += count;
        if (k) memcpy(m->c->phase_sec, lead.phase_sec, sizeof(lead.phase_sec));
        if (rc) { m->err = lead.err; m->c->err = lead.err; }
    }
}

// Scope of a merged run on its leader's view: while it lives, prove_resident / verify_resident tell the combiner when only their tail
// is left, and the run's other callers wake up for the return instead of sleeping through it (kosk_combine.hpp: near_end).
struct NearEnd {
    Ctx &c;
    NearEnd(Ctx &c_, Cohort &co, int first, int count) : c(c_)
    {
        if (count > 1) {
            Combiner *comb = co.comb.get();
            try { c.near_end_hook = [comb, first] { comb->near_end(first); }; } catch (...) { c.near_end_hook = nullptr; }
        }
    }
    ~NearEnd() { c.near_end_hook = nullptr; }
    NearEnd(const NearEnd &) = delete;
    NearEnd &operator=(const NearEnd &) = delete;
};

// Scope of a run on its leader's view: the view may run `total` proofs (its own block and the blocks of the run's other members
// behind it) with the host workers of `count` callers; both go back to one caller's worth however the run ends.
struct RunScope {
    Ctx &c;
    RunScope(Ctx &c_, int count, int total) : c(c_)
    {
        c.call_cap = std::min(std::max(total, c.own_batch), c.max_batch);
        c.nthreads = std::min(c.base_threads * count, c.reserved_threads); // a call never creates threads
    }
    ~RunScope() { c.call_cap = c.own_batch; c.nthreads = c.base_threads; }
    RunScope(const RunScope &) = delete;
    RunScope &operator=(const RunScope &) = delete;
};

static int combined_keygen(kosk_ctx *h, int n, const KeygenCall &call)
{
    Cohort &co = *h->cohort;
    h->clear_err();
    // a caller's bad arguments fail THAT caller, here, not the run it would have joined (and every innocent member of it)
    if (call.tapes && call.tape_stride < h->c->P.tape_bytes) {
        h->err = "tape_stride is smaller than one proof's tape (kosk_tape_bytes)";
        h->c->err = h->err;
        return -1;
    }
    CombineReq r;
    r.kind = call.tapes ? CK_KEYGEN : CK_ALONE; // the stateful randombytes callback is per handle
    if (h->hooks_unmerged && h->c->round_hook) r.kind = CK_ALONE; // the hook must fire on this caller's own thread (kosk_options::hooks_unmerged)
    r.n = n;
    r.full = n == co.per;
    r.args = const_cast<KeygenCall *>(&call);
    std::string what;
    const int rc = co.comb->call(h->member_i, r, [&co](int first, int count, const CombineReq *const *reqs) -> int {
        Ctx &c = *co.member[first]->c; // the run leader's view: this thread is its caller
        std::vector<KeygenIn> segs((size_t)count);
        int total = 0;
        for (int k = 0; k < count; k++) {
            const KeygenCall *a = static_cast<const KeygenCall *>(reqs[k]->args);
            const Ctx &mv = *co.member[first + k]->c; // round hooks are per handle: every member's fires with its own block of the tables
            segs[k] = KeygenIn{a->tapes, a->tape_stride, a->pk, a->sk, reqs[k]->n, k + 1 < count ? &segs[k + 1] : nullptr, mv.round_hook, mv.round_user};
            total += reqs[k]->n;
        }
        NearEnd ne(c, co, first, count);
        int rc;
        {
            RunScope scope(c, count, total);
            rc = prove_resident(c, total, false, &segs[0]);
        }
        for (int k = 0; k < count; k++) {
            Ctx &v = *co.member[first + k]->c;
            v.resident_pk_n = rc ? 0 : reqs[k]->n;
            // a merged run either copied every member's tapes into its own block of the tape buffer or read the callers' device
            // buffers in place: either way a later kosk_prove_resident on this member must not assume anything
            if (count > 1) v.tape_cur = nullptr;
        }
        run_epilogue(co, first, count, rc, c);
        return rc;
    }, &what);
    if (rc == -2) { h->err = "exception in a merged call: " + what; h->c->err = h->err; return -1; }
    return rc;
}

static int combined_verify(kosk_ctx *h, int n, const VerifyCall &call)
{
    Cohort &co = *h->cohort;
    h->clear_err();
    h->masks_n = 0;
    if (!call.pk && h->c->resident_pk_n < n) {
        h->err = "no resident public keys for this batch: pk == NULL needs a key generation (or a verifier staging call) of at least n proofs on this context";
        h->c->err = h->err;
        return -1;
    }
    CombineReq r;
    r.kind = call.pk ? CK_VERIFY_PK_GIVEN : CK_VERIFY_PK_RESIDENT;
    if (h->hooks_unmerged && h->c->round_hook) r.kind = CK_ALONE;
    r.n = n;
    r.full = n == co.per;
    r.args = const_cast<VerifyCall *>(&call);
    std::string what;
    const int rc = co.comb->call(h->member_i, r, [&co](int first, int count, const CombineReq *const *reqs) -> int {
        Ctx &c = *co.member[first]->c;
        std::vector<VerifySeg> segs((size_t)count);
        int total = 0;
        bool given = false;
        for (int k = 0; k < count; k++) {
            const VerifyCall *a = static_cast<const VerifyCall *>(reqs[k]->args);
            const Ctx &mv = *co.member[first + k]->c;
            segs[k] = VerifySeg{reqs[k]->n, a->pk, a->ok, k + 1 < count ? &segs[k + 1] : nullptr, mv.round_hook, mv.round_user};
            total += reqs[k]->n;
            given = a->pk != nullptr;
        }
        const int keep = c.resident_pk_n;
        if (!given) c.resident_pk_n = total; // every member checked its own keys before it posted
        NearEnd ne(c, co, first, count);
        int rc;
        {
            struct KeepPk { Ctx &c; int keep; bool on; ~KeepPk() { if (on) c.resident_pk_n = keep; } } keep_pk{c, keep, !given};
            RunScope scope(c, count, total);
            rc = verify_resident(c, total, nullptr, given ? 1 : 2, nullptr, &segs[0]);
        }
        int off = 0;
        for (int k = 0; k < count; k++) {
            kosk_ctx *m = co.member[first + k];
            m->masks.assign((size_t)reqs[k]->n, 0);
            if (!rc) {
                memcpy(m->masks.data(), c.h_fail + off, sizeof(uint32_t) * (size_t)reqs[k]->n);
                m->masks_n = reqs[k]->n;
                if (given) m->c->resident_pk_n = reqs[k]->n;
            }
            off += reqs[k]->n;
        }
        run_epilogue(co, first, count, rc, c);
        return rc;
    }, &what);
    if (rc == -2) { h->err = "exception in a merged call: " + what; h->c->err = h->err; return -1; }
    return rc;
}

int kosk_verifiable_keygen_resident(kosk_ctx *ctx, int n, const uint8_t *tapes, size_t tape_stride, uint8_t *pk, uint8_t *sk)
{
    if (!ctx || n < 1 || n > ctx->max_batch || !pk || !sk) return bad_args(ctx, __func__);
    GUARD(ctx)
    if (ctx->cohort) return combined_keygen(ctx, n, KeygenCall{tapes, tape_stride, pk, sk});
    const Params &P = ctx->c->P;
    std::vector<uint8_t> drawn;
    if (!tapes && ctx->sub.size() > 1) { // stateful callback: draw sequentially in proof order, then prove the sub-batches in parallel
        draw_tapes(ctx, n, drawn);
        tapes = drawn.data();
        tape_stride = P.tape_bytes;
    }
    return ctx->run(n, [&](Ctx &c, int first, int count) {
        const KeygenIn kg{tapes ? tapes + (size_t)first * tape_stride : nullptr, tape_stride, pk + (size_t)first * P.pk_bytes,
                          sk + (size_t)first * P.sk_bytes};
        return prove_resident(c, count, false, &kg);
    });
    GUARD_END
}
int kosk_verify_resident_pk(kosk_ctx *ctx, int n, const uint8_t *pk, uint8_t *ok)
{
    if (!ctx || n < 1 || n > ctx->max_batch || !ok) return bad_args(ctx, __func__);
    GUARD(ctx)
    if (ctx->cohort) return combined_verify(ctx, n, VerifyCall{pk, ok});
    const Params &P = ctx->c->P;
    reset_masks(ctx, n);
    if (ctx->run(n, [&](Ctx &c, int first, int count) {
            return verify_into(ctx, c, first, count, ok, pk ? 1 : 2, pk ? pk + (size_t)first * P.pk_bytes : nullptr);
        })) return -1;
    ctx->masks_n = n;
    return 0;
    GUARD_END
}
int kosk_set_round_hook(kosk_ctx *ctx, kosk_round_fn fn, void *user)
{
    if (!ctx) return -1;
    GUARD(ctx)
    if (fn && ctx->sub.size() > 1) { ctx->err = "kosk_set_round_hook needs streams = 1 (one digest table per round)"; return -1; }
    for (Ctx *c : ctx->sub) { c->round_hook = fn; c->round_user = user; }
    return 0;
    GUARD_END
}
int kosk_resident_digests(kosk_ctx *ctx, int round, void **d_digests, size_t *stride)
{
    if (!ctx || round < 0 || round > 1) return bad_args(ctx, __func__);
    GUARD(ctx)
    if (ctx->sub.size() > 1) { ctx->err = "kosk_resident_digests needs streams = 1 (sub-batches keep separate tables)"; return -1; }
    if (d_digests) *d_digests = round ? ctx->c->d_dig2 : ctx->c->d_dig1;
    if (stride) *stride = (size_t)NPARTY * 32;
    return 0;
    GUARD_END
}

int kosk_verifiable_keygen_batch(kosk_ctx *ctx, int n, const uint8_t *tapes, size_t tape_stride,
                                 uint8_t *pk, uint8_t *sk, uint8_t *pi)
{
    if (!ctx || n < 0 || !pk || !sk || !pi) return bad_args(ctx, __func__);
    if (n == 0) return 0;
    GUARD(ctx)
    const Params &P = ctx->c->P;
    std::vector<uint8_t> drawn;
    if (!tapes) { // stateful callback: every tape sequentially, in proof order, then the chunks in parallel
        draw_tapes(ctx, n, drawn);
        tapes = drawn.data();
        tape_stride = P.tape_bytes;
    }
    // images go straight into a buffer the CALLER page-locked (KOSK_REGISTER=0: never), else through the pinned staging buffer
    const bool pinned = ctx->c->host_register && span_is_pinned(pi, (size_t)n * P.proof_bytes);
    return run_chunks(ctx, n, [&](Ctx &c, int first, int count) {
        const KeygenIn kg{tapes + (size_t)first * tape_stride, tape_stride, pk + (size_t)first * P.pk_bytes, sk + (size_t)first * P.sk_bytes};
        if (prove_resident(c, count, false, &kg)) return -1;
        return fetch_proofs(c, count, pi + (size_t)first * P.proof_bytes, pinned);
    });
    GUARD_END
}

int kosk_verify_batch(kosk_ctx *ctx, int n, const uint8_t *pi, const uint8_t *pk, uint8_t *ok)
{
    if (!ctx || n < 0 || !pi || !pk || !ok) return bad_args(ctx, __func__);
    if (n == 0) { ctx->masks_n = 0; return 0; }
    GUARD(ctx)
    const Params &P = ctx->c->P;
    reset_masks(ctx, n);
    const bool pinned = ctx->c->host_register && span_is_pinned(pi, (size_t)n * P.proof_bytes);
    const int rc = run_chunks(ctx, n, [&](Ctx &c, int first, int count) {
        const uint8_t *src = pi + (size_t)first * P.proof_bytes;
        c.host_img = nullptr;
        if (stage_verifier_inputs(c, count, src, pk + (size_t)first * P.pk_bytes, pinned)) return -1;
        // the verifier's host reads the unopened parties' digests straight from the caller's images: nothing to copy back for them
        // (verify_resident takes the pointer and clears it first thing)
        c.host_img = src;
        c.host_img_stride = P.proof_bytes;
        return verify_into(ctx, c, first, count, ok, 0, nullptr);
    });
    if (!rc) ctx->masks_n = n;
    return rc;
    GUARD_END
}

// The two host-buffer calls with the proofs in the compact wire format: packed / unpacked on the GPU, so PCIe
// carries 78 % of the image bytes in each direction.  Same chunking and lanes as the calls above.
int kosk_verifiable_keygen_batch_compact(kosk_ctx *ctx, int n, const uint8_t *tapes, size_t tape_stride,
                                         uint8_t *pk, uint8_t *sk, uint8_t *out)
{
    if (!ctx || n < 0 || !pk || !sk || !out) return bad_args(ctx, __func__);
    if (n == 0) return 0;
    GUARD(ctx)
    const Params &P = ctx->c->P;
    const size_t cb = make_compact_plan(P).bytes;
    std::vector<uint8_t> drawn;
    if (!tapes) {
        draw_tapes(ctx, n, drawn);
        tapes = drawn.data();
        tape_stride = P.tape_bytes;
    }
    const bool pinned = ctx->c->host_register && span_is_pinned(out, (size_t)n * cb);
    return run_chunks(ctx, n, [&](Ctx &c, int first, int count) {
        const KeygenIn kg{tapes + (size_t)first * tape_stride, tape_stride, pk + (size_t)first * P.pk_bytes, sk + (size_t)first * P.sk_bytes};
        if (prove_resident(c, count, false, &kg)) return -1;
        return fetch_proofs_compact(c, count, out + (size_t)first * cb, pinned);
    });
    GUARD_END
}
int kosk_verify_batch_compact(kosk_ctx *ctx, int n, const uint8_t *in, const uint8_t *pk, uint8_t *ok)
{
    if (!ctx || n < 0 || !in || !pk || !ok) return bad_args(ctx, __func__);
    if (n == 0) { ctx->masks_n = 0; return 0; }
    GUARD(ctx)
    const Params &P = ctx->c->P;
    const size_t cb = make_compact_plan(P).bytes;
    reset_masks(ctx, n);
    const bool pinned = ctx->c->host_register && span_is_pinned(in, (size_t)n * cb);
    const int rc = run_chunks(ctx, n, [&](Ctx &c, int first, int count) {
        if (stage_verifier_inputs_compact(c, count, in + (size_t)first * cb, pk + (size_t)first * P.pk_bytes, pinned)) return -1;
        return verify_into(ctx, c, first, count, ok, 0, nullptr);
    });
    if (!rc) ctx->masks_n = n;
    return rc;
    GUARD_END
}

// ---- second-level entry points (kosk_split.cpp) -----------------------------------------------------------
size_t kosk_randomness_bytes(int k) { Params p; return make_params(k, p) ? randomness_bytes(p) : 0; }
size_t kosk_range_proof_bytes(int k) { Params p; return make_params(k, p) ? range_proof_bytes(p) : 0; }
size_t kosk_mlwe_inst_bytes(int k) { Params p; return make_params(k, p) ? mlwe_inst_bytes(p) : 0; }

int kosk_prepare_randomness(kosk_ctx *ctx, int n, const uint8_t *tapes, size_t tape_stride, uint8_t *rand_out)
{
    if (!ctx || n < 0 || !rand_out) return bad_args(ctx, __func__);
    GUARD(ctx)
    Ctx &c = *ctx->c;
    ctx->clear_err();
    for (int done = 0; done < n;) {
        const int m = (n - done) < c.own_batch ? (n - done) : c.own_batch;
        if (prepare_randomness(c, m, tapes ? tapes + (size_t)done * tape_stride : nullptr, tape_stride,
                               rand_out + (size_t)done * randomness_bytes(c.P))) return -1;
        done += m;
    }
    return 0;
    GUARD_END
}
int kosk_prepare_range_proof(kosk_ctx *ctx, int n, const uint8_t *tapes, size_t tape_stride, uint8_t *range_out)
{
    if (!ctx || n < 0 || !range_out) return bad_args(ctx, __func__);
    GUARD(ctx)
    Ctx &c = *ctx->c;
    ctx->clear_err();
    for (int done = 0; done < n;) {
        const int m = (n - done) < c.own_batch ? (n - done) : c.own_batch;
        if (prepare_range_proof(c, m, tapes ? tapes + (size_t)done * tape_stride : nullptr, tape_stride,
                                range_out + (size_t)done * range_proof_bytes(c.P))) return -1;
        done += m;
    }
    return 0;
    GUARD_END
}
int kosk_prove_prepared(kosk_ctx *ctx, int n, const uint8_t *inst, const uint8_t *rand_in, const uint8_t *range_in,
                        const uint8_t *tapes, size_t tape_stride, uint8_t *pi)
{
    if (!ctx || n < 0 || !inst || !rand_in || !range_in || !pi) return bad_args(ctx, __func__);
    GUARD(ctx)
    Ctx &c = *ctx->c;
    ctx->clear_err();
    for (int done = 0; done < n;) {
        const int m = (n - done) < c.own_batch ? (n - done) : c.own_batch;
        if (prove_prepared(c, m, inst + (size_t)done * mlwe_inst_bytes(c.P), rand_in + (size_t)done * randomness_bytes(c.P),
                           range_in + (size_t)done * range_proof_bytes(c.P), tapes ? tapes + (size_t)done * tape_stride : nullptr,
                           tape_stride, pi + (size_t)done * c.P.proof_bytes)) return -1;
        done += m;
    }
    return 0;
    GUARD_END
}
int kosk_verify_inst(kosk_ctx *ctx, int n, const uint8_t *pi, const uint8_t *inst, uint8_t *ok)
{
    if (!ctx || n < 0 || !pi || !inst || !ok) return bad_args(ctx, __func__);
    GUARD(ctx)
    Ctx &c = *ctx->c;
    ctx->clear_err();
    reset_masks(ctx, n);
    for (int done = 0; done < n;) {
        const int m = (n - done) < c.own_batch ? (n - done) : c.own_batch;
        if (stage_verifier_inst(c, m, pi + (size_t)done * c.P.proof_bytes, inst + (size_t)done * mlwe_inst_bytes(c.P))) return -1;
        if (verify_into(ctx, c, done, m, ok, 0, nullptr)) return -1;
        done += m;
    }
    ctx->masks_n = n;
    return 0;
    GUARD_END
}

// ---- compact wire format (kosk_compact.hip) ------------------------------------------------------------------
size_t kosk_compact_proof_bytes(int k) { Params p; return make_params(k, p) ? make_compact_plan(p).bytes : 0; }
int kosk_proof_compress(int k, const uint8_t *pi, uint8_t *out)
{
    Params p;
    if (!make_params(k, p) || !pi || !out) return -1;
    return guard(static_cast<const kosk_ctx *>(nullptr), __func__, [&]() -> int {
    return compact_encode(p, pi, out);
    GUARD_END
}
int kosk_proof_decompress(int k, const uint8_t *in, uint8_t *pi)
{
    Params p;
    if (!make_params(k, p) || !pi || !in) return -1;
    return guard(static_cast<const kosk_ctx *>(nullptr), __func__, [&]() -> int {
    compact_decode(p, in, pi);
    return 0;
    GUARD_END
}
int kosk_fetch_proofs_compact(kosk_ctx *ctx, int n, uint8_t *out)
{
    if (!ctx || n < 0 || n > ctx->max_batch || !out) return bad_args(ctx, __func__);
    GUARD(ctx)
    const size_t cb = make_compact_plan(ctx->c->P).bytes;
    return ctx->run(n, [&](Ctx &c, int first, int count) { return fetch_proofs_compact(c, count, out + (size_t)first * cb); });
    GUARD_END
}
int kosk_stage_verifier_inputs_compact(kosk_ctx *ctx, int n, const uint8_t *in, const uint8_t *pk)
{
    if (!ctx || n < 0 || n > ctx->max_batch || !in || !pk) return bad_args(ctx, __func__);
    GUARD(ctx)
    const Params &P = ctx->c->P;
    const size_t cb = make_compact_plan(P).bytes;
    return ctx->run(n, [&](Ctx &c, int first, int count) {
        return stage_verifier_inputs_compact(c, count, in + (size_t)first * cb, pk + (size_t)first * P.pk_bytes);
    });
    GUARD_END
}

int kosk_verify_fail_masks(const kosk_ctx *ctx, uint32_t *masks, int n)
{
    // only the masks of the LAST verify call, and only if it completed: never stale entries of an earlier, larger batch
    if (!ctx || !masks || n < 0 || n > ctx->masks_n) return bad_args(ctx, __func__);
    memcpy(masks, ctx->masks.data(), sizeof(uint32_t) * (size_t)n);
    return 0;
}

void *kosk_host_alloc(size_t bytes)
{
    void *p = nullptr;
    if (!bytes || hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    return p;
}
void kosk_host_free(void *p)
{
    if (p && hipHostFree(p) != hipSuccess) (void)hipGetLastError();
}

int kosk_path_count(const kosk_ctx *ctx, int id, long *count)
{
    if (!ctx || id < 0 || id >= PATH_COUNT || !count) return bad_args(ctx, __func__);
    long v = 0;
    for (const Ctx *c : ctx->sub) v += c->path_n[id];
    *count = v;
    return 0;
}
int kosk_host_threads(const kosk_ctx *ctx) { return ctx ? ctx->c->nthreads : -1; }

int kosk_phase_seconds(const kosk_ctx *ctx, double *out, int n)
{
    if (!ctx || !out) return -1;
    for (int i = 0; i < n && i < PH_COUNT; i++) out[i] = ctx->c->phase_sec[i];
    return 0;
}

int kosk_profile_enable(kosk_ctx *ctx, int on)
{
    if (!ctx) return -1;
    GUARD(ctx)
    for (Ctx *cp : ctx->sub) {
        Ctx &c = *cp;
        c.prof_on = on < 0 ? 0 : (on > 2 ? 2 : on);
        for (int i = 0; i < PR_COUNT; i++) { c.prof_ms[i] = 0; c.prof_n[i] = 0; c.prof_units[i] = 0; c.prof_used[i] = false; }
    }
    return 0;
    GUARD_END
}
int kosk_profile_read(const kosk_ctx *ctx, int id, double *total_ms, long *launches)
{
    if (!ctx || id < 0 || id >= PR_COUNT) return bad_args(ctx, __func__);
    double ms = 0;
    long cnt = 0;
    for (const Ctx *c : ctx->sub) { ms += c->prof_ms[id]; cnt += c->prof_n[id]; }
    if (total_ms) *total_ms = ms;
    if (launches) *launches = cnt;
    return 0;
}

int kosk_profile_read_units(const kosk_ctx *ctx, int id, double *total_ms, long *launches, long *proofs)
{
    if (!ctx || id < 0 || id >= PR_COUNT) return bad_args(ctx, __func__);
    double ms = 0;
    long cnt = 0, units = 0;
    for (const Ctx *c : ctx->sub) { ms += c->prof_ms[id]; cnt += c->prof_n[id]; units += c->prof_units[id]; }
    if (total_ms) *total_ms = ms;
    if (launches) *launches = cnt;
    if (proofs) *proofs = units;
    return 0;
}
int kosk_combine_stats(const kosk_ctx *ctx, long *calls, long *members)
{
    if (!ctx) return -1;
    if (calls) *calls = ctx->merged_calls;
    if (members) *members = ctx->merged_members;
    return 0;
}

int kosk_stream_timer_start(kosk_ctx *ctx)
{
    if (!ctx) return -1;
    GUARD(ctx)
    Ctx &c = *ctx->c;
    ctx->clear_err();
    HIPCHK_C(hipSetDevice(c.device));
    if (!c.timer_ev[0]) { HIPCHK_C(hipEventCreate(&c.timer_ev[0])); HIPCHK_C(hipEventCreate(&c.timer_ev[1])); }
    HIPCHK_C(hipEventRecord(c.timer_ev[0], c.stream));
    return 0;
    GUARD_END
}
int kosk_stream_timer_stop(kosk_ctx *ctx, double *ms)
{
    if (!ctx || !ctx->c->timer_ev[0]) return bad_args(ctx, __func__);
    GUARD(ctx)
    Ctx &c = *ctx->c;
    ctx->clear_err();
    HIPCHK_C(hipSetDevice(c.device));
    HIPCHK_C(hipEventRecord(c.timer_ev[1], c.stream));
    HIPCHK_C(hipEventSynchronize(c.timer_ev[1]));
    float f = 0;
    HIPCHK_C(hipEventElapsedTime(&f, c.timer_ev[0], c.timer_ev[1]));
    if (ms) *ms = f;
    return 0;
    GUARD_END
}

int kosk_device_synchronize(kosk_ctx *ctx)
{
    if (!ctx) return -1;
    GUARD(ctx)
    ctx->clear_err();
    for (Ctx *cp : ctx->sub) {
        Ctx &c = *cp;
        HIPCHK_C(hipSetDevice(c.device));
        HIPCHK_C(hipStreamSynchronize(c.stream));
    }
    return 0;
    GUARD_END
}
int kosk_streams(const kosk_ctx *ctx) { return ctx ? (int)ctx->sub.size() : -1; }

int kosk_resident_proofs(kosk_ctx *ctx, void **d_proofs, size_t *stride)
{
    if (!ctx) return -1;
    GUARD(ctx)
    if (ctx->sub.size() > 1) { ctx->err = "kosk_resident_proofs needs streams = 1 (sub-batches keep separate images)"; return -1; }
    if (d_proofs) *d_proofs = ctx->c->d_proof;
    if (stride) *stride = ctx->c->image_stride;
    return 0;
    GUARD_END
}

// ---- kernel-level entry points -------------------------------------------------

int kosk_sha3_256_batch(kosk_ctx *ctx, const uint8_t *d_in, size_t in_stride, size_t inlen, uint8_t *d_out, int n)
{
    if (!ctx) return -1;
    GUARD(ctx)
    Ctx &c = *ctx->c;
    ctx->clear_err();
    HIPCHK_C(hipSetDevice(c.device));
    HIPCHK_C(launch_sha3_msgs(d_in, in_stride, (int)inlen, d_out, 32, 32, n, 0x06, c.stream));
    return 0;
    GUARD_END
}
int kosk_shake256_batch(kosk_ctx *ctx, const uint8_t *d_in, size_t in_stride, size_t inlen, uint8_t *d_out, size_t outlen, int n)
{
    if (!ctx) return -1;
    GUARD(ctx)
    Ctx &c = *ctx->c;
    ctx->clear_err();
    HIPCHK_C(hipSetDevice(c.device));
    HIPCHK_C(launch_sha3_msgs(d_in, in_stride, (int)inlen, d_out, outlen, (int)outlen, n, 0x1F, c.stream));
    return 0;
    GUARD_END
}

int kosk_sha3_256_batch_pair(kosk_ctx *ctx, const uint8_t *d_in, size_t in_stride, size_t inlen, uint8_t *d_out, int n)
{
    if (!ctx) return -1;
    GUARD(ctx)
    Ctx &c = *ctx->c;
    ctx->clear_err();
    HIPCHK_C(hipSetDevice(c.device));
    HIPCHK_C(launch_sha3_msgs_pair(d_in, in_stride, (int)inlen, d_out, 32, 32, n, 0x06, c.stream));
    return 0;
    GUARD_END
}

// sha3_256 of n LONG messages, one wave per message (the sponge of the device Fiat-Shamir rounds, kosk_fs_kernels.hip)
int kosk_sha3_256_batch_wave(kosk_ctx *ctx, const uint8_t *d_in, size_t in_stride, size_t inlen, uint8_t *d_out, int n)
{
    if (!ctx || !d_in || !d_out || n < 1 || inlen > (size_t)1 << 30 || (reinterpret_cast<uintptr_t>(d_in) & 7) || (in_stride & 7)) return bad_args(ctx, __func__);
    GUARD(ctx)
    Ctx &c = *ctx->c;
    ctx->clear_err();
    HIPCHK_C(hipSetDevice(c.device));
    FsArgs fa{};
    fa.in = d_in; fa.in_stride = in_stride; fa.len = (int)inlen; fa.out_digest = d_out;
    HIPCHK_C(launch_fs_chain(fa, FS_DIGEST, n, c.stream));
    return 0;
    GUARD_END
}
// kosk_fs_alpha / kosk_fs_opened for n digest tables in HBM (what the device Fiat-Shamir mode of a handle runs inside its calls)
int kosk_fs_alpha_device(kosk_ctx *ctx, const uint8_t *d_tables, size_t table_stride, int n, uint16_t *d_alpha, uint8_t *d_h1)
{
    if (!ctx || !d_tables || !d_alpha || n < 1 || (reinterpret_cast<uintptr_t>(d_tables) & 7) || (table_stride & 7)) return bad_args(ctx, __func__);
    GUARD(ctx)
    Ctx &c = *ctx->c;
    ctx->clear_err();
    HIPCHK_C(hipSetDevice(c.device));
    FsArgs fa{};
    fa.in = d_tables; fa.in_stride = table_stride; fa.len = NPARTY * 32; fa.out_digest = d_h1;
    fa.alpha = d_alpha; fa.alpha_stride = 80; fa.J = c.P.J;
    HIPCHK_C(launch_fs_chain(fa, FS_ALPHA, n, c.stream));
    return 0;
    GUARD_END
}
int kosk_fs_opened_device(kosk_ctx *ctx, const uint8_t *d_tables, size_t table_stride, int n, uint16_t *d_sel, uint16_t *d_rest, int sel_stride, uint8_t *d_ch)
{
    if (!ctx || !d_tables || !d_sel || !d_rest || n < 1 || sel_stride < NREST || sel_stride < SEL_OPOS + NOPEN ||
        (reinterpret_cast<uintptr_t>(d_tables) & 7) || (table_stride & 7)) return bad_args(ctx, __func__);
    GUARD(ctx)
    Ctx &c = *ctx->c;
    ctx->clear_err();
    HIPCHK_C(hipSetDevice(c.device));
    FsArgs fa{};
    fa.in = d_tables; fa.in_stride = table_stride; fa.len = NPARTY * 32; fa.out_digest = d_ch;
    fa.I = d_sel; fa.rest = d_rest; fa.sel_stride = sel_stride;
    HIPCHK_C(launch_fs_chain(fa, FS_OPENED, n, c.stream));
    return 0;
    GUARD_END
}

int kosk_commit_hash_lanes(kosk_ctx *ctx, const uint16_t *d_rows, size_t row_stride, int n_lanes,
                           const uint8_t *d_prefix, int with_prefix, uint8_t *d_out)
{
    if (!ctx) return -1;
    GUARD(ctx)
    Ctx &c = *ctx->c;
    ctx->clear_err();
    HIPCHK_C(hipSetDevice(c.device));
    HashArgs ha{};
    ha.rows = d_rows;
    ha.group_stride = 0;
    ha.row_stride = (int)row_stride;
    ha.col_off = 0;
    ha.lanes_per_group = n_lanes;
    ha.lane_map = nullptr;
    ha.prefix = d_prefix;
    ha.out = d_out;
    ha.out_lanes_per_group = n_lanes;
    int variant = 0;
    HIPCHK_C(launch_commit_hash(ha, 1, c.P.K, with_prefix != 0, c.stream, &variant));
    c.path_n[(variant & 1) ? PATH_HASH_DMA : PATH_HASH_PLAIN]++;
    return 0;
    GUARD_END
}

int kosk_ntt256_batch(kosk_ctx *ctx, const int16_t *d_in, int16_t *d_out, int n)
{
    if (!ctx) return -1;
    GUARD(ctx)
    Ctx &c = *ctx->c;
    ctx->clear_err();
    HIPCHK_C(hipSetDevice(c.device));
    NttArgs na{};
    na.in = d_in;
    na.out = d_out;
    na.npg = n;
    na.npoly = n;
    na.out_canonical = 0;
    HIPCHK_C(launch_ntt(na, c.stream));
    return 0;
    GUARD_END
}

int kosk_lagrange_expand(kosk_ctx *ctx, const uint16_t *d_y407, uint16_t *d_shares, int n)
{
    if (!ctx) return -1;
    GUARD(ctx)
    Ctx &c = *ctx->c;
    ctx->clear_err();
    HIPCHK_C(hipSetDevice(c.device));
    const int cap_rows = c.own_batch * c.rm.nrows; // the row matrix doubles as scratch (this handle's own block of it)
    const int cap = std::min<long>(cap_rows, (long)(c.limb_cap / (7 * 128)) - 64);
    for (int done = 0; done < n;) {
        const int m = (n - done) < cap ? (n - done) : cap;
        HIPCHK_C(launch_rows_copy(d_y407 + (size_t)done * XLEN, XLEN, c.d_P, RS, XLEN, m, c.stream));
        const GemmSrc gs{c.d_P, 0, nullptr, RS, 0, XLEN, 0}; // caller data: folded while converted
        const GemmDst gd{c.d_P, 0, nullptr, RS, EXP_OFF};
        if (gemm_modq(c, c.t_expand, gs, gd, m, 1)) return -1;
        HIPCHK_C(launch_rows_copy(c.d_P + NSEC, RS, d_shares + (size_t)done * NPARTY, NPARTY, NPARTY, m, c.stream));
        done += m;
    }
    return 0;
    GUARD_END
}

int kosk_recon_secrets(kosk_ctx *ctx, const uint16_t *d_shares, uint16_t *d_secrets, int n, int two_d)
{
    if (!ctx) return -1;
    GUARD(ctx)
    Ctx &c = *ctx->c;
    ctx->clear_err();
    HIPCHK_C(hipSetDevice(c.device));
    const GemmTable &t = two_d ? c.t_recon_2d : c.t_recon_d;
    const int cap_rows = c.own_batch * c.rm.nrows;
    const int cap = std::min<long>(cap_rows, (long)(c.limb_cap / ((size_t)t.KS * 128)) - 64);
    for (int done = 0; done < n;) {
        const int m = (n - done) < cap ? (n - done) : cap;
        HIPCHK_C(launch_rows_copy(d_shares + (size_t)done * NPARTY, NPARTY, c.d_P + NSEC, RS, NPARTY, m, c.stream));
        const GemmSrc gs{c.d_P, 0, nullptr, RS, NSEC, t.Kdim, 0};
        const GemmDst gd{c.d_P, 0, nullptr, RS, 0};
        if (gemm_modq(c, t, gs, gd, m, 1)) return -1;
        HIPCHK_C(launch_rows_copy(c.d_P, RS, d_secrets + (size_t)done * NSEC, NSEC, NSEC, m, c.stream));
        done += m;
    }
    return 0;
    GUARD_END
}

// ---- host-only entry points -----------------------------------------------------

int kosk_keygen(int kyber_k, const uint8_t seed64[64], uint8_t *pk, uint8_t *sk, int16_t *A, int16_t *s, int16_t *e, int16_t *t)
{
    Params P;
    if (!make_params(kyber_k, P) || !seed64 || !pk || !sk) return -1;
    return guard(static_cast<const kosk_ctx *>(nullptr), __func__, [&]() -> int {
    std::unique_ptr<HostKey> key(new HostKey());
    host_keygen(P, seed64, pk, sk, *key);
    const int K = P.K;
    if (A) memcpy(A, key->A, sizeof(int16_t) * K * K * 256);
    if (s) memcpy(s, key->se, sizeof(int16_t) * K * 256);
    if (e) memcpy(e, key->se + K * 256, sizeof(int16_t) * K * 256);
    if (t) memcpy(t, key->t, sizeof(int16_t) * K * 256);
    return 0;
    GUARD_END
}

int kosk_fs_alpha(int kyber_k, const uint8_t *tcomm_all, uint16_t *alpha)
{
    Params P;
    if (!make_params(kyber_k, P) || !tcomm_all || !alpha) return -1;
    return guard(static_cast<const kosk_ctx *>(nullptr), __func__, [&]() -> int {
    fs_alpha(P, tcomm_all, alpha);
    return 0;
    GUARD_END
}

int kosk_fs_opened(const uint8_t *digests_all, uint16_t *I, uint16_t *rest)
{
    if (!digests_all || !I || !rest) return -1;
    return guard(static_cast<const kosk_ctx *>(nullptr), __func__, [&]() -> int {
    fs_opened(digests_all, I, rest);
    return 0;
    GUARD_END
}

void kosk_host_sha3_256(uint8_t out[32], const uint8_t *in, size_t inlen) { sha3_256(out, in, inlen); }
void kosk_host_shake256(uint8_t *out, size_t outlen, const uint8_t *in, size_t inlen) { shake256(out, outlen, in, inlen); }

int kosk_host_sha3_256_multi(uint8_t *out, const uint8_t *in, size_t in_stride, size_t inlen, int count, int nthreads)
{
    // same code path as the batched Fiat-Shamir rounds: SIMD groups spread over the pool
    if (!out || !in || count < 0) return -1;
    return guard(static_cast<const kosk_ctx *>(nullptr), __func__, [&]() -> int {
    std::vector<const uint8_t *> ptr(count);
    for (int i = 0; i < count; i++) ptr[i] = in + (size_t)i * in_stride;
    const int w = sha3_multi_width();
    if (nthreads <= 1) { sha3_256_multi(out, ptr.data(), inlen, count); return w; }
    const int groups = (count + w - 1) / w;
    parallel_for(groups, nthreads, [&](int g) {
        const int lo = g * w, n = (count - lo) < w ? (count - lo) : w;
        sha3_256_multi(out + 32 * (size_t)lo, ptr.data() + lo, inlen, n);
    });
    return w;
    GUARD_END
}

int kosk_lagrange_table(int which, uint16_t *out)
{
    if (!out) return -1;
    return guard(static_cast<const kosk_ctx *>(nullptr), __func__, [&]() -> int {
    if (which == 0) {
        for (int x = 0; x < NPARTY - NOPEN - 1; x++) lagrange_row(out + (size_t)x * XLEN, XLEN, 0, XLEN + x);
    } else if (which == 1) {
        for (int i = 0; i < NSEC; i++) lagrange_row(out + (size_t)i * XLEN, XLEN, NSEC, i);
    } else if (which == 2) {
        for (int i = 0; i < NSEC; i++) lagrange_row(out + (size_t)i * (DEG2 + 1), DEG2 + 1, NSEC, i);
    } else {
        return -1;
    }
    return 0;
    GUARD_END
}

} // extern "C"
