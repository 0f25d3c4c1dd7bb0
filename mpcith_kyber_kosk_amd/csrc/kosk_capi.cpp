// extern "C" boundary (include/kosk_mi355x.h) over kosk::Ctx.
#include "../../include/kosk_mi355x.h"

#include <algorithm>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "kosk_ctx.hpp"

using namespace kosk;

// A handle owns S sub-contexts (own stream + HBM workspace each).  A batch call splits its proofs into S
// contiguous sub-batches that run concurrently on S host threads: while one sub-batch sits in a host
// Fiat-Shamir round trip the GPU works on the other, and kernels of different streams share the CUs.
struct kosk_ctx {
    std::vector<Ctx *> sub;
    Ctx *c; // sub[0]: kernel-level entry points, error text, sizes
    int max_batch;
    std::string err;
    std::vector<uint32_t> masks; // fail masks of the last verify call, in the caller's proof order

    // [first, count) of sub-batch i of an n-proof call
    void split(int n, int i, int &first, int &count) const
    {
        const int S = (int)sub.size();
        const int base = n / S, rem = n % S;
        count = base + (i < rem ? 1 : 0);
        first = i * base + (i < rem ? i : rem);
    }
    // every entry point starts from a clean error state (kosk_last_error never reports a stale message)
    void clear_err()
    {
        err.clear();
        for (Ctx *x : sub) x->err.clear();
    }
    template <typename F>
    int run(int n, F &&fn)
    {
        clear_err();
        const int S = (int)sub.size();
        std::vector<int> rc(S, 0);
        std::vector<std::thread> th;
        for (int i = 1; i < S; i++) {
            int first, count;
            split(n, i, first, count);
            if (count > 0) th.emplace_back([&, i, first, count] { rc[i] = fn(*sub[i], first, count); });
        }
        int first, count;
        split(n, 0, first, count);
        if (count > 0) rc[0] = fn(*sub[0], first, count);
        for (auto &t : th) t.join();
        for (int i = 0; i < S; i++)
            if (rc[i]) { err = sub[i]->err; c->err = err; return -1; }
        return 0;
    }
};

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
// sub-contexts, which work through theirs concurrently on S host threads.  With KOSK_STREAMS >= 2 one chunk's PCIe
// transfers (tapes in, 0.68 MB of proof image out per proof; proofs and keys in for the verifier) and host hashing run
// under another chunk's kernels: the streaming write-back of SURVEY.md 8 f4 for a single caller thread.
template <typename F>
static int run_chunks(kosk_ctx *h, int n, F &&fn)
{
    h->clear_err();
    const int S = (int)h->sub.size(), per = h->sub[0]->max_batch, nchunks = (n + per - 1) / per;
    std::vector<int> rc(S, 0);
    auto lane = [&](int i) {
        for (int j = i; j < nchunks && !rc[i]; j += S) {
            const int first = j * per, count = (n - first) < per ? (n - first) : per;
            rc[i] = fn(*h->sub[i], first, count);
        }
    };
    std::vector<std::thread> th;
    for (int i = 1; i < S && i < nchunks; i++) th.emplace_back(lane, i);
    lane(0);
    for (auto &t : th) t.join();
    for (int i = 0; i < S; i++)
        if (rc[i]) { h->err = h->sub[i]->err; h->c->err = h->err; return -1; }
    return 0;
}

static bool register_enabled()
{
    static const bool on = !(getenv("KOSK_REGISTER") && atoi(getenv("KOSK_REGISTER")) == 0);
    return on;
}

static thread_local std::string g_create_err; // error text of the last failed kosk_create on this thread

#define HIPCHK_C(x)                                                        \
    do {                                                                   \
        hipError_t e_ = (x);                                               \
        if (e_ != hipSuccess) {                                            \
            c.err = std::string(#x) + ": " + hipGetErrorString(e_);        \
            return -1;                                                     \
        }                                                                  \
    } while (0)

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

int kosk_create(kosk_ctx **ctx, int device, int kyber_k, int max_batch)
{
    if (!ctx) return -1;
    g_create_err.clear();
    if (const char *e = getenv("AMD_DIRECT_DISPATCH"))
        if (atoi(e) == 0 && e[0] != '\0') {
            // measured on ROCm 7.2 (tools/stress.py): with direct dispatch off, hipStreamSynchronize returned before
            // device-to-host copies into pinned memory had landed; the host then hashed stale digests
            g_create_err = "AMD_DIRECT_DISPATCH=0 is not supported: stream synchronisation does not cover D2H copies in that mode";
            return -1;
        }
    int S = 1; // KOSK_STREAMS: sub-batches in flight per handle (1 measured best at 46 proofs: kernels sit on latency floors)
    if (const char *e = getenv("KOSK_STREAMS")) S = atoi(e) > 0 ? atoi(e) : S;
    if (S > max_batch) S = max_batch > 0 ? max_batch : 1;
    if (S > 8) S = 8;
    kosk_ctx *h = new kosk_ctx();
    h->max_batch = max_batch;
    const int per = (max_batch + S - 1) / S;
    for (int i = 0; i < S; i++) {
        Ctx *c = nullptr;
        if (ctx_create(&c, device, kyber_k, per, g_create_err)) {
            for (Ctx *x : h->sub) delete x;
            delete h;
            return -1;
        }
        h->sub.push_back(c);
    }
    h->c = h->sub[0];
    *ctx = h;
    return 0;
}
void kosk_destroy(kosk_ctx *ctx)
{
    if (!ctx) return;
    for (Ctx *c : ctx->sub) delete c;
    delete ctx;
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
    const Params &P = ctx->c->P;
    if (!tapes) {
        // the randombytes callback is stateful: draw every tape sequentially, in proof order, then stage in parallel
        std::vector<uint8_t> drawn((size_t)n * P.tape_bytes);
        Ctx &c0 = *ctx->c;
        uint8_t *tp = drawn.data();
        auto draw = [&](size_t len) { if (c0.rb) c0.rb(c0.rb_user, tp, len); else os_randombytes(tp, len); tp += len; };
        for (int b = 0; b < n; b++) { // reference call order: kosk.cpp:12, mlwe_prover.cpp:9, ss.cpp:5
            draw(64);
            for (int i = 0; i < P.M; i++) draw(32);
            for (int i = 0; i < P.nfresh; i++) draw(302);
        }
        return ctx->run(n, [&](Ctx &c, int first, int count) {
            return stage_prover_inputs(c, count, drawn.data() + (size_t)first * P.tape_bytes, P.tape_bytes,
                                       pk + (size_t)first * P.pk_bytes, sk + (size_t)first * P.sk_bytes);
        });
    }
    return ctx->run(n, [&](Ctx &c, int first, int count) {
        return stage_prover_inputs(c, count, tapes + (size_t)first * tape_stride, tape_stride,
                                   pk + (size_t)first * P.pk_bytes, sk + (size_t)first * P.sk_bytes);
    });
}
int kosk_prove_resident(kosk_ctx *ctx, int n)
{
    if (!ctx || n < 1 || n > ctx->max_batch) return bad_args(ctx, __func__);
    return ctx->run(n, [&](Ctx &c, int, int count) { return prove_resident(c, count); });
}
int kosk_fetch_proofs(kosk_ctx *ctx, int n, uint8_t *pi)
{
    if (!ctx || n < 1 || n > ctx->max_batch) return bad_args(ctx, __func__);
    const Params &P = ctx->c->P;
    return ctx->run(n, [&](Ctx &c, int first, int count) { return fetch_proofs(c, count, pi + (size_t)first * P.proof_bytes); });
}
int kosk_stage_verifier_inputs(kosk_ctx *ctx, int n, const uint8_t *pi, const uint8_t *pk)
{
    if (!ctx || n < 1 || n > ctx->max_batch) return bad_args(ctx, __func__);
    const Params &P = ctx->c->P;
    return ctx->run(n, [&](Ctx &c, int first, int count) {
        return stage_verifier_inputs(c, count, pi + (size_t)first * P.proof_bytes, pk + (size_t)first * P.pk_bytes);
    });
}
int kosk_verify_resident(kosk_ctx *ctx, int n, uint8_t *ok)
{
    if (!ctx || n < 1 || n > ctx->max_batch) return bad_args(ctx, __func__);
    ctx->masks.assign((size_t)n, 0);
    return ctx->run(n, [&](Ctx &c, int first, int count) {
        if (verify_resident(c, count, ok + first)) return -1;
        memcpy(ctx->masks.data() + first, c.h_fail, sizeof(uint32_t) * (size_t)count);
        return 0;
    });
}

int kosk_verifiable_keygen_resident(kosk_ctx *ctx, int n, const uint8_t *tapes, size_t tape_stride, uint8_t *pk, uint8_t *sk)
{
    if (!ctx || n < 1 || n > ctx->max_batch || !pk || !sk) return bad_args(ctx, __func__);
    const Params &P = ctx->c->P;
    if (!tapes && ctx->sub.size() > 1) {
        // the randombytes callback is stateful: draw sequentially in proof order, then prove the sub-batches in parallel
        std::vector<uint8_t> drawn((size_t)n * P.tape_bytes);
        Ctx &c0 = *ctx->c;
        uint8_t *tp = drawn.data();
        auto draw = [&](size_t len) { if (c0.rb) c0.rb(c0.rb_user, tp, len); else os_randombytes(tp, len); tp += len; };
        for (int b = 0; b < n; b++) {
            draw(64);
            for (int i = 0; i < P.M; i++) draw(32);
            for (int i = 0; i < P.nfresh; i++) draw(302);
        }
        return ctx->run(n, [&](Ctx &c, int first, int count) {
            const KeygenIn kg{drawn.data() + (size_t)first * P.tape_bytes, P.tape_bytes, pk + (size_t)first * P.pk_bytes, sk + (size_t)first * P.sk_bytes};
            return prove_resident(c, count, false, &kg);
        });
    }
    return ctx->run(n, [&](Ctx &c, int first, int count) {
        const KeygenIn kg{tapes ? tapes + (size_t)first * tape_stride : nullptr, tape_stride, pk + (size_t)first * P.pk_bytes,
                          sk + (size_t)first * P.sk_bytes};
        return prove_resident(c, count, false, &kg);
    });
}
int kosk_verify_resident_pk(kosk_ctx *ctx, int n, const uint8_t *pk, uint8_t *ok)
{
    if (!ctx || n < 1 || n > ctx->max_batch || !ok) return bad_args(ctx, __func__);
    const Params &P = ctx->c->P;
    ctx->masks.assign((size_t)n, 0);
    return ctx->run(n, [&](Ctx &c, int first, int count) {
        if (verify_resident(c, count, ok + first, pk ? 1 : 2, pk ? pk + (size_t)first * P.pk_bytes : nullptr)) return -1;
        memcpy(ctx->masks.data() + first, c.h_fail, sizeof(uint32_t) * (size_t)count);
        return 0;
    });
}
int kosk_set_round_hook(kosk_ctx *ctx, kosk_round_fn fn, void *user)
{
    if (!ctx) return -1;
    if (fn && ctx->sub.size() > 1) { ctx->err = "kosk_set_round_hook needs KOSK_STREAMS=1 (one digest table per round)"; return -1; }
    for (Ctx *c : ctx->sub) { c->round_hook = fn; c->round_user = user; }
    return 0;
}
int kosk_resident_digests(kosk_ctx *ctx, int round, void **d_digests, size_t *stride)
{
    if (!ctx || round < 0 || round > 1) return bad_args(ctx, __func__);
    if (ctx->sub.size() > 1) { ctx->err = "kosk_resident_digests needs KOSK_STREAMS=1 (sub-batches keep separate tables)"; return -1; }
    if (d_digests) *d_digests = round ? ctx->c->d_dig2 : ctx->c->d_dig1;
    if (stride) *stride = (size_t)NPARTY * 32;
    return 0;
}

int kosk_verifiable_keygen_batch(kosk_ctx *ctx, int n, const uint8_t *tapes, size_t tape_stride,
                                 uint8_t *pk, uint8_t *sk, uint8_t *pi)
{
    if (!ctx || n < 0 || !pk || !sk || !pi) return bad_args(ctx, __func__);
    if (n == 0) return 0;
    const Params &P = ctx->c->P;
    std::vector<uint8_t> drawn;
    if (!tapes) {
        // the randombytes callback is stateful: draw every tape sequentially, in proof order (reference call order:
        // kosk.cpp:12, mlwe_prover.cpp:9, ss.cpp:5), then work through the chunks in parallel
        drawn.resize((size_t)n * P.tape_bytes);
        Ctx &c0 = *ctx->c;
        uint8_t *tp = drawn.data();
        auto draw = [&](size_t len) { if (c0.rb) c0.rb(c0.rb_user, tp, len); else os_randombytes(tp, len); tp += len; };
        for (int b = 0; b < n; b++) {
            draw(64);
            for (int i = 0; i < P.M; i++) draw(32);
            for (int i = 0; i < P.nfresh; i++) draw(302);
        }
        tapes = drawn.data();
        tape_stride = P.tape_bytes;
    }
    // several chunks: page-lock the caller's proof buffer for the call so that the images are copied straight into it
    // (KOSK_REGISTER=0 keeps the pinned staging buffer + host memcpy)
    const bool reg = n > ctx->sub[0]->max_batch && register_enabled() &&
                     hipHostRegister(pi, (size_t)n * P.proof_bytes, hipHostRegisterDefault) == hipSuccess;
    const int rc = run_chunks(ctx, n, [&](Ctx &c, int first, int count) {
        const KeygenIn kg{tapes + (size_t)first * tape_stride, tape_stride, pk + (size_t)first * P.pk_bytes, sk + (size_t)first * P.sk_bytes};
        if (prove_resident(c, count, false, &kg)) return -1;
        return fetch_proofs(c, count, pi + (size_t)first * P.proof_bytes, reg);
    });
    if (reg) (void)hipHostUnregister(pi);
    else (void)hipGetLastError();
    return rc;
}

int kosk_verify_batch(kosk_ctx *ctx, int n, const uint8_t *pi, const uint8_t *pk, uint8_t *ok)
{
    if (!ctx || n < 0 || !pi || !pk || !ok) return bad_args(ctx, __func__);
    if (n == 0) return 0;
    const Params &P = ctx->c->P;
    ctx->masks.assign((size_t)n, 0);
    const bool reg = n > ctx->sub[0]->max_batch && register_enabled() &&
                     hipHostRegister(const_cast<uint8_t *>(pi), (size_t)n * P.proof_bytes, hipHostRegisterDefault) == hipSuccess;
    const int rc = run_chunks(ctx, n, [&](Ctx &c, int first, int count) {
        if (stage_verifier_inputs(c, count, pi + (size_t)first * P.proof_bytes, pk + (size_t)first * P.pk_bytes, reg)) return -1;
        if (verify_resident(c, count, ok + first)) return -1;
        memcpy(ctx->masks.data() + first, c.h_fail, sizeof(uint32_t) * (size_t)count);
        return 0;
    });
    if (reg) (void)hipHostUnregister(const_cast<uint8_t *>(pi));
    else (void)hipGetLastError();
    return rc;
}

// ---- second-level entry points (kosk_split.cpp) -----------------------------------------------------------
size_t kosk_randomness_bytes(int k) { Params p; return make_params(k, p) ? randomness_bytes(p) : 0; }
size_t kosk_range_proof_bytes(int k) { Params p; return make_params(k, p) ? range_proof_bytes(p) : 0; }
size_t kosk_mlwe_inst_bytes(int k) { Params p; return make_params(k, p) ? mlwe_inst_bytes(p) : 0; }

int kosk_prepare_randomness(kosk_ctx *ctx, int n, const uint8_t *tapes, size_t tape_stride, uint8_t *rand_out)
{
    if (!ctx || n < 0 || !rand_out) return bad_args(ctx, __func__);
    Ctx &c = *ctx->c;
    ctx->clear_err();
    for (int done = 0; done < n;) {
        const int m = (n - done) < c.max_batch ? (n - done) : c.max_batch;
        if (prepare_randomness(c, m, tapes ? tapes + (size_t)done * tape_stride : nullptr, tape_stride,
                               rand_out + (size_t)done * randomness_bytes(c.P))) return -1;
        done += m;
    }
    return 0;
}
int kosk_prepare_range_proof(kosk_ctx *ctx, int n, const uint8_t *tapes, size_t tape_stride, uint8_t *range_out)
{
    if (!ctx || n < 0 || !range_out) return bad_args(ctx, __func__);
    Ctx &c = *ctx->c;
    ctx->clear_err();
    for (int done = 0; done < n;) {
        const int m = (n - done) < c.max_batch ? (n - done) : c.max_batch;
        if (prepare_range_proof(c, m, tapes ? tapes + (size_t)done * tape_stride : nullptr, tape_stride,
                                range_out + (size_t)done * range_proof_bytes(c.P))) return -1;
        done += m;
    }
    return 0;
}
int kosk_prove_prepared(kosk_ctx *ctx, int n, const uint8_t *inst, const uint8_t *rand_in, const uint8_t *range_in,
                        const uint8_t *tapes, size_t tape_stride, uint8_t *pi)
{
    if (!ctx || n < 0 || !inst || !rand_in || !range_in || !pi) return bad_args(ctx, __func__);
    Ctx &c = *ctx->c;
    ctx->clear_err();
    for (int done = 0; done < n;) {
        const int m = (n - done) < c.max_batch ? (n - done) : c.max_batch;
        if (prove_prepared(c, m, inst + (size_t)done * mlwe_inst_bytes(c.P), rand_in + (size_t)done * randomness_bytes(c.P),
                           range_in + (size_t)done * range_proof_bytes(c.P), tapes ? tapes + (size_t)done * tape_stride : nullptr,
                           tape_stride, pi + (size_t)done * c.P.proof_bytes)) return -1;
        done += m;
    }
    return 0;
}
int kosk_verify_inst(kosk_ctx *ctx, int n, const uint8_t *pi, const uint8_t *inst, uint8_t *ok)
{
    if (!ctx || n < 0 || !pi || !inst || !ok) return bad_args(ctx, __func__);
    Ctx &c = *ctx->c;
    ctx->clear_err();
    for (int done = 0; done < n;) {
        const int m = (n - done) < c.max_batch ? (n - done) : c.max_batch;
        if (stage_verifier_inst(c, m, pi + (size_t)done * c.P.proof_bytes, inst + (size_t)done * mlwe_inst_bytes(c.P))) return -1;
        if (verify_resident(c, m, ok + done)) return -1;
        if (ctx->masks.size() < (size_t)n) ctx->masks.resize((size_t)n, 0);
        memcpy(ctx->masks.data() + done, c.h_fail, sizeof(uint32_t) * (size_t)m);
        done += m;
    }
    return 0;
}

// ---- compact wire format (kosk_compact.hip) ------------------------------------------------------------------
size_t kosk_compact_proof_bytes(int k) { Params p; return make_params(k, p) ? make_compact_plan(p).bytes : 0; }
int kosk_proof_compress(int k, const uint8_t *pi, uint8_t *out)
{
    Params p;
    if (!make_params(k, p) || !pi || !out) return -1;
    return compact_encode(p, pi, out);
}
int kosk_proof_decompress(int k, const uint8_t *in, uint8_t *pi)
{
    Params p;
    if (!make_params(k, p) || !pi || !in) return -1;
    compact_decode(p, in, pi);
    return 0;
}
int kosk_fetch_proofs_compact(kosk_ctx *ctx, int n, uint8_t *out)
{
    if (!ctx || n < 0 || n > ctx->max_batch || !out) return bad_args(ctx, __func__);
    const size_t cb = make_compact_plan(ctx->c->P).bytes;
    return ctx->run(n, [&](Ctx &c, int first, int count) { return fetch_proofs_compact(c, count, out + (size_t)first * cb); });
}
int kosk_stage_verifier_inputs_compact(kosk_ctx *ctx, int n, const uint8_t *in, const uint8_t *pk)
{
    if (!ctx || n < 0 || n > ctx->max_batch || !in || !pk) return bad_args(ctx, __func__);
    const Params &P = ctx->c->P;
    const size_t cb = make_compact_plan(P).bytes;
    return ctx->run(n, [&](Ctx &c, int first, int count) {
        return stage_verifier_inputs_compact(c, count, in + (size_t)first * cb, pk + (size_t)first * P.pk_bytes);
    });
}

int kosk_verify_fail_masks(const kosk_ctx *ctx, uint32_t *masks, int n)
{
    if (!ctx || !masks || n < 0 || (size_t)n > ctx->masks.size()) return bad_args(ctx, __func__);
    memcpy(masks, ctx->masks.data(), sizeof(uint32_t) * (size_t)n);
    return 0;
}

int kosk_phase_seconds(const kosk_ctx *ctx, double *out, int n)
{
    if (!ctx) return -1;
    for (int i = 0; i < n && i < PH_COUNT; i++) out[i] = ctx->c->phase_sec[i];
    return 0;
}

int kosk_profile_enable(kosk_ctx *ctx, int on)
{
    if (!ctx) return -1;
    for (Ctx *cp : ctx->sub) {
        Ctx &c = *cp;
        c.prof_on = on < 0 ? 0 : (on > 2 ? 2 : on);
        for (int i = 0; i < PR_COUNT; i++) { c.prof_ms[i] = 0; c.prof_n[i] = 0; c.prof_used[i] = false; }
    }
    return 0;
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

int kosk_stream_timer_start(kosk_ctx *ctx)
{
    if (!ctx) return -1;
    Ctx &c = *ctx->c;
    ctx->clear_err();
    HIPCHK_C(hipSetDevice(c.device));
    if (!c.timer_ev[0]) { HIPCHK_C(hipEventCreate(&c.timer_ev[0])); HIPCHK_C(hipEventCreate(&c.timer_ev[1])); }
    HIPCHK_C(hipEventRecord(c.timer_ev[0], c.stream));
    return 0;
}
int kosk_stream_timer_stop(kosk_ctx *ctx, double *ms)
{
    if (!ctx || !ctx->c->timer_ev[0]) return bad_args(ctx, __func__);
    Ctx &c = *ctx->c;
    ctx->clear_err();
    HIPCHK_C(hipSetDevice(c.device));
    HIPCHK_C(hipEventRecord(c.timer_ev[1], c.stream));
    HIPCHK_C(hipEventSynchronize(c.timer_ev[1]));
    float f = 0;
    HIPCHK_C(hipEventElapsedTime(&f, c.timer_ev[0], c.timer_ev[1]));
    if (ms) *ms = f;
    return 0;
}

int kosk_device_synchronize(kosk_ctx *ctx)
{
    if (!ctx) return -1;
    ctx->clear_err();
    for (Ctx *cp : ctx->sub) {
        Ctx &c = *cp;
        HIPCHK_C(hipSetDevice(c.device));
        HIPCHK_C(hipStreamSynchronize(c.stream));
    }
    return 0;
}
int kosk_commit_launch_groups(const kosk_ctx *ctx, int n, int *main_groups)
{
    if (!ctx || n < 1 || !main_groups) return bad_args(ctx, __func__);
    *main_groups = commit_hash_groups(*ctx->c, n);
    return 0;
}
int kosk_streams(const kosk_ctx *ctx) { return ctx ? (int)ctx->sub.size() : -1; }

int kosk_resident_proofs(kosk_ctx *ctx, void **d_proofs, size_t *stride)
{
    if (!ctx) return -1;
    if (ctx->sub.size() > 1) { ctx->err = "kosk_resident_proofs needs KOSK_STREAMS=1 (sub-batches keep separate images)"; return -1; }
    if (d_proofs) *d_proofs = ctx->c->d_proof;
    if (stride) *stride = ctx->c->image_stride;
    return 0;
}

// ---- kernel-level entry points -------------------------------------------------

int kosk_sha3_256_batch(kosk_ctx *ctx, const uint8_t *d_in, size_t in_stride, size_t inlen, uint8_t *d_out, int n)
{
    if (!ctx) return -1;
    Ctx &c = *ctx->c;
    ctx->clear_err();
    HIPCHK_C(hipSetDevice(c.device));
    HIPCHK_C(launch_sha3_msgs(d_in, in_stride, (int)inlen, d_out, 32, 32, n, 0x06, c.stream));
    return 0;
}
int kosk_shake256_batch(kosk_ctx *ctx, const uint8_t *d_in, size_t in_stride, size_t inlen, uint8_t *d_out, size_t outlen, int n)
{
    if (!ctx) return -1;
    Ctx &c = *ctx->c;
    ctx->clear_err();
    HIPCHK_C(hipSetDevice(c.device));
    HIPCHK_C(launch_sha3_msgs(d_in, in_stride, (int)inlen, d_out, outlen, (int)outlen, n, 0x1F, c.stream));
    return 0;
}

int kosk_commit_hash_lanes(kosk_ctx *ctx, const uint16_t *d_rows, size_t row_stride, int n_lanes,
                           const uint8_t *d_prefix, int with_prefix, uint8_t *d_out)
{
    if (!ctx) return -1;
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
    HIPCHK_C(launch_commit_hash(ha, 1, c.P.K, with_prefix != 0, c.stream));
    return 0;
}

int kosk_ntt256_batch(kosk_ctx *ctx, const int16_t *d_in, int16_t *d_out, int n)
{
    if (!ctx) return -1;
    Ctx &c = *ctx->c;
    ctx->clear_err();
    HIPCHK_C(hipSetDevice(c.device));
    NttArgs na{};
    na.in = d_in;
    na.out = d_out;
    na.npg = n;
    na.npoly = n;
    na.out_canonical = 0; na.fp32 = c.ntt_fp32;
    HIPCHK_C(launch_ntt(na, c.stream));
    return 0;
}

int kosk_lagrange_expand(kosk_ctx *ctx, const uint16_t *d_y407, uint16_t *d_shares, int n)
{
    if (!ctx) return -1;
    Ctx &c = *ctx->c;
    ctx->clear_err();
    HIPCHK_C(hipSetDevice(c.device));
    const int cap_rows = c.max_batch * c.rm.nrows; // the row matrix doubles as scratch
    const int cap = std::min<long>(cap_rows, (long)(c.limb_cap / (7 * 128)) - 64);
    for (int done = 0; done < n;) {
        const int m = (n - done) < cap ? (n - done) : cap;
        HIPCHK_C(launch_rows_copy(d_y407 + (size_t)done * XLEN, XLEN, c.d_P, RS, XLEN, m, c.stream));
        const GemmSrc gs{c.d_P, 0, nullptr, RS, 0, XLEN};
        const GemmDst gd{c.d_P, 0, nullptr, RS, EXP_OFF};
        if (gemm_modq(c, c.t_expand, gs, gd, m, 1)) return -1;
        HIPCHK_C(launch_rows_copy(c.d_P + NSEC, RS, d_shares + (size_t)done * NPARTY, NPARTY, NPARTY, m, c.stream));
        done += m;
    }
    return 0;
}

int kosk_recon_secrets(kosk_ctx *ctx, const uint16_t *d_shares, uint16_t *d_secrets, int n, int two_d)
{
    if (!ctx) return -1;
    Ctx &c = *ctx->c;
    ctx->clear_err();
    HIPCHK_C(hipSetDevice(c.device));
    const GemmTable &t = two_d ? c.t_recon_2d : c.t_recon_d;
    const int cap_rows = c.max_batch * c.rm.nrows;
    const int cap = std::min<long>(cap_rows, (long)(c.limb_cap / ((size_t)t.KS * 128)) - 64);
    for (int done = 0; done < n;) {
        const int m = (n - done) < cap ? (n - done) : cap;
        HIPCHK_C(launch_rows_copy(d_shares + (size_t)done * NPARTY, NPARTY, c.d_P + NSEC, RS, NPARTY, m, c.stream));
        const GemmSrc gs{c.d_P, 0, nullptr, RS, NSEC, t.Kdim};
        const GemmDst gd{c.d_P, 0, nullptr, RS, 0};
        if (gemm_modq(c, t, gs, gd, m, 1)) return -1;
        HIPCHK_C(launch_rows_copy(c.d_P, RS, d_secrets + (size_t)done * NSEC, NSEC, NSEC, m, c.stream));
        done += m;
    }
    return 0;
}

// ---- host-only entry points -----------------------------------------------------

int kosk_keygen(int kyber_k, const uint8_t seed64[64], uint8_t *pk, uint8_t *sk, int16_t *A, int16_t *s, int16_t *e, int16_t *t)
{
    Params P;
    if (!make_params(kyber_k, P) || !seed64 || !pk || !sk) return -1;
    HostKey *key = new HostKey();
    host_keygen(P, seed64, pk, sk, *key);
    const int K = P.K;
    if (A) memcpy(A, key->A, sizeof(int16_t) * K * K * 256);
    if (s) memcpy(s, key->se, sizeof(int16_t) * K * 256);
    if (e) memcpy(e, key->se + K * 256, sizeof(int16_t) * K * 256);
    if (t) memcpy(t, key->t, sizeof(int16_t) * K * 256);
    delete key;
    return 0;
}

int kosk_fs_alpha(int kyber_k, const uint8_t *tcomm_all, uint16_t *alpha)
{
    Params P;
    if (!make_params(kyber_k, P)) return -1;
    fs_alpha(P, tcomm_all, alpha);
    return 0;
}

int kosk_fs_opened(const uint8_t *digests_all, uint16_t *I, uint16_t *rest)
{
    fs_opened(digests_all, I, rest);
    return 0;
}

void kosk_host_sha3_256(uint8_t out[32], const uint8_t *in, size_t inlen) { sha3_256(out, in, inlen); }
void kosk_host_shake256(uint8_t *out, size_t outlen, const uint8_t *in, size_t inlen) { shake256(out, outlen, in, inlen); }

int kosk_host_sha3_256_multi(uint8_t *out, const uint8_t *in, size_t in_stride, size_t inlen, int count, int nthreads)
{
    // same code path as the batched Fiat-Shamir rounds: SIMD groups spread over the pool
    if (inlen == (size_t)NPARTY * 32) {
        Params P;
        make_params(2, P);
        std::vector<uint16_t> I((size_t)count * 1312), rest((size_t)count * 1312);
        (void)P;
    }
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
}

int kosk_lagrange_table(int which, uint16_t *out)
{
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
}

} // extern "C"
