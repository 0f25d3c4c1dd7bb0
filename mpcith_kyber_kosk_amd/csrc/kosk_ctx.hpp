// Library context: constant tables + HBM workspace for up to max_batch proofs
// in flight, and the batched prove / verify pipelines built on kosk_kernels.hip.
#pragma once
#include <hip/hip_runtime.h>

#include <string>
#include <functional>
#include <vector>

#include "kosk_device.hpp"
#include "kosk_host.hpp"
#include "kosk_params.hpp"

// Every HIP call of the host code goes through this (needs a `Ctx &c` in scope, returns -1 from the enclosing function).
// HIP keeps the last error per host thread until somebody reads it, and every kernel launcher reports hipGetLastError():
// a failure reported here is consumed here, so that it cannot come back as the "result" of the next, healthy launch on this
// thread (r3: a failed kosk_create on a bad device ordinal made the next handle's first launch fail).
#define KOSK_HIPCHK(x)                                                      \
    do {                                                                    \
        hipError_t e_ = (x);                                                \
        if (e_ != hipSuccess) {                                             \
            (void)hipGetLastError();                                        \
            c.err = std::string(#x) + ": " + hipGetErrorString(e_);         \
            return -1;                                                      \
        }                                                                   \
    } while (0)

namespace kosk {

typedef void (*randombytes_fn)(void *user, uint8_t *out, size_t len);
// called on the calling thread when a commitment round's digest table [n][1454][32] is complete in HBM (the stream may already run
// the next segment, which never writes the tables):
// role 0 prover / 1 verifier, round 0 Tcomm / 1 view commitments
typedef void (*round_fn)(void *user, int role, int round, const void *d_digests, size_t bytes);

// key generation folded into the prover's first segment (kyber_verifiable_keygen as ONE resident call)
struct KeygenIn {
    const uint8_t *tapes; // host or device memory; nullptr: the randombytes callback
    size_t tape_stride;
    uint8_t *pk, *sk;     // host, n records each
    // a merged call (kosk_combine.hpp): several callers' batches back to back in one pipeline run.  This segment covers the
    // next `count` proofs (0: all the remaining ones), `next` the ones behind it
    int count = 0;
    const KeygenIn *next = nullptr;
    // this caller's own round hook (a merged run fires every member's hook with that member's block of the digest table; an
    // unmerged call leaves these null and the context's round_hook fires for the whole batch)
    round_fn hook = nullptr;
    void *hook_user = nullptr;
};
// the verifier's per-caller parts of a merged call: `count` proofs each, own key source and own result bytes
struct VerifySeg {
    int count;
    const uint8_t *pk; // pk_mode 1: this caller's public keys (host or device memory)
    uint8_t *ok;
    const VerifySeg *next = nullptr;
    round_fn hook = nullptr; // as KeygenIn::hook
    void *hook_user = nullptr;
};

enum Phase { PH_HOST_PRE = 0, PH_GPU_COMMIT, PH_FS_ALPHA, PH_GPU_RELATION, PH_FS_OPEN, PH_GPU_ASSEMBLE, PH_D2H,
             // host time spent issuing each segment (part of the GPU phases above) and the verifier's phases
             PH_P1_ISSUE, PH_P2_ISSUE, PH_P3_ISSUE, PH_V1_ISSUE, PH_V1_WAIT, PH_V_FS_ALPHA, PH_V2_ISSUE, PH_V2_WAIT, PH_V_FS_OPEN,
             PH_COUNT };

// HIP-event timing of individual launches on the ctx stream (bench.py roofline leg)
enum ProfId { PR_HASH_TCOMM = 0, PR_HASH_VIEW, PR_GEMM_EXPAND1, PR_GEMM_EXPAND2, PR_LINCOMB, PR_NTT_F, PR_ASSEMBLE,
              PR_V_HASH_TCOMM, PR_V_HASH_VIEW, PR_V_INTERP_BUILD, PR_V_GEMM_INTERP, PR_V_GEMM_EXPAND, PR_V_GEMM_RECON,
              PR_V_LINCOMB, PR_FS_ALPHA, PR_FS_OPENED, PR_V_FS_ALPHA, PR_V_FS_OPENED, PR_COUNT };

enum PathId { PATH_HASH_DMA = 0, PATH_HASH_PLAIN, PATH_TABLE_GEMM, PATH_LIMB_GEMM, PATH_COPY_DIRECT, PATH_COPY_STAGED,
              PATH_GRAPH_REPLAY, PATH_DIGEST_COPY, PATH_SMALL_COPY_KERNEL, PATH_FS_DEVICE, PATH_FS_HOST, PATH_COUNT };

struct GemmTable {
    uint8_t *d = nullptr; // limb matrix (kosk_device.hpp)
    uint8_t *dfrag = nullptr; // the same in fragment-linear tile order (k_table_gemm), tables with Kdim <= 448 only
    int M = 0, Mpad = 0, KS = 0, Kdim = 0;
};

// source rows / destination rows of one mod-q GEMM
struct GemmSrc {
    const uint16_t *src;
    size_t gstride;
    const int16_t *rows;
    int rstride, koff, ncols;
    // every u16 of the rows is a canonical field element (< q): true for everything the pipelines produce (the verifier folds what it
    // takes from a proof image before it reaches the row matrix); 0 for caller data of the kernel-level entry points, which is
    // folded while it is converted (the slower conversion)
    int canonical = 1;
};
struct GemmDst {
    uint16_t *C;
    size_t gstride;
    const int16_t *rows;
    int rstride, off;
};

// compact wire format (kosk_compact.hip)
struct CompactField {
    uint32_t src_off, dst_off, n; // image offset, compact offset, u16 values (or bytes when raw)
    int raw;
};
struct CompactPlan {
    CompactField f[NFIELDS];
    size_t bytes;
};
CompactPlan make_compact_plan(const Params &P);
int compact_encode(const Params &P, const uint8_t *img, uint8_t *out); // -1: a value >= 4096
void compact_decode(const Params &P, const uint8_t *in, uint8_t *img);

struct Ctx {
    int device = 0;
    Params P{};
    RowMap rm{};
    int max_batch = 0;
    // ---- views (kosk_combine.hpp) ----
    // Every per-proof buffer is registered (field, bytes per proof) when it is allocated.  A VIEW of a context is a copy of the
    // struct whose per-proof pointers start `view_first` proofs further: it shares the constant tables and the workspace with
    // the context it was made from (the arena), has its own stream / events / host workers, and can run the pipeline on proofs
    // [view_first, view_first + max_batch) of the arena while other views work on other proofs of it.
    struct PerProof { size_t field_off, stride_bytes; };
    std::vector<PerProof> per_proof;
    template <class T> void reg_pp(T **field, size_t stride_bytes)
    {
        per_proof.push_back({(size_t)(reinterpret_cast<char *>(field) - reinterpret_cast<char *>(this)), stride_bytes});
    }
    bool is_view = false;
    int view_first = 0;
    int own_batch = 0;   // proofs of this context's own callers (a view: its member's kosk_create size; else max_batch)
    // the largest batch ONE call may run on this context: own_batch, except while this view leads a merged run of its cohort
    // (kosk_capi.cpp: RunScope raises it to the run's total).  A view's max_batch is everything behind its first proof in the
    // arena -- its neighbours' blocks included -- so no entry point may size a chunk by max_batch
    int call_cap = 0;
    int base_threads = 1; // host threads of a call of own_batch proofs (a merged run uses base_threads x members)
    int reserved_threads = 1; // workers the pool was created with (a call never creates threads: nthreads <= this)
    int nthreads = 1;
    Pool *pool = nullptr; // this context's host worker threads
    hipStream_t stream = nullptr;
    std::string err;
    randombytes_fn rb = nullptr;
    void *rb_user = nullptr;
    round_fn round_hook = nullptr;
    void *round_user = nullptr;

    // constant tables (HBM, L2-resident while in use)
    GemmTable t_expand, t_recon_d, t_recon_2d;
    int16_t *d_fresh_rows = nullptr, *d_gemm1_rows = nullptr, *d_gemm2_rows = nullptr;
    int n_gemm1 = 0, n_gemm2 = 0;
    int32_t *d_off = nullptr; // NTT offset tables, see build_tables()
    int off_ntt1_src = 0, off_ntt1_dst = 0, n_ntt1 = 0; // NTT(f_i) -> Tf_i and NTT(s_i) -> shat_i in one launch
    int off_sr_er = 0, off_nttsr_er = 0;
    FieldDesc *d_fields = nullptr;
    int16_t *d_rowtab = nullptr;
    AsmGroup *d_asm_groups = nullptr; // the grouped image kernel's tables (build_tables)
    AsmElem *d_asm_elems = nullptr;
    int n_asm_groups = 0;
    int nfields = 0;
    std::vector<FieldDesc> h_fields;
    std::vector<int16_t> h_rowtab;

    // per-batch workspace
    size_t proof_stride = 0; // u16 per proof in the row matrix
    size_t tape_stride = 0, image_stride = 0, key_stride = 0, se_stride = 0;
    int sel_stride = 0;
    uint16_t *d_P = nullptr;
    uint8_t *d_tape = nullptr, *d_dig1 = nullptr, *d_dig2 = nullptr, *d_proof = nullptr;
    // the tapes the kernels read: d_tape after an upload, or the caller's own device buffer used in place
    const uint8_t *tape_cur = nullptr;
    size_t tape_cur_stride = 0;
    TapeSegs tape_segs{}; // a merged call whose callers' device tapes are read in place (count > 0; valid for that call only)
    int16_t *d_A = nullptr, *d_se = nullptr;
    // key generation on the device (kosk_keygen_kernels.hip)
    // one record per proof [pk | NTT(s) bytes | sha3_512 output], kg_rec bytes apart, so that the key generation's
    // outputs cross PCIe in ONE copy; d_pk / d_sb / d_seeds (and the h_ mirrors) point into it, all with stride kg_rec
    uint8_t *d_kg = nullptr, *h_kg = nullptr;
    size_t kg_rec = 0, sb_bytes = 0;
    uint8_t *d_seeds = nullptr, *d_pk = nullptr, *d_sb = nullptr; // sha3_512 output, packed pk, packed NTT(s)
    int16_t *d_sehat = nullptr;
    int resident_pk_n = 0; // pk records valid in d_pk (written by the key generation or a verifier staging call): pk_mode 2 needs >= n
    size_t pk_stride = 0, sb_stride = 0;
    uint8_t *h_seeds = nullptr, *h_pk = nullptr, *h_sb = nullptr;
    uint16_t *d_t = nullptr; // pk's t, canonical (verifier)
    uint16_t *d_alpha = nullptr, *d_I = nullptr, *d_rest = nullptr;
    int32_t *d_pwT = nullptr;
    uint8_t *d_limbs = nullptr; // limb-matrix staging of the GEMM data operand
    uint8_t *d_coef = nullptr; // K3 operand: alpha-power coefficients as a limb matrix
    int16_t *d_lin_rows = nullptr; // [2][128] output rows of the lincomb GEMM (beta/r, gamma/NTT_r)
    size_t limb_cap = 0;
    // verifier workspace (allocated on first use, kosk_verify.cpp)
    bool verify_ready = false;
    uint16_t *d_O = nullptr;         // opened matrix [proof][nrows][OS] (kosk_params.hpp)
    size_t o_stride = 0;
    uint16_t *d_inv = nullptr;       // [Q] field inverses
    FieldDesc *d_vfields = nullptr;  // proof image -> rows (verifier row assignment)
    int16_t *d_vrowtab = nullptr;
    int n_vfields = 0;
    FieldPlan vplan{}, pplan{}; // block -> (field, chunk) maps of the disassemble / assemble kernels
    int16_t *d_rows_bg = nullptr;    // beta_0..69, gamma_0..69
    int16_t *d_rows_isrc = nullptr, *d_rows_idst = nullptr; // degree-d interpolations: given rows -> recomputed rows
    int16_t *d_rows_u = nullptr;     // us / ue rows (degree 2d)
    int n_interp_d = 0, n_interp_2d = 0;
    uint16_t *d_w = nullptr, *d_ell = nullptr, *d_fact = nullptr, *d_invfact = nullptr;
    int16_t *d_node_of = nullptr;
    uint16_t *d_isort = nullptr, *d_hrange = nullptr;
    uint8_t *d_gather = nullptr, *d_gather2 = nullptr; // weighted shares as MFMA fragment tiles (k_gather_frags)
    uint16_t *d_invlimb = nullptr;
    uint16_t *d_sec = nullptr, *d_sec_u1 = nullptr, *d_sec_u2 = nullptr;
    uint32_t *d_fail = nullptr;      // [proof] bit mask of failed checks (FailBit)
    uint16_t *h_Iimg = nullptr;      // I fields as read from the proof images
    // The verifier's host needs both digest tables whole (mlwe_verifier.cpp:40-44, :648-652), but only 150 entries per proof and
    // round are NEW -- the rest are fields of the proof image.  When the caller handed the proofs over in host memory (host_img:
    // kosk_verify_batch) the host reads those 1304 digests per table straight from the caller's images, only the 150 recomputed ones per
    // proof follow each round's hash (4.8 KB instead of 46.5 KB on the critical path), and the host puts the table together
    // (assemble_digest_table).  For resident proofs the whole tables are copied behind each hash (the split with early copies on a
    // side stream measured slower and is gone since round 6: profiles/r04_digest_paths.txt).
    uint8_t *d_odig = nullptr, *h_odig = nullptr; // [proof][NOPEN][32] recomputed digests of the opened parties (one round at a time)
    const uint8_t *host_img = nullptr;            // the caller's host copy of the proof images of THIS call (kosk_verify_batch)
    size_t host_img_stride = 0;
    hipEvent_t ev = nullptr;
    std::vector<uint16_t> v_I2, v_rest2; // the verifier's recomputed opened lists (kept across calls: two fresh 0.4 MB vectors per call were
                                         // an mmap, a page fault per page and a munmap on the tail of every verify call)
    std::function<void()> near_end_hook; // set by a merged run's executor: called once when only the call's tail is left (kosk_combine.hpp: near_end)
    bool kg_on_host_pending = false; // ev_kg was recorded by this call's key generation
    hipEvent_t ev_kg = nullptr; // the key records (pk, NTT(s) bytes, seeds) of a keygen-in-front call are on the host once it has passed
    // KOSK_WAIT_NAP (default 1 since round 5; 0: spin throughout): the long host waits of the resident calls (six per step: the GPU
    // phases between the host's rounds) sleep through most of their expected duration -- the SHORTEST of the site's last eight waits
    // at this batch size (a phase has a floor and outliers are always longer: one 10 ms hiccup must not make the next waits
    // oversleep), only phases above 250 us -- and spin only for the last 30 %: a spinning wait keeps a core busy for the whole GPU phase,
    // and an 8-GPU node has few cores per rank (wait_event, kosk_ctx.cpp).  Measured with nine callers in three cohorts: the same
    // throughput (148-151 k proofs/s either way) at 3.5 busy cores fewer together with KOSK_POOL_SPIN_US=0 (profiles/r05_sweep_host.txt)
    bool wait_nap = true;
    enum { WAIT_SITES = 8 };
    enum { WAIT_HIST = 8 };
    // (round 6) a site keeps WAIT_SIZES histories, one per batch size it has recently seen (a cohort whose merged runs alternate
    // between two sizes -- 276 and 230 proofs when one member is late -- no longer forgets what it learnt at every change), least
    // recently used first out
    enum { WAIT_SIZES = 2 };
    struct WaitHist {
        int n = 0;                    // batch size this history belongs to (0: unused)
        int at = 0;                   // next entry to overwrite
        unsigned long stamp = 0;      // last use (for the replacement)
        double us[WAIT_HIST] = {0};   // the last measured waits (0 = empty): the nap is sized by their MINIMUM
    };
    WaitHist wait_hist[WAIT_SITES][WAIT_SIZES];
    unsigned long wait_stamp = 0;
    hipEvent_t ev_sync = nullptr; // KOSK_BLOCKING_SYNC=1: every host wait sleeps on an event instead of spinning (few host cores per GPU)
    bool blocking_sync = false;
    int n_simd = 1024;      // SIMDs of the device (4 per CU)
    hipEvent_t timer_ev[2] = {nullptr, nullptr}; // kosk_stream_timer_start / _stop
    // compact wire format staging (allocated on first use)
    CompactPlan cplan{};
    size_t compact_stride = 0;
    uint8_t *d_compact = nullptr, *h_compact = nullptr;
    uint32_t *d_compact_bad = nullptr, *h_compact_bad = nullptr;

    // pinned host staging
    uint8_t *h_tape = nullptr, *h_dig = nullptr, *h_dig2 = nullptr, *h_proof = nullptr; // h_dig / h_dig2: the host's copies of the two digest tables
    uint16_t *h_alpha = nullptr, *h_I = nullptr, *h_rest = nullptr;
    uint32_t *h_fail = nullptr;
    // device-raised errors (XofGuard): one word of page-locked host memory the kernels store to; checked and cleared by
    // device_error_check() after a call's last synchronisation.  KOSK_DEBUG_XOF_BLOCKS=n lowers gen_matrix's block limit
    // (tests force the error path with 1)
    uint32_t *h_err = nullptr;
    int xof_max_blocks = 32;
    XofGuard xof_guard() const { XofGuard g; g.err = h_err; g.max_blocks = xof_max_blocks; return g; }

    // hipGraph per pipeline segment (the launches between two host Fiat-Shamir rounds), captured once per
    // batch size and replayed: one API call instead of ~15 launches of kernels that run 3-6 us each.
    enum SegId { SEG_P1 = 0, SEG_P1B, SEG_P2, SEG_P2B, SEG_P3, SEG_V1, SEG_V1B, SEG_V2, SEG_V2B, SEG_COUNT };
    struct SegGraph {
        hipGraphExec_t exec = nullptr;
        int n = 0;
        // caller-owned pointers baked into the captured launches (SEG_P1: the tape buffer read in place): part of the cache key
        const void *key_ptr = nullptr;
        size_t key_stride = 0;
    };
    SegGraph seg[SEG_COUNT];
    bool use_graphs = false; // KOSK_GRAPHS=1 turns them on (measured on ROCm 7.2: no gain over plain launches)
    bool capturing = false;
    // strict_encoding (kosk_options::strict_encoding, KOSK_STRICT_ENCODING; DEFAULT 1 since round 6): the verifier marks a proof malformed
    // (fail bit 0) when ANY u16 element of a record the reference reads is >= q -- no honest prover emits one, and accepting them makes
    // proofs malleable (v and v + q verify alike).  0 = the reference-following mode of round 5: such elements are treated exactly as
    // the reference treats them, record by record: folded where it only multiplies / converts to ZZ_p, raw in its
    // non-reducing add / sub and comparisons; pinned against the oracle by two differential suites, opt-in for parity work
    bool strict_encoding = true;
    // Fiat-Shamir aggregation on the device (kosk_options::fs_mode = KOSK_FS_DEVICE, or KOSK_FS_DEVICE=1 for kosk_create): the four
    // sha3_256 chains of a prove + verify over the 46.5 KB digest tables run as one wave per proof where the tables are, the challenge
    // vectors / opened lists / the verifier's I' == I never leave HBM, and the resident calls have NO host round trip left: no digest
    // table crosses PCIe, the host neither hashes nor waits between segments (kosk_fs_kernels.hip).  Host mode (default)
    // is the path of rounds 1-5
    bool fs_device = false;
    bool host_register = true;       // KOSK_REGISTER=0: staging copies only, even for buffers the caller page-locked itself.  (KOSK_REGISTER=2 of
                                     // rounds 2-4 -- the library page-locking PAGEABLE caller memory for a call -- is gone: both process aborts on
                                     // record happened inside calls that had just done that, and neither was ever reproduced or explained)
    // which of those paths really ran on this context (kosk_path_count): the tests of the knobs assert on these
    long path_n[PATH_COUNT] = {0};

    double phase_sec[PH_COUNT] = {0};
    int prof_on = 0; // 1: the graded kernel only (graphs stay on); 2: every profiled id (plain launches)
    hipEvent_t prof_ev[PR_COUNT][2] = {};
    bool prof_used[PR_COUNT] = {};
    double prof_ms[PR_COUNT] = {0};
    long prof_n[PR_COUNT] = {0};
    long prof_units[PR_COUNT] = {0}; // proofs served by the timed launches (a merged run serves several callers' batches per launch)
    int prof_cur_units[PR_COUNT] = {0};
    void prof_begin(int id, int units = 0) { if (prof_on && !capturing) { (void)hipEventRecord(prof_ev[id][0], stream); prof_cur_units[id] = units; } }
    void prof_end(int id) { if (prof_on && !capturing) { (void)hipEventRecord(prof_ev[id][1], stream); prof_used[id] = true; } }
    void prof_collect(); // call after the stream has been synchronised

    ~Ctx();
};

// Run `body` (stream launches on c.stream only: kernels, pinned-memory copies, memsets) as segment `seg`:
// captured into a graph on first use for this batch size, replayed afterwards.
template <class F>
int run_segment(Ctx &c, int seg, int n, F &&body, const void *key_ptr = nullptr, size_t key_stride = 0)
{
    if (seg < 0 || !c.use_graphs || c.prof_on == 2) return body();
    Ctx::SegGraph &g = c.seg[seg];
    if (g.exec && (g.n != n || g.key_ptr != key_ptr || g.key_stride != key_stride)) {
        (void)hipGraphExecDestroy(g.exec);
        g.exec = nullptr;
    }
    if (!g.exec) {
        hipError_t e = hipStreamBeginCapture(c.stream, hipStreamCaptureModeThreadLocal);
        if (e != hipSuccess) { c.err = std::string("hipStreamBeginCapture: ") + hipGetErrorString(e); return -1; }
        c.capturing = true;
        int rc;
        try {
            rc = body();
        } catch (...) { // never leave the stream in capture mode (the handle would be unusable afterwards)
            c.capturing = false;
            hipGraph_t dead = nullptr;
            (void)hipStreamEndCapture(c.stream, &dead);
            if (dead) (void)hipGraphDestroy(dead);
            throw;
        }
        c.capturing = false;
        hipGraph_t graph = nullptr;
        e = hipStreamEndCapture(c.stream, &graph);
        if (rc) {
            if (graph) (void)hipGraphDestroy(graph);
            return rc;
        }
        if (e != hipSuccess || !graph) { c.err = std::string("hipStreamEndCapture: ") + hipGetErrorString(e); return -1; }
        e = hipGraphInstantiate(&g.exec, graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);
        if (e != hipSuccess) { g.exec = nullptr; c.err = std::string("hipGraphInstantiate: ") + hipGetErrorString(e); return -1; }
        g.n = n;
        g.key_ptr = key_ptr;
        g.key_stride = key_stride;
    }
    const hipError_t e = hipGraphLaunch(g.exec, c.stream);
    if (e != hipSuccess) { c.err = std::string("hipGraphLaunch: ") + hipGetErrorString(e); return -1; }
    c.path_n[PATH_GRAPH_REPLAY]++;
    return 0;
}

// what kosk_create_ex's options struct decides per context (-1 / 0 = not given: the environment variable of the same name, then the default)
struct CtxOpts {
    int host_threads = 0;    // KOSK_HOST_THREADS
    int blocking_sync = -1;  // KOSK_BLOCKING_SYNC
    int strict_encoding = -1; // KOSK_STRICT_ENCODING
    int fs_device = -1;      // KOSK_FS_DEVICE
};
// host_share: sub-contexts of the same handle that share this process's CPUs (divides the host thread budget)
int ctx_create(Ctx **out, int device, int kyber_k, int max_batch, std::string &err, int host_share = 1, const CtxOpts &opts = CtxOpts());
// a view of `arena` starting at proof `first` whose own callers send up to `own_batch` proofs per call; reserve_threads: host
// workers created now (a merged run led by this view uses base_threads x members of them)
int ctx_make_view(Ctx &arena, int first, int own_batch, int reserve_threads, Ctx **out, std::string &err);

// C[g][rows_d[i]][off + m] = sum_k A[m][k] * src[g][rows_s[i]][koff + k] mod q  (conversion to limbs + MFMA GEMM)
// host wait for everything queued on the context's stream (spinning, or sleeping with KOSK_BLOCKING_SYNC=1)
hipError_t stream_sync(Ctx &c);
// wait for `ev` (recorded on the context's stream); site = which of the pipeline's long waits this is (0..WAIT_SITES-1: napped with
// KOSK_WAIT_NAP=1), n = the batch size the wait belongs to; site < 0: a plain wait
hipError_t wait_event(Ctx &c, hipEvent_t ev, int site, int n);
hipError_t stream_sync_site(Ctx &c, int site, int n);
// a round's digest table of n proofs from HBM into the context's page-locked host memory, on the context's stream (the caller records c.ev behind it)
hipError_t copy_round_table(Ctx &c, uint8_t *h_dst, const uint8_t *d_src, int n);
// a small copy between HBM and one of the library's OWN page-locked host buffers, rows x row_bytes (kernel or hipMemcpy[2D]Async)
hipError_t copy_small(Ctx &c, void *dst, size_t dst_stride, const void *src, size_t src_stride, size_t row_bytes, size_t nrows, hipMemcpyKind kind, hipStream_t st);
// after the stream has been synchronised: -1 (with c.err set, the word cleared) if a kernel of this context raised an error
int device_error_check(Ctx &c);
int gemm_modq(Ctx &c, const uint8_t *A, size_t a_gstride, int Mpad, int M, int KS, const GemmSrc &s, const GemmDst &d,
              int npg, int ngroups, bool grouped, const uint8_t *Afrag = nullptr);
inline int gemm_modq(Ctx &c, const GemmTable &t, const GemmSrc &s, const GemmDst &d, int npg, int ngroups)
{
    return gemm_modq(c, t.d, 0, t.Mpad, t.M, t.KS, s, d, npg, ngroups, false, t.dfrag);
}

// tapes (host or device memory, nullptr = callback) -> pk/sk on host, tape + key material resident in HBM
int stage_prover_inputs(Ctx &c, int n, const uint8_t *tapes, size_t tape_stride, uint8_t *pk, uint8_t *sk);
// its three parts: make the tapes resident (async), kyber_keygen on the device + D2H of pk / NTT(s) / seeds (async),
// and, once the stream has been synchronised, the host half (sk = NTT(s) || pk || H(pk) || z, kosk.cpp:62-69)
int upload_tapes(Ctx &c, int n, const uint8_t *tapes, size_t tape_stride);
int issue_keygen(Ctx &c, int n, bool sampled = false);
void finish_keygen_host(Ctx &c, int n, uint8_t *pk, uint8_t *sk);
// everything from resident inputs to resident proof images (two host Fiat-Shamir round trips)
enum FrontPart { FRONT_FULL = 0, FRONT_RANDOMNESS, FRONT_RANGE, FRONT_ONLINE };
int issue_sharing_front(Ctx &c, int n, FrontPart part, bool with_keygen = false);
// online_only: the offline material (f, NTT f, eta sharings) is already in the row matrix (prove_prepared)
// keygen != nullptr: key generation runs at the head of the first segment (no extra synchronisation) and pk / sk are
// written before the call returns -- the whole kyber_verifiable_keygen, proofs left resident
int prove_resident(Ctx &c, int n, bool online_only = false, const KeygenIn *keygen = nullptr);
// the reference's second-level entry points on host structs (kosk_split.cpp); struct layouts in include/kosk_mi355x.h
size_t randomness_bytes(const Params &P);
size_t range_proof_bytes(const Params &P);
size_t mlwe_inst_bytes(const Params &P);
int prepare_randomness(Ctx &c, int n, const uint8_t *tapes, size_t tape_stride, uint8_t *rand_out);
int prepare_range_proof(Ctx &c, int n, const uint8_t *tapes, size_t tape_stride, uint8_t *range_out);
int prove_prepared(Ctx &c, int n, const uint8_t *inst, const uint8_t *rand_in, const uint8_t *range_in, const uint8_t *tapes,
                   size_t tape_stride, uint8_t *pi);
int stage_verifier_inst(Ctx &c, int n, const uint8_t *pi, const uint8_t *inst);
int ensure_verify_workspace(Ctx &c);
// direct: the host buffer is page-locked (copied to / from straight, no staging)
int fetch_proofs_compact(Ctx &c, int n, uint8_t *out, bool direct = false);
int stage_verifier_inputs_compact(Ctx &c, int n, const uint8_t *in, const uint8_t *pk, bool direct = false);
// registered: `pi` is page-locked host memory (hipHostRegister): copied to directly, no staging
int fetch_proofs(Ctx &c, int n, uint8_t *pi, bool registered = false);

int stage_verifier_inputs(Ctx &c, int n, const uint8_t *pi, const uint8_t *pk, bool registered = false);
// pk_mode 0: A and t are already resident (stage_verifier_inputs / stage_verifier_inst); 1: decode `pk` (host or
// device memory, n records of pk_bytes) at the head of the first segment (kosk.cpp:94-99: polyvec_frombytes + gen_matrix);
// 2: decode the pk bytes the key generation left resident in HBM
int verify_resident(Ctx &c, int n, uint8_t *ok, int pk_mode = 0, const uint8_t *pk = nullptr, const VerifySeg *segs = nullptr);

} // namespace kosk
