// Context construction and the batched prover pipeline.
//
// Stage map (reference prove(), mlwe_prover.cpp:81-538):
//   stage_prover_inputs : tape upload + host keygen (kosk.cpp:4-70)
//   prove_resident      : [GPU] prepare_randomness / prepare_range_proof / P1 as
//                         ONE Lagrange-expansion GEMM over every fresh sharing,
//                         gates P13-P15, Tcomm P3 -> [host] alpha P4 -> [GPU] P5-P12
//                         -> view hash P16 -> [host] I P17 -> [GPU] wire image P18.
// All linear steps act on whole evaluation-point rows (secrets included), so the
// values the reference obtains by recon_secrets_ddeg/2ddeg are simply the x < 256
// part of the same rows.
#include "kosk_ctx.hpp"

#include <atomic>

#include <chrono>
#include <cstdio>
#include <cstring>
#include <ctime>
#include <thread>

#include "kosk_math.hpp"

namespace kosk {

#define HIPCHK(x) KOSK_HIPCHK(x)

static double now_sec()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

template <typename T>
static hipError_t dalloc(T **p, size_t n)
{
    return hipMalloc(reinterpret_cast<void **>(p), n * sizeof(T));
}
template <typename T>
static hipError_t halloc(T **p, size_t n)
{
    return hipHostMalloc(reinterpret_cast<void **>(p), n * sizeof(T), hipHostMallocDefault);
}

void Ctx::prof_collect()
{
    if (!prof_on) return;
    for (int i = 0; i < PR_COUNT; i++)
        if (prof_used[i]) {
            float ms = 0;
            if (hipEventElapsedTime(&ms, prof_ev[i][0], prof_ev[i][1]) == hipSuccess) { prof_ms[i] += ms; prof_n[i]++; prof_units[i] += prof_cur_units[i]; }
            prof_used[i] = false;
        }
}

Ctx::~Ctx()
{
    for (auto &g : seg)
        if (g.exec) (void)hipGraphExecDestroy(g.exec);
    for (auto &pe : prof_ev)
        for (auto e : pe)
            if (e) (void)hipEventDestroy(e);
    if (ev_sync) (void)hipEventDestroy(ev_sync);
    if (is_view) {
        if (h_err) (void)hipHostFree(h_err);
        // a view owns its events, host workers and compact staging; tables, workspace and the stream belong to the arena
        if (d_compact) (void)hipFree(d_compact);
        if (d_compact_bad) (void)hipFree(d_compact_bad);
        if (h_compact) (void)hipHostFree(h_compact);
        if (h_compact_bad) (void)hipHostFree(h_compact_bad);
        if (pool) pool_destroy(pool);
        if (ev) (void)hipEventDestroy(ev);
        if (ev_kg) (void)hipEventDestroy(ev_kg);
        for (auto e : timer_ev)
            if (e) (void)hipEventDestroy(e);
        return; // the stream is the arena's
    }
    void *dev[] = {t_expand.d, t_recon_d.d, t_recon_2d.d, t_expand.dfrag, t_recon_d.dfrag, t_recon_2d.dfrag, d_fresh_rows, d_gemm1_rows, d_gemm2_rows, d_off, d_fields, d_asm_groups, d_asm_elems,
                   d_rowtab, d_P, d_tape, d_dig1, d_dig2, d_proof, d_A, d_se, d_kg, d_sehat, d_t, d_alpha, d_I, d_pwT, d_limbs, d_coef, d_lin_rows,
                   d_gather, d_gather2, d_O, d_w, d_ell, d_sec, d_sec_u1, d_sec_u2, d_fail, d_inv, d_invlimb, d_vfields,
                   d_vrowtab, d_rows_bg, d_rows_isrc, d_rows_idst, d_rows_u, d_fact, d_invfact, d_node_of, d_isort, d_hrange, d_odig};
    for (void *p : dev)
        if (p) (void)hipFree(p);
    void *host[] = {h_err, h_tape, h_dig, h_dig2, h_proof, h_alpha, h_I, h_fail, h_Iimg, h_kg, h_odig};
    for (void *p : host)
        if (p) (void)hipHostFree(p);
    if (d_compact) (void)hipFree(d_compact);
    if (d_compact_bad) (void)hipFree(d_compact_bad);
    if (h_compact) (void)hipHostFree(h_compact);
    if (h_compact_bad) (void)hipHostFree(h_compact_bad);
    if (pool) pool_destroy(pool);
    if (ev) (void)hipEventDestroy(ev);
    if (ev_kg) (void)hipEventDestroy(ev_kg);
    for (auto e : timer_ev)
        if (e) (void)hipEventDestroy(e);
    if (stream) (void)hipStreamDestroy(stream);
}

static int upload_table(Ctx &c, GemmTable &t, const std::vector<uint16_t> &A, int M, int Kdim)
{
    t.M = M;
    t.Kdim = Kdim;
    t.Mpad = (M + 127) / 128 * 128;
    t.KS = (Kdim + 63) / 64;
    std::vector<uint8_t> pk;
    pack_limb_table(A, M, Kdim, t.Mpad, t.KS, pk);
    HIPCHK(dalloc(&t.d, pk.size()));
    HIPCHK(hipMemcpyAsync(t.d, pk.data(), pk.size(), hipMemcpyHostToDevice, c.stream)); // c.stream, not the legacy null stream
    HIPCHK(hipStreamSynchronize(c.stream));
    if (t.KS == 7 || t.KS == 13) {
        pack_frag_table(A, M, Kdim, t.Mpad, t.KS, pk);
        HIPCHK(dalloc(&t.dfrag, pk.size()));
        HIPCHK(hipMemcpyAsync(t.dfrag, pk.data(), pk.size(), hipMemcpyHostToDevice, c.stream));
        HIPCHK(hipStreamSynchronize(c.stream));
    }
    return 0;
}

// one commitment round of n proofs: a single launch, timed under the round's profile id
static hipError_t commit_hash_batch(Ctx &c, const HashArgs &ha, int n, int K, bool view, hipStream_t st)
{
    c.prof_begin(view ? PR_HASH_VIEW : PR_HASH_TCOMM, n);
    int variant = 0;
    const hipError_t e = launch_commit_hash(ha, n, K, view, st, &variant);
    if (!c.capturing) c.path_n[(variant & 1) ? PATH_HASH_DMA : PATH_HASH_PLAIN]++; // a captured launch runs at replay time (PATH_GRAPH_REPLAY counts those)
    c.prof_end(view ? PR_HASH_VIEW : PR_HASH_TCOMM);
    return e;
}

int device_error_check(Ctx &c)
{
    if (!c.h_err) return 0;
    const uint32_t e = *reinterpret_cast<volatile uint32_t *>(c.h_err);
    if (!e) return 0;
    *reinterpret_cast<volatile uint32_t *>(c.h_err) = 0;
    c.err = (e & DEVERR_XOF_BLOCKS) ? "gen_matrix: SHAKE128 block limit reached before 256 coefficients were accepted (indcpa.c:124-145 would squeeze on; "
                                      "probability < 2^-300 with the default limit of 32 blocks): no result was produced"
                                    : "a kernel reported an internal error";
    return -1;
}

hipError_t copy_round_table(Ctx &c, uint8_t *h_dst, const uint8_t *d_src, int n)
{
    // the runtime's copy: every kernel-driven store into host memory measured slower for the pipeline as a whole, and so did the table in
    // pieces with an event behind each (profiles/r04_copy_kernel.txt, r05_copy_kernel.txt, r05_table_chunks.txt; both removed in round 6)
    return hipMemcpyAsync(h_dst, d_src, (size_t)n * NPARTY * 32, hipMemcpyDeviceToHost, c.stream);
}

hipError_t copy_small(Ctx &c, void *dst, size_t dst_stride, const void *src, size_t src_stride, size_t row_bytes, size_t nrows, hipMemcpyKind kind, hipStream_t st)
{
    if (!row_bytes || !nrows) return hipSuccess;
    if (copy_small_ok(src, src_stride, dst, dst_stride, row_bytes)) {
        if (!c.capturing) c.path_n[PATH_SMALL_COPY_KERNEL]++;
        return launch_copy_small(src, src_stride, dst, dst_stride, row_bytes, nrows, st);
    }
    if (nrows == 1 || (src_stride == row_bytes && dst_stride == row_bytes)) return hipMemcpyAsync(dst, src, row_bytes * nrows, kind, st);
    return hipMemcpy2DAsync(dst, dst_stride, src, src_stride, row_bytes, nrows, kind, st);
}

hipError_t stream_sync(Ctx &c)
{
    if (!c.blocking_sync) return hipStreamSynchronize(c.stream);
    const hipError_t e = hipEventRecord(c.ev_sync, c.stream);
    return e != hipSuccess ? e : hipEventSynchronize(c.ev_sync);
}

hipError_t wait_event(Ctx &c, hipEvent_t ev, int site, int n)
{
    if (c.blocking_sync || !c.wait_nap || site < 0 || site >= Ctx::WAIT_SITES) return hipEventSynchronize(ev);
    // this site's history at this batch size (or the least recently used one, started afresh)
    Ctx::WaitHist *h = nullptr, *lru = &c.wait_hist[site][0];
    for (auto &w : c.wait_hist[site]) {
        if (w.n == n) h = &w;
        if (w.stamp < lru->stamp) lru = &w;
    }
    if (!h) {
        h = lru;
        *h = Ctx::WaitHist{};
        h->n = n;
    }
    h->stamp = ++c.wait_stamp;
    int filled = 0;
    double est = 0;
    for (int i = 0; i < Ctx::WAIT_HIST; i++)
        if (h->us[i] > 0) { est = filled ? (h->us[i] < est ? h->us[i] : est) : h->us[i]; filled++; }
    const double t0 = now_sec();
    bool overslept = false;
    if (filled >= 4 && est > 250.0) { // (no nap before the site has been seen four times at this batch size)
        const double keep = est * 0.3 > 100.0 ? est * 0.3 : 100.0; // spin through the last 30 % (at least 100 us: the sleep's own wake-up jitter)
        double nap_us = est - keep;
        if (nap_us > 200000.0) nap_us = 200000.0; // never more than 0.2 s at a time (tv_nsec must stay below 1e9; a phase that long is an anomaly)
        struct timespec ts;
        ts.tv_sec = 0;
        ts.tv_nsec = (long)(nap_us * 1e3);
        nanosleep(&ts, nullptr);
        const hipError_t q = hipEventQuery(ev);
        if (q == hipSuccess) overslept = true;
        else if (q != hipErrorNotReady) return q;
    }
    const hipError_t e = overslept ? hipSuccess : hipEventSynchronize(ev);
    const double took = (now_sec() - t0) * 1e6;
    if (overslept) {
        // the phase was shorter than the nap: what the site remembers is too long (a history seeded by cold-start waits of several ms would
        // otherwise oversleep call after call while it halves).  Forget it: the next four waits spin and measure afresh
        for (int i = 0; i < Ctx::WAIT_HIST; i++) h->us[i] = 0;
        h->at = 0;
    } else {
        h->us[h->at] = took > 1.0 ? took : 1.0;
        h->at = (h->at + 1) % Ctx::WAIT_HIST;
    }
    return e;
}
hipError_t stream_sync_site(Ctx &c, int site, int n)
{
    if (c.blocking_sync || !c.wait_nap) return stream_sync(c);
    const hipError_t e = hipEventRecord(c.ev_sync, c.stream);
    return e != hipSuccess ? e : wait_event(c, c.ev_sync, site, n);
}

int gemm_modq(Ctx &c, const uint8_t *A, size_t a_gstride, int Mpad, int M, int KS, const GemmSrc &s, const GemmDst &d,
              int npg, int ngroups, bool grouped, const uint8_t *Afrag)
{
    if (npg <= 0 || ngroups <= 0) return 0;
    GemmArgs ga{};
    ga.Afrag = Afrag;
    ga.A = A; ga.a_gstride = a_gstride; ga.Mpad = Mpad; ga.M = M; ga.KS = KS;
    ga.C = d.C; ga.c_gstride = d.gstride; ga.c_rows = d.rows; ga.c_rstride = d.rstride; ga.c_off = d.off;
    ga.npg = npg; ga.npg_pad = grouped ? (npg + 63) / 64 * 64 : npg; ga.ngroups = ngroups; ga.grouped = grouped ? 1 : 0;
    if (!grouped && Afrag) {
        // shared table, 407-wide inputs: data rows resident in LDS, no limb matrix in HBM (k_table_gemm)
        GemmArgs ta = ga;
        ta.B = nullptr; ta.BRT = 0;
        ta.src = s.src; ta.src_gstride = s.gstride; ta.src_rows = s.rows; ta.src_rstride = s.rstride; ta.src_koff = s.koff;
        ta.src_canonical = s.canonical;
        if (table_gemm_usable(ta)) {
            HIPCHK(launch_table_gemm(ta, reinterpret_cast<uint16_t *>(c.d_limbs), c.stream));
            if (!c.capturing) c.path_n[PATH_TABLE_GEMM]++;
            return 0;
        }
    }
    const int rows = ((grouped ? ga.npg_pad * ngroups : npg * ngroups) + 63) / 64 * 64;
    const int mtiles = Mpad / 128;
    if (mtiles >= 4 && rows >= 2048 && (size_t)(rows / 16) * KS * 2048 <= c.limb_cap) {
        // large product: every data row is used by many table tiles -> convert it to limbs once, in its own pass
        LimbArgs la{};
        la.src = s.src; la.src_gstride = s.gstride; la.rows = s.rows; la.src_rstride = s.rstride; la.src_koff = s.koff;
        la.ncols = s.ncols; la.KS = KS; la.dst = c.d_limbs; la.RT = rows / 16; la.npg = npg; la.npg_pad = ga.npg_pad; la.ngroups = ngroups;
        HIPCHK(launch_rows_to_limbs(la, c.stream));
        ga.B = c.d_limbs; ga.BRT = rows / 16;
    } else {
        // small product (latency-bound): convert inside the GEMM's own staging and save a launch
        ga.B = nullptr; ga.BRT = 0;
        ga.src = s.src; ga.src_gstride = s.gstride; ga.src_rows = s.rows; ga.src_rstride = s.rstride; ga.src_koff = s.koff;
    }
    HIPCHK(launch_gemm(ga, c.stream));
    if (!c.capturing) c.path_n[PATH_LIMB_GEMM]++;
    return 0;
}

template <typename T>
static int upload_vec(Ctx &c, T **d, const std::vector<T> &v)
{
    HIPCHK(dalloc(d, v.size() ? v.size() : 1));
    if (!v.empty()) {
        HIPCHK(hipMemcpyAsync(*d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, c.stream)); // c.stream, not the legacy null stream
        HIPCHK(hipStreamSynchronize(c.stream));
    }
    return 0;
}

static int build_tables(Ctx &c)
{
    const Params &P = c.P;
    const RowMap &rm = c.rm;
    const int K = P.K, M = P.M, E = P.E, Z = P.Z;

    { // Lagrange expansion operator for the points EXP_OFF..RS-1 from the 407 values at 0..406
        std::vector<uint16_t> A((size_t)EXP_M * XLEN, 0);
        for (int m = 0; m < EXP_M; m++) {
            const int x = EXP_OFF + m;
            if (x < XLEN) A[(size_t)m * XLEN + x] = 1;             // already an input: identity
            else if (x < NPTS) lagrange_row(&A[(size_t)m * XLEN], XLEN, 0, x); // ss.cpp:26-27
        }
        if (upload_table(c, c.t_expand, A, EXP_M, XLEN)) return -1;
    }
    { // reconstruction of the packed secrets from parties 0..406 / 0..812   ss.cpp:47, :66
        std::vector<uint16_t> A((size_t)NSEC * XLEN), A2((size_t)NSEC * (DEG2 + 1));
        for (int i = 0; i < NSEC; i++) {
            lagrange_row(&A[(size_t)i * XLEN], XLEN, NSEC, i);
            lagrange_row(&A2[(size_t)i * (DEG2 + 1)], DEG2 + 1, NSEC, i);
        }
        if (upload_table(c, c.t_recon_d, A, NSEC, XLEN)) return -1;
        if (upload_table(c, c.t_recon_2d, A2, NSEC, DEG2 + 1)) return -1;
    }

    // fresh sharings in randomness-tape order
    std::vector<int16_t> fresh;
    for (int i = 0; i < M; i++) { fresh.push_back((int16_t)(rm.f + i)); fresh.push_back((int16_t)(rm.tf + i)); }   // mlwe_prover.cpp:29-38
    for (int i = 0; i < K; i++)
        for (int j = 0; j < E; j++) { fresh.push_back((int16_t)(rm.seta + i * E + j)); fresh.push_back((int16_t)(rm.eeta + i * E + j)); } // :51-58
    for (int i = 0; i < K; i++) { fresh.push_back((int16_t)(rm.s + i)); fresh.push_back((int16_t)(rm.e + i)); }    // :98-101
    for (int i = 0; i < K; i++) fresh.push_back((int16_t)(rm.nttas + i));                                         // :316
    for (int i = 0; i < K; i++)
        for (int j = 0; j < Z; j++) { fresh.push_back((int16_t)rm.zs(i, j)); fresh.push_back((int16_t)rm.ze(i, j)); } // :369,:371
    if ((int)fresh.size() != P.nfresh) { c.err = "internal: fresh row count"; return -1; }
    if (upload_vec(c, &c.d_fresh_rows, fresh)) return -1;
    if (upload_vec(c, &c.d_gemm1_rows, fresh)) return -1;
    c.n_gemm1 = (int)fresh.size();
    std::vector<int16_t> g2;
    for (int i = 0; i < K; i++) { g2.push_back((int16_t)(rm.nttsr + i)); g2.push_back((int16_t)(rm.ntter + i)); g2.push_back((int16_t)(rm.nttasr + i)); }
    if (upload_vec(c, &c.d_gemm2_rows, g2)) return -1;
    c.n_gemm2 = (int)g2.size();

    { // output rows of the lincomb GEMM: group 0 (f rows) -> beta_j, r_j ; group 1 (NTT f rows) -> gamma_j, NTT_r_j
        std::vector<int16_t> lr(256, 0);
        for (int j = 0; j < P.J; j++) {
            lr[j] = (int16_t)(j < NCHK ? rm.beta(j) : rm.r + (j - NCHK));
            lr[128 + j] = (int16_t)(j < NCHK ? rm.gamma(j) : rm.nttr + (j - NCHK));
        }
        if (upload_vec(c, &c.d_lin_rows, lr)) return -1;
    }

    // NTT source / destination offsets (u16 units inside one proof group)
    std::vector<int32_t> off;
    auto push_rows = [&](int row0, int n) { int b = (int)off.size(); for (int i = 0; i < n; i++) off.push_back((row0 + i) * RS); return b; };
    c.off_ntt1_src = push_rows(rm.f, M);
    push_rows(rm.s, K);
    c.off_ntt1_dst = push_rows(rm.tf, M);
    push_rows(rm.shat, K);
    c.n_ntt1 = M + K;
    c.off_sr_er = push_rows(rm.sr, 2 * K);        // sr rows then er rows (adjacent in the row map)
    c.off_nttsr_er = push_rows(rm.nttsr, 2 * K);  // nttsr rows then ntter rows
    if (rm.er != rm.sr + K || rm.ntter != rm.nttsr + K) { c.err = "internal: row map adjacency"; return -1; }
    if (upload_vec(c, &c.d_off, off)) return -1;

    // wire image fields (mlwe_prover.hpp:57-75): element e of a party's record <- row
    auto add = [&](int fid, int sel, int width, auto rowfn) {
        FieldDesc fd{};
        fd.off = (uint32_t)P.off[fid];
        fd.sel = sel;
        fd.width = width;
        fd.rowtab_off = (int)c.h_rowtab.size();
        for (int e = 0; e < width; e++) c.h_rowtab.push_back((int16_t)rowfn(e));
        c.h_fields.push_back(fd);
    };
    add(F_F, 0, M, [&](int e) { return rm.f + e; });
    add(F_NTTF, 0, M, [&](int e) { return rm.tf + e; });
    add(F_BETA, 1, NCHK, [&](int e) { return rm.beta(e); });
    add(F_GAMMA, 1, NCHK, [&](int e) { return rm.gamma(e); });
    add(F_S, 0, K, [&](int e) { return rm.s + e; });
    add(F_E, 0, K, [&](int e) { return rm.e + e; });
    add(F_T, 1, K, [&](int e) { return rm.t + e; });
    add(F_NTTS, 0, K, [&](int e) { return rm.ntts + e; });
    add(F_NTTE, 0, K, [&](int e) { return rm.ntte + e; });
    add(F_NTTAR, 0, K, [&](int e) { return rm.nttar + e; });
    add(F_NTTAS, 0, K, [&](int e) { return rm.nttas + e; });
    add(F_SR, 1, K, [&](int e) { return rm.sr + e; });
    add(F_ER, 1, K, [&](int e) { return rm.er + e; });
    add(F_SETA, 1, K * E, [&](int e) { return rm.seta + e; });
    add(F_EETA, 1, K * E, [&](int e) { return rm.eeta + e; });
    add(F_SSUB, 0, K * E, [&](int e) { return rm.ssub + e; });
    add(F_ESUB, 0, K * E, [&](int e) { return rm.esub + e; });
    add(F_ZS, 0, K * Z, [&](int e) { return rm.zs(e / Z, e % Z); });
    add(F_ZE, 0, K * Z, [&](int e) { return rm.ze(e / Z, e % Z); });
    add(F_US, 1, K * Z, [&](int e) { return rm.us(e / Z, e % Z); });
    add(F_UE, 1, K * Z, [&](int e) { return rm.ue(e / Z, e % Z); });
    c.nfields = (int)c.h_fields.size();
    c.pplan = make_field_plan(c.h_fields.data(), c.nfields);
    // groups of the grouped image kernel: fields of one kind packed greedily, in declaration order, into groups of <= 80 rows
    // (K = 3: unopened {beta}, {gamma}, {t, s+r, e+r, eta x 2, u x 2}; opened {f}, {NTT f}, {s, e, NTT s, NTT e, NTT Ar, NTT As, s - eta, e - eta, z x 2})
    {
        std::vector<AsmGroup> groups;
        std::vector<AsmElem> elems;
        for (int sel = 1; sel >= 0; sel--) {
            AsmGroup g{};
            auto flush = [&]() {
                if (g.nsub) groups.push_back(g);
                g = AsmGroup{};
            };
            for (int f = 0; f < c.nfields; f++) {
                const FieldDesc &fd = c.h_fields[f];
                if (fd.sel != sel) continue;
                if (g.nsub && (g.nrows + fd.width > 80 || g.nsub == 12)) flush();
                if (!g.nsub) { g.sel = sel; g.rowtab_off = (int)c.h_rowtab.size(); g.elem_off = (int)elems.size(); }
                g.sub_off[g.nsub] = fd.off; g.sub_width[g.nsub] = (int16_t)fd.width; g.sub_col[g.nsub] = (int16_t)g.nrows;
                for (int k = 0; k < fd.width; k++) {
                    c.h_rowtab.push_back(c.h_rowtab[fd.rowtab_off + k]);
                    elems.push_back(AsmElem{(int32_t)(fd.off / 2 + k), (int16_t)fd.width, (int16_t)(64 * g.nrows + k)});
                }
                g.nrows += fd.width;
                g.nsub++;
            }
            flush();
        }
        // a group's row table is read up to 80 entries deep with clamped indices; the element table per lane up to 128
        if ((int)groups.size() > ASM_MAX_GROUPS) { c.err = "internal: image field groups"; return -1; }
        for (int i = 0; i < 128; i++) elems.push_back(AsmElem{0, 1, 0});
        c.n_asm_groups = (int)groups.size();
        if (upload_vec(c, &c.d_asm_groups, groups)) return -1;
        if (upload_vec(c, &c.d_asm_elems, elems)) return -1;
    }
    if (upload_vec(c, &c.d_fields, c.h_fields)) return -1;
    if (upload_vec(c, &c.d_rowtab, c.h_rowtab)) return -1;
    return 0;
}

int ctx_create(Ctx **out, int device, int kyber_k, int max_batch, std::string &err, int host_share, const CtxOpts &opts)
{
    Ctx *cp = new Ctx();
    Ctx &c = *cp;
    auto fail = [&]() { err = c.err; delete cp; return -1; };
    if (!make_params(kyber_k, c.P)) { c.err = "kyber_k must be 2, 3 or 4"; return fail(); }
    if (max_batch < 1) { c.err = "max_batch must be >= 1"; return fail(); }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        (void)hipGetLastError();
        c.err = "no HIP device available: the KOSK path has no CPU fallback";
        return fail();
    }
    if (device < 0 || device >= ndev) {
        c.err = "device " + std::to_string(device) + " out of range: " + std::to_string(ndev) + " HIP device(s) visible";
        return fail();
    }
    c.device = device;
    c.max_batch = max_batch;
    c.rm = make_rowmap(c.P);
    // host threads per context: at most 8, and never more than this process's CPUs (sched_getaffinity) divided by the
    // sub-contexts of the handle; the workers exist before the first call (no thread is ever created inside a batch call)
    const int cpus = host_cpu_count() / (host_share > 0 ? host_share : 1);
    c.nthreads = cpus > 8 ? 8 : (cpus < 1 ? 1 : cpus);
    if (opts.host_threads > 0) c.nthreads = opts.host_threads > 64 ? 64 : opts.host_threads;
    c.pool = pool_create();
    c.nthreads = pool_reserve(c.pool, c.nthreads);
    c.base_threads = c.nthreads;
    c.reserved_threads = c.nthreads;
    c.own_batch = max_batch;
    c.call_cap = max_batch;
    // debug knobs: everything a host decides per handle is in kosk_options, not in the environment
    if (const char *e = getenv("KOSK_GRAPHS")) c.use_graphs = atoi(e) != 0;
    if (const char *e = getenv("KOSK_WAIT_NAP")) c.wait_nap = atoi(e) != 0;
    if (const char *e = getenv("KOSK_REGISTER")) c.host_register = atoi(e) != 0;
    if (const char *e = getenv("KOSK_DEBUG_XOF_BLOCKS")) c.xof_max_blocks = atoi(e) > 0 ? atoi(e) : c.xof_max_blocks;
    // what the caller's options struct decides (kosk_create_ex)
    if (opts.blocking_sync >= 0) c.blocking_sync = opts.blocking_sync != 0;
    if (opts.strict_encoding >= 0) c.strict_encoding = opts.strict_encoding != 0;
    if (opts.fs_device >= 0) c.fs_device = opts.fs_device != 0;

    auto body = [&]() -> int {
        HIPCHK(hipSetDevice(device));
        {
            int cus = 0;
            HIPCHK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device));
            if (cus > 0) c.n_simd = 4 * cus;
        }
        HIPCHK(hipStreamCreateWithFlags(&c.stream, hipStreamNonBlocking));
        HIPCHK(hipEventCreateWithFlags(&c.ev, hipEventDisableTiming | (c.blocking_sync ? hipEventBlockingSync : 0)));
        HIPCHK(hipEventCreateWithFlags(&c.ev_kg, hipEventDisableTiming | (c.blocking_sync ? hipEventBlockingSync : 0)));
        if (c.blocking_sync) HIPCHK(hipEventCreateWithFlags(&c.ev_sync, hipEventDisableTiming | hipEventBlockingSync));
        else if (c.wait_nap) HIPCHK(hipEventCreateWithFlags(&c.ev_sync, hipEventDisableTiming));
        for (auto &pe : c.prof_ev)
            for (auto &e : pe) HIPCHK(hipEventCreate(&e));
        if (build_tables(c)) return -1;
        const Params &P = c.P;
        const size_t B = (size_t)max_batch;
        c.proof_stride = (size_t)c.rm.nrows * RS;
        c.tape_stride = (P.tape_bytes + 63) / 64 * 64;
        c.image_stride = (P.proof_bytes + 63) / 64 * 64;
        c.key_stride = (size_t)P.K * P.K * 256;
        c.se_stride = (size_t)2 * P.K * 256;
        c.sel_stride = 1312;
        // per-proof buffers: allocated for B proofs and registered with their bytes per proof (views, kosk_combine.hpp)
        auto dev = [&](auto **p, size_t per) -> hipError_t {
            const hipError_t e = dalloc(p, B * per);
            if (e == hipSuccess) c.reg_pp(p, per * sizeof(**p));
            return e;
        };
        auto host = [&](auto **p, size_t per) -> hipError_t {
            const hipError_t e = halloc(p, B * per);
            if (e == hipSuccess) c.reg_pp(p, per * sizeof(**p));
            return e;
        };
        HIPCHK(dev(&c.d_P, c.proof_stride));
        HIPCHK(hipMemsetAsync(c.d_P, 0, B * c.proof_stride * 2, c.stream));
        HIPCHK(dev(&c.d_tape, c.tape_stride));
        HIPCHK(dev(&c.d_dig1, (size_t)NPARTY * 32));
        HIPCHK(dev(&c.d_dig2, (size_t)NPARTY * 32));
        HIPCHK(dev(&c.d_proof, c.image_stride));
        HIPCHK(dev(&c.d_A, c.key_stride));
        HIPCHK(dev(&c.d_se, c.se_stride));
        HIPCHK(dev(&c.d_t, (size_t)P.K * 256));
        const size_t pkpad = (P.pk_bytes + 15) / 16 * 16;
        c.sb_bytes = (size_t)384 * P.K;
        c.kg_rec = pkpad + c.sb_bytes + 64;
        c.pk_stride = c.sb_stride = c.kg_rec;
        HIPCHK(dev(&c.d_kg, c.kg_rec));
        HIPCHK(host(&c.h_kg, c.kg_rec));
        c.d_pk = c.d_kg; c.d_sb = c.d_kg + pkpad; c.d_seeds = c.d_kg + pkpad + c.sb_bytes;
        c.h_pk = c.h_kg; c.h_sb = c.h_kg + pkpad; c.h_seeds = c.h_kg + pkpad + c.sb_bytes;
        c.reg_pp(&c.d_pk, c.kg_rec); c.reg_pp(&c.d_sb, c.kg_rec); c.reg_pp(&c.d_seeds, c.kg_rec);
        c.reg_pp(&c.h_pk, c.kg_rec); c.reg_pp(&c.h_sb, c.kg_rec); c.reg_pp(&c.h_seeds, c.kg_rec);
        HIPCHK(dev(&c.d_sehat, c.se_stride));
        HIPCHK(dev(&c.d_alpha, 80));
        HIPCHK(dalloc(&c.d_I, 2 * B * c.sel_stride)); // I rows then complement rows: one upload
        c.d_rest = c.d_I + B * c.sel_stride;
        c.reg_pp(&c.d_I, (size_t)c.sel_stride * 2); c.reg_pp(&c.d_rest, (size_t)c.sel_stride * 2);
        HIPCHK(dev(&c.d_pwT, (size_t)MAXM * 80));
        // data operand of the largest GEMM: every fresh sharing of every proof (<= 256 per proof), 13 k-steps
        c.limb_cap = ((B * 256 + 63) / 64 * 64 / 16) * (size_t)13 * 2048;
        HIPCHK(dev(&c.d_limbs, (size_t)(256 / 16) * 13 * 2048));
        HIPCHK(dev(&c.d_coef, 2 * (size_t)8 * 2 * 2048));
        HIPCHK(dev(&c.d_fail, 1));
        HIPCHK(host(&c.h_tape, c.tape_stride));
        HIPCHK(host(&c.h_dig, (size_t)NPARTY * 32));
        HIPCHK(host(&c.h_dig2, (size_t)NPARTY * 32));
        HIPCHK(host(&c.h_proof, c.image_stride));
        HIPCHK(host(&c.h_alpha, 80));
        HIPCHK(halloc(&c.h_I, 2 * B * c.sel_stride));
        c.h_rest = c.h_I + B * c.sel_stride;
        c.reg_pp(&c.h_I, (size_t)c.sel_stride * 2); c.reg_pp(&c.h_rest, (size_t)c.sel_stride * 2);
        HIPCHK(host(&c.h_fail, 1));
        HIPCHK(halloc(&c.h_err, 16));
        memset(c.h_err, 0, 16 * sizeof(uint32_t));
        memset(c.h_alpha, 0, B * 80 * sizeof(uint16_t));
        HIPCHK(stream_sync(c));
        return 0;
    };
    if (body()) return fail();
    *out = cp;
    return 0;
}

// A view of `arena` (kosk_ctx.hpp): same tables, same workspace, proofs [first, first + reach) of it, own stream / events / workers.
int ctx_make_view(Ctx &arena, int first, int own_batch, int reserve_threads, Ctx **out, std::string &err)
{
    if (arena.is_view || first < 0 || own_batch < 1 || first + own_batch > arena.max_batch) { err = "internal: bad view range"; return -1; }
    Ctx *vp = new Ctx(arena); // memberwise copy: constants, strides, table pointers, knobs
    Ctx &c = *vp;
    c.is_view = true;
    c.view_first = first;
    c.max_batch = arena.max_batch - first;
    c.own_batch = own_batch;
    c.call_cap = own_batch;
    c.err.clear();
    // what the copy must not share with the arena.  The STREAM is shared on purpose: one HIP stream per cohort.  In steady state
    // a cohort has one merged run in flight, so nothing is lost -- and the number of streams that carry work stays at the
    // number of cohorts: ROCm deals streams to its (four) hardware queues in creation order, and with a stream per member the
    // streams of the run leaders of three cohorts of three all landed on the SAME hardware queue (measured: 138-proof runs
    // took 2.8 ms instead of 1.6).  Calls end with a stream synchronisation, so members never leave work behind for each other.
    c.use_graphs = false; // stream capture is per stream: not with several callers on one
    c.ev = nullptr; c.ev_kg = nullptr; c.ev_sync = nullptr; c.pool = nullptr;
    c.host_img = nullptr;
    c.h_err = nullptr;
    for (auto &e : c.timer_ev) e = nullptr;
    for (auto &pe : c.prof_ev) for (auto &e : pe) e = nullptr;
    for (auto &g : c.seg) g = Ctx::SegGraph{};
    c.d_compact = nullptr; c.h_compact = nullptr; c.d_compact_bad = nullptr; c.h_compact_bad = nullptr;
    for (auto &x : c.path_n) x = 0;
    for (int i = 0; i < PR_COUNT; i++) { c.prof_ms[i] = 0; c.prof_n[i] = 0; c.prof_units[i] = 0; c.prof_used[i] = false; }
    c.tape_cur = nullptr;
    c.resident_pk_n = 0;
    c.rb = nullptr; c.rb_user = nullptr; c.round_hook = nullptr; c.round_user = nullptr;
    for (const Ctx::PerProof &pp : arena.per_proof) {
        char *base = *reinterpret_cast<char *const *>(reinterpret_cast<const char *>(&arena) + pp.field_off);
        *reinterpret_cast<char **>(reinterpret_cast<char *>(&c) + pp.field_off) = base ? base + (size_t)first * pp.stride_bytes : nullptr;
    }
    c.limb_cap = (size_t)c.max_batch * (256 / 16) * 13 * 2048;
    auto fail = [&]() { err = c.err; delete vp; return -1; };
    auto body = [&]() -> int {
        HIPCHK(hipSetDevice(c.device));
        HIPCHK(hipEventCreateWithFlags(&c.ev, hipEventDisableTiming | (c.blocking_sync ? hipEventBlockingSync : 0)));
        HIPCHK(hipEventCreateWithFlags(&c.ev_kg, hipEventDisableTiming | (c.blocking_sync ? hipEventBlockingSync : 0)));
        if (c.blocking_sync) HIPCHK(hipEventCreateWithFlags(&c.ev_sync, hipEventDisableTiming | hipEventBlockingSync));
        else if (c.wait_nap) HIPCHK(hipEventCreateWithFlags(&c.ev_sync, hipEventDisableTiming));
        for (auto &pe : c.prof_ev)
            for (auto &e : pe) HIPCHK(hipEventCreate(&e));
        HIPCHK(halloc(&c.h_err, 16)); // a view's kernels report to the view's own word
        memset(c.h_err, 0, 16 * sizeof(uint32_t));
        return 0;
    };
    if (body()) return fail();
    c.pool = pool_create();
    c.base_threads = arena.base_threads;
    const int got = pool_reserve(c.pool, reserve_threads > c.base_threads ? reserve_threads : c.base_threads);
    c.reserved_threads = got;
    c.nthreads = got < c.base_threads ? got : c.base_threads;
    *out = vp;
    return 0;
}

static bool is_device_pointer(const void *p)
{
    hipPointerAttribute_t at{};
    if (hipPointerGetAttributes(&at, p) != hipSuccess) {
        (void)hipGetLastError(); // plain host memory: not an error
        return false;
    }
    return at.type == hipMemoryTypeDevice;
}

// one caller's tapes for proofs [first, first + n): copied into the context's own tape buffer (merged calls, host memory,
// the callback) -- or, for a single caller whose aligned device buffer can be read in place, just noted
static int upload_tapes_part(Ctx &c, int first, int n, const uint8_t *tapes, size_t tape_stride, bool in_place_ok)
{
    const Params &P = c.P;
    if (tapes && tape_stride < P.tape_bytes) { c.err = "tape_stride smaller than kosk_tape_bytes"; return -1; }
    uint8_t *d_dst = c.d_tape + (size_t)first * c.tape_stride, *h_dst = c.h_tape + (size_t)first * c.tape_stride;
    if (tapes && is_device_pointer(tapes)) {
        // the caller keeps its randomness in HBM: use it in place when the kernels' 8-byte loads are aligned, else one D2D copy
        if (in_place_ok && tape_stride % 8 == 0 && (reinterpret_cast<uintptr_t>(tapes) & 7) == 0) {
            c.tape_cur = tapes;
            c.tape_cur_stride = tape_stride;
            return 0;
        }
        HIPCHK(hipMemcpy2DAsync(d_dst, c.tape_stride, tapes, tape_stride, P.tape_bytes, n, hipMemcpyDeviceToDevice, c.stream));
        c.tape_cur = c.d_tape;
        c.tape_cur_stride = c.tape_stride;
        return 0;
    }
    if (!tapes) {
        // draw through the randombytes callback in the reference's call order and
        // lengths (kosk.cpp:12, mlwe_prover.cpp:9, ss.cpp:5), proof after proof
        for (int b = 0; b < n; b++) {
            uint8_t *tp = h_dst + (size_t)b * c.tape_stride;
            auto draw = [&](size_t len) {
                if (c.rb) c.rb(c.rb_user, tp, len);
                else os_randombytes(tp, len);
                tp += len;
            };
            draw(64);
            for (int i = 0; i < P.M; i++) draw(32);
            for (int i = 0; i < P.nfresh; i++) draw(302);
        }
    } else {
        parallel_for(c.pool, n, c.nthreads, [&](int b) { memcpy(h_dst + (size_t)b * c.tape_stride, tapes + (size_t)b * tape_stride, P.tape_bytes); });
    }
    HIPCHK(hipMemcpyAsync(d_dst, h_dst, (size_t)n * c.tape_stride, hipMemcpyHostToDevice, c.stream));
    c.tape_cur = c.d_tape;
    c.tape_cur_stride = c.tape_stride;
    return 0;
}

int upload_tapes(Ctx &c, int n, const uint8_t *tapes, size_t tape_stride)
{
    c.tape_segs.count = 0;
    return upload_tapes_part(c, 0, n, tapes, tape_stride, true);
}

// the tapes of every caller of a (possibly merged) call, segment after segment
static int upload_tapes_segs(Ctx &c, int n, const KeygenIn &kg)
{
    c.tape_segs.count = 0;
    if (!kg.next && (kg.count == 0 || kg.count == n)) return upload_tapes(c, n, kg.tapes, kg.tape_stride);
    { // every caller keeps aligned tapes in HBM, equal strides, full blocks except the last: read them where they are
        const int per = kg.count;
        int j = 0, total = 0;
        bool ok = per > 0;
        for (const KeygenIn *s = &kg; s && ok; s = s->next, j++) {
            const int cnt = s->count ? s->count : n - total;
            ok = j < 16 && s->tapes && s->tape_stride == kg.tape_stride && s->tape_stride >= c.P.tape_bytes && s->tape_stride % 8 == 0 &&
                 (reinterpret_cast<uintptr_t>(s->tapes) & 7) == 0 && cnt >= 1 && (cnt == per || (!s->next && cnt < per)) && is_device_pointer(s->tapes);
            if (ok) c.tape_segs.ptr[j] = s->tapes;
            total += cnt;
        }
        if (ok && total == n) {
            c.tape_segs.per = per;
            c.tape_segs.count = j;
            c.tape_cur = kg.tapes;
            c.tape_cur_stride = kg.tape_stride;
            return 0;
        }
    }
    int first = 0;
    for (const KeygenIn *s = &kg; s; s = s->next) {
        const int cnt = s->count ? s->count : n - first;
        if (cnt < 1 || first + cnt > n) { c.err = "internal: merged call segments do not add up"; return -1; }
        if (upload_tapes_part(c, first, cnt, s->tapes, s->tape_stride, false)) return -1;
        first += cnt;
    }
    if (first != n) { c.err = "internal: merged call segments do not add up"; return -1; }
    return 0;
}

// kyber_keygen on the device (kosk.cpp:4-70): A, s, e never leave HBM; only pk, NTT(s) bytes and the seeds come back
int issue_keygen(Ctx &c, int n, bool sampled)
{
    const Params &P = c.P;
    const int K = P.K;
    // `sampled`: seeds, A, s, e were already produced as roles of the prover's first launch (issue_sharing_front)
    if (!sampled) HIPCHK(launch_keygen(c.tape_cur, c.tape_cur_stride, c.d_seeds, c.kg_rec, c.d_A, c.key_stride, c.d_se, c.se_stride, K, P.eta1, n, c.stream, c.xof_guard()));
    NttArgs na{};
    na.in = c.d_se; na.in_gstride = c.se_stride; na.src_off = nullptr;
    na.out = c.d_sehat; na.out_gstride = c.se_stride; na.dst_off = nullptr;
    na.npg = 2 * K; na.npoly = 2 * K * n; na.out_canonical = 0;
    HIPCHK(launch_ntt(na, c.stream)); // polyvec_ntt(s), polyvec_ntt(e)   kosk.cpp:39-40
    HIPCHK(launch_keygen_pack(c.d_A, c.key_stride, c.d_sehat, c.se_stride, c.d_seeds, c.kg_rec, c.d_t, c.d_pk, c.pk_stride, c.d_sb,
                              c.sb_stride, K, n, c.stream));
    HIPCHK(copy_small(c, c.h_kg, 0, c.d_kg, 0, (size_t)n * c.kg_rec, 1, hipMemcpyDeviceToHost, c.stream)); // pk, NTT(s) bytes, seeds: one copy
    c.kg_on_host_pending = false;
    if (!c.capturing) { HIPCHK(hipEventRecord(c.ev_kg, c.stream)); c.kg_on_host_pending = true; } // (a keygen-in-front call is never captured)
    c.resident_pk_n = n;
    return 0;
}

static void finish_keygen_part(Ctx &c, int first, int n, uint8_t *pk, uint8_t *sk)
{
    const Params &P = c.P;
    // device Fiat-Shamir: this is the only host job of a step and the GPU is far from waiting for it -- on the caller's own thread, without
    // waking the workers of every member of a merged run for 46 records of ~4 us each (native callers, sixteen per cohort: 4.8 busy host
    // cores with three workers per caller, 2.9 with this)
    parallel_for(c.pool, n, c.fs_device ? 1 : c.nthreads, [&](int i) { // sk = NTT(s) bytes || pk || H(pk) || z, z = noise seed   kosk.cpp:62-69
        const int b = first + i;
        uint8_t *pkb = pk + (size_t)i * P.pk_bytes, *skb = sk + (size_t)i * P.sk_bytes;
        memcpy(pkb, c.h_pk + (size_t)b * c.pk_stride, P.pk_bytes);
        memcpy(skb, c.h_sb + (size_t)b * c.sb_stride, c.sb_bytes);
        memcpy(skb + c.sb_bytes, pkb, P.pk_bytes);
        sha3_256(skb + P.sk_bytes - 64, pkb, P.pk_bytes);
        memcpy(skb + P.sk_bytes - 32, c.h_seeds + (size_t)b * c.kg_rec + 32, 32);
    });
}

void finish_keygen_host(Ctx &c, int n, uint8_t *pk, uint8_t *sk)
{
    finish_keygen_part(c, 0, n, pk, sk);
}

static void finish_keygen_segs(Ctx &c, int n, const KeygenIn &kg)
{
    // ONE job over the proofs of all callers of a merged run (round 6; a job per caller woke the run's workers once per member)
    struct Seg { int first, cnt; uint8_t *pk, *sk; };
    constexpr int MAXSEG = 16; // kosk_combine.hpp: Combiner::MAX_WIDTH
    Seg segs[MAXSEG];
    int nseg = 0, first = 0;
    const KeygenIn *s = &kg;
    for (; s && first < n && nseg < MAXSEG; s = s->next) {
        const int cnt = s->count ? s->count : n - first;
        segs[nseg++] = Seg{first, cnt, s->pk, s->sk};
        first += cnt;
    }
    for (int rest = first; s && rest < n; s = s->next) { // (more members than a cohort can have: one by one)
        const int cnt = s->count ? s->count : n - rest;
        finish_keygen_part(c, rest, cnt, s->pk, s->sk);
        rest += cnt;
    }
    if (nseg == 1) { finish_keygen_part(c, segs[0].first, segs[0].cnt, segs[0].pk, segs[0].sk); return; }
    const Params &P = c.P;
    parallel_for(c.pool, first, c.fs_device ? 1 : c.nthreads, [&](int b) {
        int k = 0;
        while (k + 1 < nseg && b >= segs[k + 1].first) k++;
        const int i = b - segs[k].first;
        uint8_t *pkb = segs[k].pk + (size_t)i * P.pk_bytes, *skb = segs[k].sk + (size_t)i * P.sk_bytes;
        memcpy(pkb, c.h_pk + (size_t)b * c.pk_stride, P.pk_bytes);
        memcpy(skb, c.h_sb + (size_t)b * c.sb_stride, c.sb_bytes);
        memcpy(skb + c.sb_bytes, pkb, P.pk_bytes);
        sha3_256(skb + P.sk_bytes - 64, pkb, P.pk_bytes);
        memcpy(skb + P.sk_bytes - 32, c.h_seeds + (size_t)b * c.kg_rec + 32, 32);
    });
}

int stage_prover_inputs(Ctx &c, int n, const uint8_t *tapes, size_t tape_stride, uint8_t *pk, uint8_t *sk)
{
    if (n < 1 || n > c.call_cap) { c.err = "batch size out of range"; return -1; }
    if (!pk || !sk) { c.err = "pk / sk output buffers are required"; return -1; }
    HIPCHK(hipSetDevice(c.device));
    const double t0 = now_sec();
    if (upload_tapes(c, n, tapes, tape_stride)) return -1;
    if (issue_keygen(c, n)) return -1;
    HIPCHK(stream_sync(c));
    if (device_error_check(c)) return -1;
    finish_keygen_host(c, n, pk, sk);
    c.phase_sec[PH_HOST_PRE] = now_sec() - t0;
    return 0;
}

// The sharing front of the prover for the fresh sharings [s0, s1) of the tape order: tape / witness expansion, NTTs,
// A*NTT(s), and the Lagrange expansion of exactly those rows.  FRONT_FULL is what kyber_verifiable_keygen needs;
// the other three are the reference's separate entry points (mlwe_prover.cpp:4-39, :41-59, and the sharing part of :81-).
int issue_sharing_front(Ctx &c, int n, FrontPart part, bool with_keygen)
{
    const Params &P = c.P;
    const RowMap &rm = c.rm;
    const int K = P.K, noff_f = 2 * P.M, noff = 2 * P.M + 2 * K * P.E;
    hipStream_t st = c.stream;
    int s0 = 0, s1 = P.nfresh, witness = 1, ntt_first = 0, ntt_count = c.n_ntt1;
    bool expand = true, matvec = true;
    if (part == FRONT_RANDOMNESS) { s1 = noff_f; witness = 0; ntt_count = P.M; matvec = false; }
    else if (part == FRONT_RANGE) { s0 = noff_f; s1 = noff; witness = 2; ntt_count = 0; expand = false; matvec = false; }
    else if (part == FRONT_ONLINE) { s0 = noff; expand = false; ntt_first = P.M; ntt_count = K; }
    const KeygenFront kgf{c.d_seeds, c.kg_rec, c.d_A, c.key_stride, c.d_se, c.se_stride, c.xof_guard()};
    HIPCHK(launch_prover_pre(c.tape_cur, c.tape_cur_stride, c.d_P, c.proof_stride, rm.f, P.M, 64 + 32 * P.M, c.d_fresh_rows, s0, s1, expand,
                             witness, c.d_se, c.se_stride, rm, P.eta1, n, st, with_keygen ? &kgf : nullptr, c.tape_segs.count ? &c.tape_segs : nullptr));
    if (with_keygen && issue_keygen(c, n, true)) return -1; // NTT(s), NTT(e), t = A o s + e, pk / sk bytes and their D2H
    if (ntt_count > 0) {
        NttArgs na{};
        na.in = reinterpret_cast<const int16_t *>(c.d_P);
        na.in_gstride = c.proof_stride;
        na.src_off = c.d_off + c.off_ntt1_src + ntt_first; // NTT(f_i) -> Tf_i secrets (mlwe_prover.cpp:17-26) and NTT(s_i) (:256)
        na.out = reinterpret_cast<int16_t *>(c.d_P);
        na.out_gstride = c.proof_stride;
        na.dst_off = c.d_off + c.off_ntt1_dst + ntt_first;
        na.npg = ntt_count;
        na.npoly = ntt_count * n;
        na.out_canonical = 1;
        c.prof_begin(PR_NTT_F, n);
        HIPCHK(launch_ntt(na, st));
        c.prof_end(PR_NTT_F);
    }
    if (matvec) HIPCHK(launch_matvec_ntt(c.d_A, c.key_stride, c.d_P, c.proof_stride, rm.shat, rm.nttas, K, n, st)); // :284-285
    const GemmSrc xsrc{c.d_P, c.proof_stride, c.d_gemm1_rows + s0, RS, 0, XLEN};
    const GemmDst xdst{c.d_P, c.proof_stride, c.d_gemm1_rows + s0, RS, EXP_OFF};
    c.prof_begin(PR_GEMM_EXPAND1, n);
    if (gemm_modq(c, c.t_expand, xsrc, xdst, s1 - s0, n)) return -1;
    c.prof_end(PR_GEMM_EXPAND1);
    return 0;
}

int prove_resident(Ctx &c, int n, bool online_only, const KeygenIn *keygen)
{
    if (n < 1 || n > c.call_cap) { c.err = "batch size out of range"; return -1; }
    for (const KeygenIn *s = keygen; s; s = s->next)
        if (!s->pk || !s->sk) { c.err = "pk / sk output buffers are required"; return -1; }
    if (!keygen && !c.tape_cur) { c.err = "no resident prover inputs: call kosk_stage_prover_inputs first"; return -1; }
    if (!keygen) c.tape_segs.count = 0;
    HIPCHK(hipSetDevice(c.device));
    const Params &P = c.P;
    const RowMap &rm = c.rm;
    const int K = P.K;
    hipStream_t st = c.stream;
    double t0 = now_sec(), t1;

    HashArgs ha{};
    ha.rows = c.d_P;
    ha.group_stride = c.proof_stride;
    ha.row_stride = RS;
    ha.col_off = NSEC;
    ha.lanes_per_group = NPARTY;
    ha.lane_map = nullptr;
    ha.out_lanes_per_group = NPARTY;

    // ---- key generation rides in the first launch of P1 (tape pointers may change from call to call: never part of a captured graph)
    if (keygen && upload_tapes_segs(c, n, *keygen)) return -1;
    // ---- P1: offline phase + witness sharing (secrets, randoms, one expansion GEMM), Tcomm of every party
    if (run_segment(c, (online_only || keygen) ? -1 : (int)Ctx::SEG_P1, n, [&]() -> int {
        if (issue_sharing_front(c, n, online_only ? FRONT_ONLINE : FRONT_FULL, keygen != nullptr)) return -1;
        HashArgs h1 = ha;
        h1.prefix = nullptr;
        h1.out = c.d_dig1;
        HIPCHK(commit_hash_batch(c, h1, n, K, false, st));
        if (c.fs_device) return 0; // the table stays where it is: the chain kernel below hashes it in HBM
        HIPCHK(copy_round_table(c, c.h_dig, c.d_dig1, n));
        if (!c.capturing) c.path_n[PATH_DIGEST_COPY]++;
        return 0;
    }, c.tape_cur, c.tape_cur_stride)) return -1; // the tape pointer is baked into the captured launch: part of the graph's key
    HIPCHK(hipEventRecord(c.ev, st)); // the Tcomm digests are on the host (device Fiat-Shamir: complete in HBM) once this event has passed
    if (c.fs_device) {
        // ---- Fiat-Shamir round 1 on the device: h1 = sha3_256(Tcomm[0..N)), alpha = BE16(PRF(h1, 1)) % q, one wave per proof   :130-153
        FsArgs fa{};
        fa.in = c.d_dig1; fa.in_stride = (size_t)NPARTY * 32; fa.len = NPARTY * 32;
        fa.alpha = c.d_alpha; fa.alpha_stride = 80; fa.J = P.J;
        c.prof_begin(PR_FS_ALPHA, n);
        HIPCHK(launch_fs_chain(fa, FS_ALPHA, n, st));
        c.prof_end(PR_FS_ALPHA);
        c.path_n[PATH_FS_DEVICE]++;
    }
    c.phase_sec[PH_P1_ISSUE] = now_sec() - t0;

    // ---- P1B: what neither Tcomm nor alpha needs is queued behind the digest copy BEFORE the host waits for it, and runs
    // while the host hashes: the multiplication gates on the expanded shares (:338-381) and the transposed limb form of
    // the f rows for the beta/gamma product
    if (run_segment(c, Ctx::SEG_P1B, n, [&]() -> int {
        HIPCHK(launch_post_gates(c.d_P, c.proof_stride, rm, n, st));
        return 0;
    })) return -1;
    // the host half of the key generation (sk = NTT(s) bytes || pk || H(pk) || z per proof) needs only the key records, which left
    // the GPU behind the first launches: done HERE, under the expansion product, the Tcomm hash and the digest copy, not after them
    // in front of the host's first round (where it was 30-40 us of a step's critical path with the GPU idle)
    bool keys_done = false;
    if (keygen && c.kg_on_host_pending) {
        HIPCHK(hipEventSynchronize(c.ev_kg));
        finish_keygen_segs(c, n, *keygen);
        keys_done = true;
    }
    c.kg_on_host_pending = false;
    // device Fiat-Shamir: the host has nothing to wait for here unless somebody wants to be told that the round's table is complete
    auto any_hook = [&]() {
        bool h = c.round_hook != nullptr;
        for (const KeygenIn *s = keygen; s; s = s->next) h |= s->hook != nullptr;
        return h;
    };
    const bool hooks = c.fs_device && any_hook();
    if (!c.fs_device || hooks) HIPCHK(wait_event(c, c.ev, 0, n)); // the table
    t1 = now_sec(); c.phase_sec[PH_GPU_COMMIT] = t1 - t0; t0 = t1;
    // a round's table is complete in HBM: the hook of every caller of this run with ITS block of the table (a merged run), else the
    // context's own hook with the whole batch
    auto fire_hooks = [&](int rnd, const uint8_t *d_table) {
        bool seg_hooks = false;
        for (const KeygenIn *s = keygen; s; s = s->next) seg_hooks |= s->hook != nullptr;
        if (!seg_hooks) {
            if (c.round_hook) c.round_hook(c.round_user, 0, rnd, d_table, (size_t)n * NPARTY * 32);
            return;
        }
        int first = 0;
        for (const KeygenIn *s = keygen; s && first < n; s = s->next) {
            const int cnt = s->count ? s->count : n - first;
            if (s->hook) s->hook(s->hook_user, 0, rnd, d_table + (size_t)first * NPARTY * 32, (size_t)cnt * NPARTY * 32);
            first += cnt;
        }
    };
    if (!c.fs_device || hooks) fire_hooks(0, c.d_dig1);

    // ---- Fiat-Shamir round 1 on the host
    if (keygen && !keys_done) finish_keygen_segs(c, n, *keygen);
    if (!c.fs_device) {
        fs_alpha_batch(P, n, c.h_dig, (size_t)NPARTY * 32, c.h_alpha, 80, c.nthreads, c.pool);
        c.path_n[PATH_FS_HOST]++;
    }
    t1 = now_sec(); c.phase_sec[PH_FS_ALPHA] = t1 - t0; t0 = t1;

    // ---- P2: beta, gamma, r, NTT_r on every evaluation point (per proof a [J x M] x [M x 1710] product mod q, :159-203),
    // s + r / e + r (:222-245), then the view commitments, which read nothing else of the relation phase
    if (run_segment(c, Ctx::SEG_P2, n, [&]() -> int {
        // host mode: the challenge vectors are read by k_coef_limbs straight from the page-locked host table (160 bytes per proof): no copy
        // launch between the host's round and the product; device mode: from where the chain kernel wrote them
        const uint16_t *alpha_src = c.fs_device ? c.d_alpha : c.h_alpha;
        c.prof_begin(PR_LINCOMB, n);
        HIPCHK(launch_coef_limbs(alpha_src, P.J, P.M, c.d_coef, n, st));
        HIPCHK(launch_lincomb_stream(c.d_P, c.proof_stride, rm, c.d_coef, c.d_P, c.d_lin_rows, P.J, n, st)); // includes s + r, e + r
        c.prof_end(PR_LINCOMB);
        return 0;
    })) return -1;
    // the graded kernel stays a plain launch so that HIP events can bracket it inside the timed region
    ha.prefix = c.d_dig1;
    ha.out = c.d_dig2;
    HIPCHK(commit_hash_batch(c, ha, n, K, true, st));
    if (!c.fs_device) {
        HIPCHK(copy_round_table(c, c.h_dig2, c.d_dig2, n));
        c.path_n[PATH_DIGEST_COPY]++;
    }
    HIPCHK(hipEventRecord(c.ev, st));
    if (c.fs_device) {
        // ---- Fiat-Shamir round 2 on the device: ch = sha3_256(ch_seeds), I from PRF(ch, 1) with the reference's probing, its complement,
        // the window boundaries and the sorted opened list, straight into the rows the wire-image kernel reads   :445-474
        FsArgs fa{};
        fa.in = c.d_dig2; fa.in_stride = (size_t)NPARTY * 32; fa.len = NPARTY * 32;
        fa.I = c.d_I; fa.rest = c.d_rest; fa.sel_stride = c.sel_stride;
        c.prof_begin(PR_FS_OPENED, n);
        HIPCHK(launch_fs_chain(fa, FS_OPENED, n, st));
        c.prof_end(PR_FS_OPENED);
        c.path_n[PATH_FS_DEVICE]++;
    }
    c.phase_sec[PH_P2_ISSUE] = now_sec() - t0;

    // ---- P2B: the NTT-domain half of the relation is not hashed, only opened; queued behind the digest copy before the
    // host waits for it, it runs while the host derives I
    if (run_segment(c, Ctx::SEG_P2B, n, [&]() -> int {
        NttArgs na{};
        na.in = reinterpret_cast<const int16_t *>(c.d_P);
        na.in_gstride = c.proof_stride;
        na.src_off = c.d_off + c.off_sr_er; // NTT of the opened s+r, e+r     :260-277
        na.npg = 2 * K;
        na.npoly = 2 * K * n;
        na.out = reinterpret_cast<int16_t *>(c.d_P);
        na.out_gstride = c.proof_stride;
        na.dst_off = c.d_off + c.off_nttsr_er;
        na.out_canonical = 1;
        HIPCHK(launch_relation_ntt(na, c.d_A, c.key_stride, c.d_P, c.proof_stride, rm, n, st)); // NTT, A o NTT(s+r) (:287-288), tails
        const GemmSrc x2src{c.d_P, c.proof_stride, c.d_gemm2_rows, RS, 0, XLEN};
        const GemmDst x2dst{c.d_P, c.proof_stride, c.d_gemm2_rows, RS, EXP_OFF};
        c.prof_begin(PR_GEMM_EXPAND2, n);
        if (gemm_modq(c, c.t_expand, x2src, x2dst, c.n_gemm2, n)) return -1; // recompute_share_secrets_ddeg x 3K   :298-299,:315
        c.prof_end(PR_GEMM_EXPAND2);
        HIPCHK(launch_post_relation(c.d_P, c.proof_stride, rm, n, st));
        return 0;
    })) return -1;

    if (!c.fs_device || hooks) {
        HIPCHK(wait_event(c, c.ev, 1, n));
        t1 = now_sec(); c.phase_sec[PH_GPU_RELATION] = t1 - t0; t0 = t1;
        fire_hooks(1, c.d_dig2);
    }

    // ---- Fiat-Shamir round 2 on the host
    // I, its complement, and the complement entries owned by each aligned 64-party window (k_assemble_fields), all derived by
    // the worker that hashed the proof's table
    if (!c.fs_device) {
        fs_opened_batch(n, c.h_dig2, (size_t)NPARTY * 32, c.h_I, c.h_rest, c.sel_stride, c.nthreads, c.pool, true);
        c.path_n[PATH_FS_HOST]++;
    }
    t1 = now_sec(); c.phase_sec[PH_FS_OPEN] = t1 - t0; t0 = t1;

    // ---- P3: wire image
    if (run_segment(c, Ctx::SEG_P3, n, [&]() -> int {
        if (c.fs_device) {
            // the lists are in HBM already (k_fs_chain<FS_OPENED>)
        } else if (!c.is_view) {
            HIPCHK(copy_small(c, c.d_I, 0, c.h_I, 0, ((size_t)c.max_batch + n) * c.sel_stride * 2, 1, hipMemcpyHostToDevice, st)); // I and its complement
        } else { // a view's lists sit inside the arena's two blocks: other views' lists lie between them
            HIPCHK(copy_small(c, c.d_I, 0, c.h_I, 0, (size_t)n * c.sel_stride * 2, 1, hipMemcpyHostToDevice, st));
            HIPCHK(copy_small(c, c.d_rest, 0, c.h_rest, 0, (size_t)n * c.sel_stride * 2, 1, hipMemcpyHostToDevice, st));
        }
        AssembleArgs aa{};
        aa.P = c.d_P;
        aa.proof_stride = c.proof_stride;
        aa.rowtab = c.d_rowtab;
        aa.opened = c.d_I;
        aa.rest = c.d_rest;
        aa.sel_stride = c.sel_stride;
        aa.dig1 = c.d_dig1;
        aa.dig2 = c.d_dig2;
        aa.proof = c.d_proof;
        aa.image_stride = c.image_stride;
        aa.groups = c.d_asm_groups;
        aa.elems = c.d_asm_elems;
        aa.ngroups = c.n_asm_groups;
        c.prof_begin(PR_ASSEMBLE, n);
        HIPCHK(launch_assemble(aa, P.off[F_TCOMM], P.off[F_COMM], P.off[F_I], n, st));
        c.prof_end(PR_ASSEMBLE);
        if (c.near_end_hook) c.near_end_hook(); // the last kernel is queued: a merged run's sleeping callers get ready for the return
        return 0;
    })) return -1;
    c.phase_sec[PH_P3_ISSUE] = now_sec() - t0;
    c.tape_segs.count = 0; // the callers' tape buffers are only promised for this call
    HIPCHK(stream_sync_site(c, 2, n));
    t1 = now_sec(); c.phase_sec[PH_GPU_ASSEMBLE] = t1 - t0;
    c.prof_collect();
    if (device_error_check(c)) return -1; // e.g. the key generation's gen_matrix hit its block limit: pk / sk / proofs are not valid
    return 0;
}

int fetch_proofs(Ctx &c, int n, uint8_t *pi, bool registered)
{
    if (n < 1 || n > c.call_cap) { c.err = "batch size out of range"; return -1; }
    HIPCHK(hipSetDevice(c.device));
    const double t0 = now_sec();
    if (registered) {
        // the caller's buffer is page-locked for the duration of the call (kosk_capi.cpp): the images go straight there
        HIPCHK(hipMemcpy2DAsync(pi, c.P.proof_bytes, c.d_proof, c.image_stride, c.P.proof_bytes, n, hipMemcpyDeviceToHost, c.stream));
        HIPCHK(stream_sync(c));
        c.path_n[PATH_COPY_DIRECT]++;
    } else {
        c.path_n[PATH_COPY_STAGED]++;
        HIPCHK(hipMemcpyAsync(c.h_proof, c.d_proof, (size_t)n * c.image_stride, hipMemcpyDeviceToHost, c.stream));
        HIPCHK(stream_sync(c));
        parallel_for(c.pool, n, c.nthreads, [&](int b) { memcpy(pi + (size_t)b * c.P.proof_bytes, c.h_proof + (size_t)b * c.image_stride, c.P.proof_bytes); });
    }
    c.phase_sec[PH_D2H] = now_sec() - t0;
    return 0;
}

} // namespace kosk
