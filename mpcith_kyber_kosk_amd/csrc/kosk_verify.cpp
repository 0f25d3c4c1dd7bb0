// Batched verifier pipeline (reference verify(), mlwe_verifier.cpp:4-686, called
// from kyber_kosk_verify, kosk.cpp:88-117).
//
//   V0  I / rest from the proof image (host, 300 bytes per proof)        :8-19
//   V1  scatter the proof into rows; Tcomm of the opened parties          :22-38   k_commit_hash (lane map)
//       -> host: alpha                                                    :40-65
//   V2  beta/gamma/r/NTT_r on the opened columns; recon x140; NTT check   :67-170  k_lincomb, k_gemm_modq, k_ntt256
//   V4  interpolate the unopened s+r, e+r, t, eta shares: per-proof
//       operator W (407x407) built on the GPU, applied as a GEMM, then
//       the Lagrange expansion and the share comparisons                  :173-247, :316-352, :382-444
//   V5-V8 NTT / A(s+r) / relation checks on the opened columns            :257-312, :365-376, :447-466
//   V9  multiplication gates: u on opened columns, 813-node interpolation
//       operator W2 (256x813), recon_secrets_2ddeg                        :469-571
//   V10 view hashes of the opened parties -> host: I' == I                :584-683
// Every check sets a bit of fail[proof]; the verify bit is fail == 0.
#include <atomic>
#include <cstring>
#include <functional>
#include <mutex>
#include <utility>
#include <vector>

#include "kosk_ctx.hpp"

#include <chrono>
#include "kosk_math.hpp"

namespace kosk {

static double now_sec()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

#define HIPCHK(x) KOSK_HIPCHK(x)

template <typename T>
static hipError_t dalloc(T **p, size_t n) { return hipMalloc(reinterpret_cast<void **>(p), (n ? n : 1) * sizeof(T)); }
template <typename T>
static int upload_vec(Ctx &c, T **d, const std::vector<T> &v)
{
    HIPCHK(dalloc(d, v.size()));
    // on the context's own stream (a non-blocking stream is NOT ordered against the legacy null stream that a plain hipMemcpy /
    // hipMemset uses), and complete before `v` goes away: the caller synchronises the stream before it returns
    if (!v.empty()) {
        HIPCHK(hipMemcpyAsync(*d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, c.stream));
        HIPCHK(hipStreamSynchronize(c.stream));
    }
    return 0;
}

int ensure_verify_workspace(Ctx &c)
{
    if (c.verify_ready) return 0;
    if (c.is_view) { c.err = "internal: a view's verifier workspace is allocated with its arena"; return -1; }
    const Params &P = c.P;
    const RowMap &rm = c.rm;
    const int K = P.K, M = P.M, E = P.E, Z = P.Z;
    const size_t B = (size_t)c.max_batch;

    std::vector<uint16_t> inv(Q + 1, 0); // one u16 of padding: k_interp_build copies the table to LDS as dwords
    for (int a = 0; a < Q; a++) inv[a] = gf_inv_host((uint16_t)a);
    if (upload_vec(c, &c.d_inv, inv)) return -1;
    { // limb pairs of 1/d for every difference d = k - x_j the interpolation meets (k_interp_apply)
        const int len = interp_table_len(), off = interp_table_off();
        std::vector<uint16_t> il(len, 0);
        for (int i = 0; i + 2 < len; i++) {
            const int d = ((i - off) % Q + Q) % Q;
            int c0, c1;
            limb_split(gf_center(inv[d]), c0, c1);
            il[i] = (uint16_t)((c0 & 0xFF) | ((c1 & 0xFF) << 8));
        }
        if (upload_vec(c, &c.d_invlimb, il)) return -1;
    }
    std::vector<uint16_t> fact(Q), invfact(Q);
    fact[0] = 1;
    for (int a = 1; a < Q; a++) fact[a] = (uint16_t)((uint32_t)fact[a - 1] * a % Q);
    for (int a = 0; a < Q; a++) invfact[a] = gf_inv_host(fact[a]);
    if (upload_vec(c, &c.d_fact, fact) || upload_vec(c, &c.d_invfact, invfact)) return -1;

    // where each proof field lands in the verifier's row matrix
    std::vector<FieldDesc> vf;
    std::vector<int16_t> rt;
    auto add = [&](int fid, int sel, int width, auto rowfn, int limit = 0, int by_party = 0, int noncanon_bit = -1) {
        FieldDesc fd{};
        fd.noncanon_bit = noncanon_bit;
        fd.off = (uint32_t)P.off[fid];
        fd.sel = sel;
        fd.width = width;
        fd.limit = limit;
        fd.limit_by_party = by_party;
        fd.rowtab_off = (int)rt.size();
        for (int e = 0; e < width; e++) rt.push_back((int16_t)rowfn(e));
        vf.push_back(fd);
    };
    add(F_F, 0, M, [&](int e) { return rm.f + e; });
    add(F_NTTF, 0, M, [&](int e) { return rm.tf + e; });
    add(F_BETA, 1, NCHK, [&](int e) { return rm.beta(e); }, DEG + 1, 1);   // recon_secrets_ddeg: shares of parties 0..406  :106-107
    add(F_GAMMA, 1, NCHK, [&](int e) { return rm.gamma(e); }, DEG + 1, 1);
    add(F_S, 0, K, [&](int e) { return rm.s + e; });
    add(F_E, 0, K, [&](int e) { return rm.e + e; });
    add(F_T, 1, K, [&](int e) { return rm.t_in + e; }, DEG + 1);            // interpolation nodes: the first 407 unopened parties  :321-323
    add(F_NTTS, 0, K, [&](int e) { return rm.ntts + e; });
    add(F_NTTE, 0, K, [&](int e) { return rm.ntte + e; });
    add(F_NTTAR, 0, K, [&](int e) { return rm.nttar + e; });
    add(F_NTTAS, 0, K, [&](int e) { return rm.nttas + e; });
    // s + r / e + r shares of EVERY unopened party are compared raw with the recomputed (canonical) ones: an element >= q fails that check  :232-246
    add(F_SR, 1, K, [&](int e) { return rm.sr_in + e; }, 0, 0, FB_SR_ER_SHARES);
    add(F_ER, 1, K, [&](int e) { return rm.er_in + e; }, 0, 0, FB_SR_ER_SHARES);
    add(F_SETA, 1, K * E, [&](int e) { return rm.seta_in + e; }, DEG + 1); // :390-394
    add(F_EETA, 1, K * E, [&](int e) { return rm.eeta_in + e; }, DEG + 1);
    add(F_SSUB, 0, K * E, [&](int e) { return rm.ssub + e; });
    add(F_ESUB, 0, K * E, [&](int e) { return rm.esub + e; });
    add(F_ZS, 0, K * Z, [&](int e) { return rm.zs(e / Z, e % Z); });
    add(F_ZE, 0, K * Z, [&](int e) { return rm.ze(e / Z, e % Z); });
    add(F_US, 1, K * Z, [&](int e) { return rm.us(e / Z, e % Z); }, DEG2 + 1); // nodes: the first 813 unopened parties; recon: parties 0..812  :503-507, :555-556
    add(F_UE, 1, K * Z, [&](int e) { return rm.ue(e / Z, e % Z); }, DEG2 + 1);
    c.n_vfields = (int)vf.size();
    c.vplan = make_field_plan(vf.data(), c.n_vfields);
    if (upload_vec(c, &c.d_vfields, vf)) return -1;
    if (upload_vec(c, &c.d_vrowtab, rt)) return -1;

    std::vector<int16_t> bg, isrc, idst, urows;
    for (int j = 0; j < NCHK; j++) bg.push_back((int16_t)rm.beta(j));
    for (int j = 0; j < NCHK; j++) bg.push_back((int16_t)rm.gamma(j));
    auto pair = [&](int src0, int dst0, int n) { for (int i = 0; i < n; i++) { isrc.push_back((int16_t)(src0 + i)); idst.push_back((int16_t)(dst0 + i)); } };
    pair(rm.sr_in, rm.sr, K);
    pair(rm.er_in, rm.er, K);
    pair(rm.t_in, rm.t, K);
    pair(rm.seta_in, rm.seta, K * E);
    pair(rm.eeta_in, rm.eeta, K * E);
    for (int i = 0; i < K; i++) for (int k = 0; k < Z; k++) urows.push_back((int16_t)rm.us(i, k));
    for (int i = 0; i < K; i++) for (int k = 0; k < Z; k++) urows.push_back((int16_t)rm.ue(i, k));
    c.n_interp_d = (int)isrc.size();
    c.n_interp_2d = (int)urows.size();
    if (upload_vec(c, &c.d_rows_bg, bg) || upload_vec(c, &c.d_rows_isrc, isrc) || upload_vec(c, &c.d_rows_idst, idst) ||
        upload_vec(c, &c.d_rows_u, urows))
        return -1;

    c.o_stride = (size_t)rm.nrows * OS;
    // per-proof buffers, registered for views like the ones of ctx_create
    auto dev = [&](auto **p, size_t per) -> hipError_t {
        const hipError_t e = dalloc(p, B * per);
        if (e == hipSuccess) c.reg_pp(p, per * sizeof(**p));
        return e;
    };
    HIPCHK(dev(&c.d_O, c.o_stride));
    HIPCHK(hipMemsetAsync(c.d_O, 0, B * c.o_stride * sizeof(uint16_t), c.stream));
    HIPCHK(dev(&c.d_w, (size_t)2 * 832));
    HIPCHK(dev(&c.d_ell, 416));
    HIPCHK(dev(&c.d_node_of, 416));
    HIPCHK(dev(&c.d_isort, (size_t)c.sel_stride));
    HIPCHK(dev(&c.d_hrange, 4));
    HIPCHK(dev(&c.d_gather, interp_y_bytes(0)));
    HIPCHK(dev(&c.d_gather2, interp_y_bytes(1)));
    HIPCHK(dev(&c.d_sec, (size_t)2 * NCHK * 256));
    HIPCHK(dev(&c.d_sec_u1, (size_t)c.n_interp_2d * 256));
    HIPCHK(dev(&c.d_sec_u2, (size_t)c.n_interp_2d * 256));
    HIPCHK(hipHostMalloc(reinterpret_cast<void **>(&c.h_Iimg), B * 2 * NOPEN, hipHostMallocDefault));
    c.reg_pp(&c.h_Iimg, (size_t)2 * NOPEN);
    HIPCHK(dev(&c.d_odig, (size_t)NOPEN * 32));
    HIPCHK(hipHostMalloc(reinterpret_cast<void **>(&c.h_odig), B * NOPEN * 32, hipHostMallocDefault));
    c.reg_pp(&c.h_odig, (size_t)NOPEN * 32);
    HIPCHK(hipStreamSynchronize(c.stream)); // every table and the zeroed opened matrix are in HBM before the first verifier kernel is queued
    c.verify_ready = true;
    return 0;
}

int stage_verifier_inputs(Ctx &c, int n, const uint8_t *pi, const uint8_t *pk, bool registered)
{
    if (n < 1 || n > c.call_cap) { c.err = "batch size out of range"; return -1; }
    HIPCHK(hipSetDevice(c.device));
    const Params &P = c.P;
    parallel_for(c.pool, n, c.nthreads, [&](int b) {
        memcpy(c.h_pk + (size_t)b * c.pk_stride, pk + (size_t)b * P.pk_bytes, P.pk_bytes);
        if (!registered) memcpy(c.h_proof + (size_t)b * c.image_stride, pi + (size_t)b * P.proof_bytes, P.proof_bytes);
    });
    HIPCHK(hipMemcpyAsync(c.d_pk, c.h_pk, (size_t)n * c.pk_stride, hipMemcpyHostToDevice, c.stream));
    c.resident_pk_n = n;
    c.path_n[registered ? PATH_COPY_DIRECT : PATH_COPY_STAGED]++;
    if (registered) // page-locked caller memory (kosk_capi.cpp): no staging copy
        HIPCHK(hipMemcpy2DAsync(c.d_proof, c.image_stride, pi, P.proof_bytes, P.proof_bytes, n, hipMemcpyHostToDevice, c.stream));
    else
        HIPCHK(hipMemcpyAsync(c.d_proof, c.h_proof, (size_t)n * c.image_stride, hipMemcpyHostToDevice, c.stream));
    // polyvec_frombytes(t) and gen_matrix(A, seed) on the device   kosk.cpp:94-99
    HIPCHK(launch_decode_pk(c.d_pk, c.pk_stride, c.d_t, c.d_A, c.key_stride, P.K, n, c.stream, c.xof_guard()));
    HIPCHK(stream_sync(c));
    if (device_error_check(c)) return -1;
    return 0;
}

static bool is_device_pointer(const void *p)
{
    hipPointerAttribute_t at{};
    if (hipPointerGetAttributes(&at, p) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return at.type == hipMemoryTypeDevice;
}

int verify_resident(Ctx &c, int n, uint8_t *ok, int pk_mode, const uint8_t *pk, const VerifySeg *segs)
{
    // the caller's host copy of the images is valid for THIS call only, whichever way the call ends (an early error return must
    // not leave the pointer behind for the next resident call on this context)
    const uint8_t *himg = c.host_img;
    const size_t himg_stride = c.host_img_stride;
    c.host_img = nullptr;
    if (n < 1 || n > c.call_cap) { c.err = "batch size out of range"; return -1; }
    // a merged call (kosk_combine.hpp): `segs` lists the callers' parts, each with its own keys and result bytes
    const VerifySeg whole{n, pk, ok, nullptr, nullptr, nullptr};
    if (!segs) segs = &whole;
    {
        int total = 0;
        for (const VerifySeg *s = segs; s; s = s->next) {
            if (!s->ok) { c.err = "ok output buffer is required"; return -1; }
            if (pk_mode == 1 && !s->pk) { c.err = "pk_mode 1 needs the public keys"; return -1; }
            total += s->count;
        }
        if (total != n) { c.err = "internal: merged call segments do not add up"; return -1; }
    }
    if (pk_mode == 2 && c.resident_pk_n < n) {
        c.err = "no resident public keys for this batch: pk == NULL needs a key generation (or a verifier staging call) of at least n proofs on this context";
        return -1;
    }
    HIPCHK(hipSetDevice(c.device));
    if (ensure_verify_workspace(c)) return -1;
    if (pk_mode) {
        const Params &Pk = c.P;
        if (pk_mode == 1) {
            int first = 0;
            for (const VerifySeg *s = segs; s; s = s->next) {
                if (is_device_pointer(s->pk)) {
                    HIPCHK(hipMemcpy2DAsync(c.d_pk + (size_t)first * c.pk_stride, c.pk_stride, s->pk, Pk.pk_bytes, Pk.pk_bytes, s->count,
                                            hipMemcpyDeviceToDevice, c.stream));
                } else {
                    for (int b = 0; b < s->count; b++)
                        memcpy(c.h_pk + (size_t)(first + b) * c.pk_stride, s->pk + (size_t)b * Pk.pk_bytes, Pk.pk_bytes);
                    HIPCHK(hipMemcpyAsync(c.d_pk + (size_t)first * c.pk_stride, c.h_pk + (size_t)first * c.pk_stride, (size_t)s->count * c.pk_stride,
                                          hipMemcpyHostToDevice, c.stream));
                }
                first += s->count;
            }
            c.resident_pk_n = n;
        }
        // the decoding itself (polyvec_frombytes + gen_matrix) is issued with segment V1B: nothing before needs A or t, and
        // there it runs while the host hashes instead of in front of the first digests
    }
    const Params &P = c.P;
    const RowMap &rm = c.rm;
    const int K = P.K;
    hipStream_t st = c.stream;

    double t0 = now_sec(), t1;
    VerifyArgs va{};
    OpenedHashArgs oh{};
    va.P = c.d_P;
    va.proof_stride = c.proof_stride;
    va.O = c.d_O;
    va.o_stride = c.o_stride;
    va.rm = rm;
    va.eta1 = P.eta1;
    va.opened = c.d_I;
    va.rest = c.d_rest;
    va.sel_stride = c.sel_stride;
    va.fail = c.d_fail;
    va.strict = c.strict_encoding ? 1 : 0;
    oh.proof = c.d_proof;
    oh.image_stride = c.image_stride;
    oh.off_s = (uint32_t)P.off[F_S]; oh.off_e = (uint32_t)P.off[F_E]; oh.off_f = (uint32_t)P.off[F_F];
    oh.off_nttf = (uint32_t)P.off[F_NTTF]; oh.off_zs = (uint32_t)P.off[F_ZS]; oh.off_ze = (uint32_t)P.off[F_ZE];
    oh.P = c.d_P;
    oh.proof_stride = c.proof_stride;
    oh.O = c.d_O;
    oh.o_stride = c.o_stride;
    oh.rm = rm;
    oh.opened = c.d_I;
    oh.sel_stride = c.sel_stride;
    oh.prefix = nullptr;
    oh.out = c.d_dig1;
    // how the host gets its two digest tables (see Ctx::d_odig): with the caller's host copy of the images at hand only the 150 recomputed
    // digests per proof follow each round's hash, else the whole tables.  Device Fiat-Shamir: both tables are put together in HBM anyway
    // (k_disassemble_fields copies the images' 1304 digests per table, the opened-party hashes store their 150), hashed there, and the
    // host sees nothing of them
    const bool split_tables = !c.fs_device && himg != nullptr;
    oh.out_compact = split_tables ? c.d_odig : nullptr;
    const size_t dig_bytes = split_tables ? (size_t)n * NOPEN * 32 : (size_t)n * NPARTY * 32;
    // the table of proof b for the host's hash of round r, put together on the worker that hashes it
    auto table_prep = [&](int r) {
        return std::function<void(int)>([&c, himg, himg_stride, r](int b) {
            const uint8_t *unopened = himg + (size_t)b * himg_stride + c.P.off[r ? F_COMM : F_TCOMM];
            assemble_digest_table(c.h_dig + (size_t)b * NPARTY * 32, c.h_Iimg + (size_t)b * NOPEN, unopened, c.h_odig + (size_t)b * NOPEN * 32);
        });
    };
    if (run_segment(c, Ctx::SEG_V1, n, [&]() -> int {
    // ---- V0: opened list from the image, validated and expanded on the GPU (no host round trip); the host
    // only needs the list itself for the final Fiat-Shamir comparison and receives it with the first digests
    HIPCHK(launch_opened_setup(c.d_proof, c.image_stride, P.off[F_I], c.d_I, c.d_rest, c.d_isort, c.d_hrange, c.sel_stride, c.d_fail, n, st));
    if (!c.fs_device) HIPCHK(copy_small(c, c.h_Iimg, 2 * NOPEN, c.d_proof + P.off[F_I], c.image_stride, 2 * NOPEN, n, hipMemcpyDeviceToHost, st));


    // ---- V1: scatter, gate outputs on opened columns, Tcomm of the opened parties
    const GateOffsets go{(uint32_t)P.off[F_SSUB], (uint32_t)P.off[F_ESUB], (uint32_t)P.off[F_ZS], (uint32_t)P.off[F_ZE]};
    HIPCHK(launch_disassemble(va, c.d_vfields, c.vplan, c.d_vrowtab, c.d_proof, c.image_stride, P.off[F_TCOMM],
                              P.off[F_COMM], c.d_dig1, c.d_dig2, go, n, st)); // + digests + gate outputs of the opened parties
    c.prof_begin(PR_V_HASH_TCOMM, n);
    HIPCHK(launch_opened_hash(oh, K, false, n, st)); // Tcomm of the opened parties, read from the image   :22-35
    c.prof_end(PR_V_HASH_TCOMM);
    if (c.fs_device) return 0;
    if (split_tables) HIPCHK(hipMemcpyAsync(c.h_odig, c.d_odig, dig_bytes, hipMemcpyDeviceToHost, st));
    else HIPCHK(copy_round_table(c, c.h_dig, c.d_dig1, n));
    return 0;
    }, c.fs_device ? nullptr : split_tables ? c.h_odig : c.h_dig)) return -1; // which table copy the captured segment holds is part of its graph's key
    HIPCHK(hipEventRecord(c.ev, st)); // the opened parties' Tcomm digests are on the host (device Fiat-Shamir: the table is complete in HBM) once this event has passed
    if (c.fs_device) {
        // ---- alpha on the device from the verifier's own table   mlwe_verifier.cpp:37-65
        FsArgs fa{};
        fa.in = c.d_dig1; fa.in_stride = (size_t)NPARTY * 32; fa.len = NPARTY * 32;
        fa.alpha = c.d_alpha; fa.alpha_stride = 80; fa.J = P.J;
        c.prof_begin(PR_V_FS_ALPHA, n);
        HIPCHK(launch_fs_chain(fa, FS_ALPHA, n, st));
        c.prof_end(PR_V_FS_ALPHA);
        c.path_n[PATH_FS_DEVICE]++;
    }
    t1 = now_sec(); c.phase_sec[PH_V1_ISSUE] = t1 - t0; t0 = t1;

    // ---- alpha-independent GPU work, issued before the host hashes so that it runs meanwhile: interpolation of
    // the unopened shares
    if (run_segment(c, Ctx::SEG_V1B, n, [&]() -> int {
    if (pk_mode) HIPCHK(launch_decode_pk(c.d_pk, c.pk_stride, c.d_t, c.d_A, c.key_stride, K, n, st, c.xof_guard())); // kosk.cpp:94-99
    InterpArgs ia{};
    ia.rest = c.d_rest;
    ia.isort = c.d_isort;
    ia.hrange = c.d_hrange;
    ia.sel_stride = c.sel_stride;
    ia.inv = c.d_inv;
    ia.invlimb = c.d_invlimb;
    ia.fact = c.d_fact;
    ia.invfact = c.d_invfact;
    ia.w = c.d_w;
    ia.ell = c.d_ell;
    ia.node_of = c.d_node_of;
    c.prof_begin(PR_V_INTERP_BUILD, n);
    HIPCHK(launch_interp_setup(ia, n, st));
    c.prof_end(PR_V_INTERP_BUILD);
    HIPCHK(launch_gather_frags(c.d_P, c.proof_stride, c.d_rows_isrc, c.n_interp_d, c.d_gather, c.d_rows_u, c.n_interp_2d, c.d_gather2,
                               c.d_rest, c.sel_stride, c.d_w, n, st));
    { // values at points 0..406 of every interpolated sharing (:201-219 etc.) and the 813-node Cauchy sums of the u shares
      // at the packed positions (:523-543) with the per-proof operators built on the fly; recon_secrets_2ddeg of the merged
      // u rows (:555-556) is a product with a fixed table
        c.prof_begin(PR_V_GEMM_INTERP, n);
        HIPCHK(launch_interp_apply(ia, c.d_P, c.proof_stride, c.d_rows_isrc, c.d_rows_idst, c.n_interp_d, c.d_gather, c.n_interp_2d,
                                   c.d_gather2, c.d_sec_u1, n, st));
        c.prof_end(PR_V_GEMM_INTERP);
        const GemmSrc gs3{c.d_P, c.proof_stride, c.d_rows_u, RS, NSEC, DEG2 + 1};
        const GemmDst gd3{c.d_sec_u2, (size_t)c.n_interp_2d * 256, nullptr, 256, 0};
        if (gemm_modq(c, c.t_recon_2d, gs3, gd3, c.n_interp_2d, n)) return -1;
        const GemmSrc xs{c.d_P, c.proof_stride, c.d_rows_idst, RS, 0, XLEN};
        const GemmDst xd{c.d_P, c.proof_stride, c.d_rows_idst, RS, EXP_OFF};
        c.prof_begin(PR_V_GEMM_EXPAND, n);
        if (gemm_modq(c, c.t_expand, xs, xd, c.n_interp_d, n)) return -1; // recompute_share_secrets_ddeg   :224-225, :351, :441-442
        c.prof_end(PR_V_GEMM_EXPAND);
    }
    HIPCHK(launch_check_batch(va, c.d_t, c.d_sec_u1, c.d_sec_u2, c.n_interp_2d, n, st));
    // NTT(s+r), NTT(e+r), A(s+r) and their re-sharing depend only on the interpolated rows   :257-271, :287-301
    NttArgs na{};
    na.in = reinterpret_cast<const int16_t *>(c.d_P);
    na.in_gstride = c.proof_stride;
    na.src_off = c.d_off + c.off_sr_er;
    na.npg = 2 * K;
    na.npoly = 2 * K * n;
    na.out = reinterpret_cast<int16_t *>(c.d_P);
    na.out_gstride = c.proof_stride;
    na.dst_off = c.d_off + c.off_nttsr_er;
    na.out_canonical = 1;
    HIPCHK(launch_relation_ntt(na, c.d_A, c.key_stride, c.d_P, c.proof_stride, rm, n, st));
    {
        const GemmSrc xs{c.d_P, c.proof_stride, c.d_gemm2_rows, RS, 0, XLEN};
        const GemmDst xd{c.d_P, c.proof_stride, c.d_gemm2_rows, RS, EXP_OFF};
        if (gemm_modq(c, c.t_expand, xs, xd, c.n_gemm2, n)) return -1;
    }
    return 0;
    })) return -1;

    auto any_hook = [&]() {
        bool h = c.round_hook != nullptr;
        for (const VerifySeg *sg = segs; sg; sg = sg->next) h |= sg->hook != nullptr;
        return h;
    };
    const bool hooks = c.fs_device && any_hook();
    if (!c.fs_device || hooks) HIPCHK(wait_event(c, c.ev, 3, n));
    t1 = now_sec(); c.phase_sec[PH_V1_WAIT] = t1 - t0; t0 = t1;
    auto fire_hooks = [&](int rnd, const uint8_t *d_table) { // as in prove_resident: per caller of a merged run, else the context's own
        bool seg_hooks = false;
        for (const VerifySeg *s = segs; s; s = s->next) seg_hooks |= s->hook != nullptr;
        if (!seg_hooks) {
            if (c.round_hook) c.round_hook(c.round_user, 1, rnd, d_table, (size_t)n * NPARTY * 32);
            return;
        }
        int first = 0;
        for (const VerifySeg *s = segs; s; s = s->next) {
            if (s->hook) s->hook(s->hook_user, 1, rnd, d_table + (size_t)first * NPARTY * 32, (size_t)s->count * NPARTY * 32);
            first += s->count;
        }
    };
    if (!c.fs_device || hooks) fire_hooks(0, c.d_dig1);

    // ---- host: alpha while the GPU works
    if (!c.fs_device) {
        c.path_n[PATH_FS_HOST]++;
        const std::function<void(int)> prep = table_prep(0);
        fs_alpha_batch(P, n, c.h_dig, (size_t)NPARTY * 32, c.h_alpha, 80, c.nthreads, c.pool, split_tables ? &prep : nullptr);
    }
    t1 = now_sec(); c.phase_sec[PH_V_FS_ALPHA] = t1 - t0; t0 = t1;
    if (run_segment(c, Ctx::SEG_V2, n, [&]() -> int {
    // host mode: read from the page-locked host table by k_pow_table itself (as the prover's k_coef_limbs); device mode: from HBM
    const uint16_t *alpha_src = c.fs_device ? c.d_alpha : c.h_alpha;

    // ---- V2/V3: beta, gamma, r, NTT_r on the opened columns; reconstruction and NTT check
    HIPCHK(launch_pow_table(alpha_src, P.J, P.M, c.d_pwT, n, st));
    LincombArgs la{};
    la.P = c.d_P;
    la.proof_stride = c.proof_stride;
    la.O = c.d_O;
    la.o_stride = c.o_stride;
    la.rm = rm;
    la.J = P.J;
    la.pwT = c.d_pwT;
    la.ncols = NOPEN;
    la.col_map = c.d_I;
    la.col_map_stride = c.sel_stride;
    c.prof_begin(PR_V_LINCOMB, n);
    HIPCHK(launch_lincomb(la, n, st));
    c.prof_end(PR_V_LINCOMB);
    return 0;
    })) return -1;

    // ---- V10: view hashes of the opened parties (plain launch: HIP events can bracket it)
    oh.prefix = c.d_dig1;
    oh.out = c.d_dig2;
    c.prof_begin(PR_V_HASH_VIEW, n);
    HIPCHK(launch_opened_hash(oh, K, true, n, st));
    c.prof_end(PR_V_HASH_VIEW);
    if (c.fs_device) {
        // ---- I' on the device, compared with the proof's own list: fail bit FB_OPENED_SET   mlwe_verifier.cpp:634-683
        HIPCHK(hipEventRecord(c.ev, st));
        FsArgs fa{};
        fa.in = c.d_dig2; fa.in_stride = (size_t)NPARTY * 32; fa.len = NPARTY * 32;
        fa.proof = c.d_proof; fa.image_stride = c.image_stride; fa.off_I = (uint32_t)P.off[F_I]; fa.fail = c.d_fail;
        c.prof_begin(PR_V_FS_OPENED, n);
        HIPCHK(launch_fs_chain(fa, FS_CHECK, n, st));
        c.prof_end(PR_V_FS_OPENED);
        c.path_n[PATH_FS_DEVICE]++;
    } else {
        if (split_tables) HIPCHK(hipMemcpyAsync(c.h_odig, c.d_odig, dig_bytes, hipMemcpyDeviceToHost, st));
        else HIPCHK(copy_round_table(c, c.h_dig, c.d_dig2, n));
        HIPCHK(hipEventRecord(c.ev, st));
    }
    t1 = now_sec(); c.phase_sec[PH_V2_ISSUE] = t1 - t0; t0 = t1;

    // ---- V2B: the checks that feed no hash run while the host derives the opened set: reconstruction of the 140
    // beta/gamma secrets with the NTT comparison (:106-131) and the relation checks on the opened columns
    if (run_segment(c, Ctx::SEG_V2B, n, [&]() -> int {
    NttArgs na{};
    {
        const GemmSrc gs{c.d_P, c.proof_stride, c.d_rows_bg, RS, NSEC, XLEN};
        const GemmDst gd{c.d_sec, (size_t)2 * NCHK * 256, nullptr, 256, 0};
        c.prof_begin(PR_V_GEMM_RECON, n);
        if (gemm_modq(c, c.t_recon_d, gs, gd, 2 * NCHK, n)) return -1; // recon_secrets_ddeg x 140   :106-107
        c.prof_end(PR_V_GEMM_RECON);
    }
    na = NttArgs{};
    na.in = reinterpret_cast<const int16_t *>(c.d_sec);
    na.in_gstride = (size_t)2 * NCHK * 256;
    na.src_off = nullptr;
    na.out = reinterpret_cast<int16_t *>(c.d_sec);
    na.out_gstride = (size_t)2 * NCHK * 256;
    na.dst_off = nullptr;
    na.npg = NCHK;
    na.npoly = NCHK * n;
    na.out_canonical = 1;
    na.cmp_fail = c.d_fail; // NTT(beta_j) is compared with gamma_j (70 polynomials further) as it is produced
    na.cmp_delta = NCHK * 256;
    na.cmp_bit = FB_BETA_GAMMA;
    HIPCHK(launch_ntt(na, st));
    HIPCHK(launch_check_opened(va, n, st));
    HIPCHK(copy_small(c, c.h_fail, 0, c.d_fail, 0, sizeof(uint32_t) * n, 1, hipMemcpyDeviceToHost, st));
    return 0;
    })) return -1;

    if (c.fs_device) {
        if (hooks) {
            HIPCHK(wait_event(c, c.ev, 4, n));
            fire_hooks(1, c.d_dig2);
        }
        if (c.near_end_hook) c.near_end_hook();
        HIPCHK(stream_sync_site(c, 5, n)); // the only wait of the call: fail masks of V2B, the chain's bit among them
        c.prof_collect();
        if (device_error_check(c)) return -1;
        int b = 0;
        for (const VerifySeg *sg = segs; sg; sg = sg->next)
            for (int i = 0; i < sg->count; i++, b++) sg->ok[i] = c.h_fail[b] == 0;
        c.phase_sec[PH_V2_WAIT] = now_sec() - t0;
        c.phase_sec[PH_V_FS_OPEN] = 0;
        return 0;
    }
    HIPCHK(wait_event(c, c.ev, 4, n)); // the view digests are on the host; V2B keeps running
    t1 = now_sec(); c.phase_sec[PH_V2_WAIT] = t1 - t0; t0 = t1;
    fire_hooks(1, c.d_dig2);
    if (c.near_end_hook) c.near_end_hook(); // only the host's last round is left: a merged run's sleeping callers get ready for the return
    if (c.v_I2.size() < (size_t)n * c.sel_stride) { c.v_I2.resize((size_t)n * c.sel_stride); c.v_rest2.resize((size_t)n * c.sel_stride); }
    std::vector<uint16_t> &I2 = c.v_I2, &rest2 = c.v_rest2; // every entry that is read below is written by fs_opened_batch first
    {
        const std::function<void(int)> prep = table_prep(1);
        fs_opened_batch(n, c.h_dig, (size_t)NPARTY * 32, I2.data(), rest2.data(), c.sel_stride, c.nthreads, c.pool, false, split_tables ? &prep : nullptr);
    }
    HIPCHK(stream_sync_site(c, 5, n)); // fail masks of V2B
    c.prof_collect();
    if (device_error_check(c)) return -1; // gen_matrix of a public key hit its block limit: no verdict on these proofs
    {
        int b = 0;
        for (const VerifySeg *s = segs; s; s = s->next)
            for (int i = 0; i < s->count; i++, b++) {
                uint32_t f = c.h_fail[b];
                if (memcmp(&I2[(size_t)b * c.sel_stride], c.h_Iimg + (size_t)b * NOPEN, sizeof(uint16_t) * NOPEN) != 0) f |= 1u << FB_OPENED_SET;
                c.h_fail[b] = f;
                s->ok[i] = f == 0;
            }
    }
    c.phase_sec[PH_V_FS_OPEN] = now_sec() - t0;
    return 0;
}

} // namespace kosk
