// Kernel argument blocks and launcher prototypes (kosk_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "kosk_params.hpp"

namespace kosk {

// gen_matrix's rejection sampling squeezes SHAKE128 blocks until it has 256 coefficients (indcpa.c:124-145 loops without a
// bound; three blocks suffice with probability 1 - 2^-40).  A GPU loop needs an exit every wave reaches: after max_blocks
// blocks the sampler stops, zero-fills the rest and raises *err -- a word in page-locked HOST memory that the host checks after
// the call's last synchronisation: the call then fails with rc -1 (never a silently wrong matrix, never an abort).
struct XofGuard {
    uint32_t *err = nullptr;
    int max_blocks = 32;
};
enum : uint32_t { DEVERR_XOF_BLOCKS = 1u };


// K4: lane `l` of group `g` hashes  [prefix(32 B)] || rows[g][r][col_off + col(l)], r = 0..NROWS-1
struct HashArgs {
    const uint16_t *rows;      // row 0 of group 0
    size_t group_stride;       // u16 between groups
    int row_stride;            // u16 between rows
    int col_off;               // column of lane 0 (NSEC for party lanes)
    int lanes_per_group;
    const uint16_t *lane_map;  // optional: lane -> party (opened list), else identity
    int lane_map_stride;
    const uint8_t *prefix;     // [group][out_lanes_per_group][32], read at the output index
    uint8_t *out;              // [group][out_lanes_per_group][32]
    int out_lanes_per_group;
};

struct NttArgs {
    const int16_t *in;
    const int32_t *src_off; // per-polynomial offset (u16 units) inside a group, or null: i*256
    size_t in_gstride;
    int16_t *out;
    const int32_t *dst_off;
    size_t out_gstride;
    int npg;   // polynomials per group
    int npoly; // total
    int out_canonical; // 1: [0,q) (encode_to_gf3329), 0: centred int16 (poly_ntt)
    // compare mode (cmp_fail != null): nothing is stored; the result is compared with the polynomial cmp_delta u16 after its
    // output position and bit cmp_bit of cmp_fail[group] is set on a mismatch (NTT(beta_j) == gamma_j, mlwe_verifier.cpp:110-124)
    uint32_t *cmp_fail;
    int cmp_delta, cmp_bit;
    uint32_t npg_magic; // floor(2^32 / npg), filled in by the launchers: polynomial -> (group, index) without an integer division
};
inline uint32_t ntt_npg_magic(int npg) { return npg <= 1 ? 0xFFFFFFFFu : (uint32_t)((1ull << 32) / (uint32_t)npg); }

// ---- "limb matrix": the MFMA operand format of the mod-q GEMM -------------------------------
// A matrix X[r][k] of field elements is stored as two int8 limbs of its centred representative
// c = c0 + 64 c1 (c0 in [-32,31], c1 in [-26,26]) in 16-row x 64-k tiles of 1 KiB each:
//   byte(r, k, limb) = ((kstep * RT + r/16) * 2 + limb) * 1024 + (r%16)*64 + ((k%64/16) ^ swz(r%16))*16 + k%16
// kstep = k/64, RT = row tiles.  swz(r) = (-(r>>2)) & 3 makes the 16-byte fragment reads of
// v_mfma_i32_16x16x64_i8 (lane l: row l&15, k-chunk l>>4) conflict-free for ds_read_b128.
KOSK_HD inline int limb_swz(int r) { return (-(r >> 2)) & 3; }
KOSK_HD inline size_t limb_offset(int r, int k, int limb, int RT)
{
    const int ks = k >> 6, kc = (k >> 4) & 3, rr = r & 15;
    return ((size_t)(ks * RT + (r >> 4)) * 2 + limb) * 1024 + rr * 64 + ((kc ^ limb_swz(rr)) << 4) + (k & 15);
}
KOSK_HD inline void limb_split(int32_t centred, int &c0, int &c1)
{
    c0 = ((centred + 32) & 63) - 32;
    c1 = (centred - c0) >> 6;
}

// rows of canonical u16 -> limb matrix (k_rows_to_limbs)
struct LimbArgs {
    const uint16_t *src;
    size_t src_gstride;    // u16 between groups
    const int16_t *rows;   // row index per i (null: i)
    int src_rstride;       // u16 between rows
    int src_koff;
    int ncols;             // valid k (rest zero), padded to KS*64
    int KS;
    uint8_t *dst;
    int RT;                // row tiles of the destination (= total rows / 16)
    int npg, npg_pad, ngroups; // destination row = g*npg_pad + i
};

// C[n][c_off + m] = sum_k A[m][k] * B[n][k] mod q, both operands limb matrices
struct GemmArgs {
    const uint8_t *A;  // table limb matrix, RT = Mpad/16
    const uint8_t *Afrag; // the same table with fragment-linear tiles (k_table_gemm), or null
    size_t a_gstride;  // bytes between per-group operands (grouped mode)
    int Mpad, M, KS;   // Mpad multiple of 128; m < M is stored
    const uint8_t *B;  // data operand as a limb matrix, rows n = g*npg_pad + i (null: convert from `src` on the fly)
    int BRT;           // row tiles of B
    // data operand as canonical u16 rows, k contiguous: row (g, i) at src + g*src_gstride + src_rows[i]*src_rstride + src_koff
    const uint16_t *src;
    size_t src_gstride;
    const int16_t *src_rows; // null: i
    int src_rstride, src_koff;
    int src_canonical; // 1: every source value is < q (no folding while converting to limbs)
    uint16_t *C;
    size_t c_gstride;
    const int16_t *c_rows; // output row per i (null: i)
    int c_rstride;
    int c_off;
    int npg, npg_pad, ngroups;
    int grouped; // 1: A + g*a_gstride (npg_pad must be a multiple of 64)
    int c_gdiv;  // groups per output block: C + (g / c_gdiv) * c_gstride, rows from c_rows + (g % c_gdiv) * c_rows_gstride
    int c_rows_gstride;
};

struct LincombArgs {
    uint16_t *P;
    size_t proof_stride;
    uint16_t *O;      // opened matrix (verifier): inputs and outputs live there; beta / gamma of parties < 407 also go to P
    size_t o_stride;
    RowMap rm;
    int J;
    const int32_t *pwT; // [proof][MAXM][80]
    int ncols;
    const uint16_t *col_map; // optional party list (verifier: opened parties)
    int col_map_stride;
    int nxb, njc, ngroups; // grid decomposition (set by launch_lincomb)
};

struct FieldDesc {
    uint32_t off;
    int sel;   // 0: opened parties in I order, 1: unopened ascending
    int width; // u16 per party
    int rowtab_off;
    // verifier, fields of unopened parties: which records the reference's verify() ever reads -- record i (party p = rest[i]) is
    // read iff (limit_by_party ? p : i) < limit.  What it never reads is not range-checked either (any bytes there are accepted
    // by the reference, mlwe_verifier.cpp:106-107, :321-323, :390-394, :503-507).  0 = every record.
    int limit, limit_by_party;
    // verifier, default (reference-following) mode: what a u16 element >= q in a READ record of this field does.  -1: nothing --
    // the reference only ever reduces it (gf3329_mul's `% 3329`, gf3329.c:282-284; NTL's conversion to ZZ_p) or emulating its
    // arithmetic on the raw value is the consumer's job (opened-party fields, kept raw in the opened matrix); otherwise the fail
    // bit of the raw comparison the reference makes against a canonical value (s + r / e + r shares, mlwe_verifier.cpp:232-246).
    int noncanon_bit;
};

// which (field, chunk) a block of the assemble / disassemble kernels works on: blocks [0, nrest * NWIN) walk the
// fields of unopened parties window by window, the rest the fields of opened parties in chunks of 64
struct FieldPlan {
    uint8_t rest_ids[NFIELDS], open_ids[NFIELDS];
    int nrest, nopen;
};
inline FieldPlan make_field_plan(const FieldDesc *fields, int nfields)
{
    FieldPlan p{};
    for (int f = 0; f < nfields; f++) {
        if (fields[f].sel) p.rest_ids[p.nrest++] = (uint8_t)f;
        else p.open_ids[p.nopen++] = (uint8_t)f;
    }
    return p;
}

// (round 5) the grouped image kernel: a block = (group of image fields of one kind, aligned window of 64 party columns).  A group's rows
// are the concatenated rows of its fields (<= 80); element e of the group belongs to one field: AsmElem says where it goes.
//   fields of UNOPENED parties: the wave gathers at the window's unopened columns (one 128-byte line per row) and keeps every field's
//     [party][width] block contiguous in its LDS tile (tile = 64 field_col + party width + k), so each field is written out as one run;
//   fields of OPENED parties: the wave gathers the window's 64 columns DENSELY (the same single line per row; the one-shot kernel
//     gathered 64 entries of I per wave: 64 scattered columns = about 20 lines per row and instruction), keeps [party][rows] in the
//     tile, and writes the records of the window's opened parties (6.6 on average) to their positions in the list I.
struct AsmElem {
    int32_t dst;    // u16 index inside the proof image of this element of party record 0 (field offset / 2 + k)
    int16_t width;  // u16 per party of the element's field
    int16_t tile;   // unopened kind: 64 field_col + k (the tile index of party 0's element)
};
struct AsmGroup {
    int sel;        // 0 opened parties, 1 unopened
    int nrows;      // rows = elements of the group (<= 80)
    int rowtab_off; // first entry in the row table
    int elem_off;   // first entry in the AsmElem table
    int nsub;
    uint32_t sub_off[12]; // image byte offset, width and first tile column of the group's fields (unopened kind: the write-out runs)
    int16_t sub_width[12], sub_col[12];
};
constexpr int ASM_MAX_GROUPS = 8;

struct AssembleArgs {
    const uint16_t *P;
    size_t proof_stride;
    const int16_t *rowtab;
    const AsmGroup *groups;
    const AsmElem *elems;
    int ngroups;
    const uint16_t *opened, *rest; // [proof][sel_stride]
    int sel_stride;
    const uint8_t *dig1, *dig2; // [proof][NPARTY][32]
    uint8_t *proof;
    size_t image_stride;
    int img_align; // filled in by the launcher: 16, 8, 4 or 2 -- the widest store the digest blocks may use on the image
};

// ---- verifier (kosk_verify_kernels.hip) ----
enum FailBit {
    FB_MALFORMED = 0, FB_BETA_GAMMA, FB_SR_ER_SHARES, FB_NTT_S_E, FB_A_SR, FB_T_PK, FB_T_RELATION,
    FB_ETA_CONST, FB_SUB_ETA, FB_U_INTERP, FB_U_RECON, FB_OPENED_SET
};

// image offsets of the opened parties' s - eta, e - eta, z_s, z_e records (fields 17-20 of mpcith_proof)
struct GateOffsets {
    uint32_t ssub, esub, zs, ze;
};

struct VerifyArgs {
    uint16_t *P;
    size_t proof_stride;
    uint16_t *O;      // opened matrix [proof][row][OS]
    size_t o_stride;
    RowMap rm;
    int eta1;
    const uint16_t *opened, *rest; // [proof][sel_stride]
    int sel_stride;
    uint32_t *fail; // [proof]
    int strict;     // KOSK_STRICT_ENCODING=1: a u16 element >= q in any record the reference reads is FB_MALFORMED (rounds 1-4)
};

// The reference's field operations on RAW u16 operands (utils/gf3329.c:274-280): the sum / difference is formed in int and
// truncated to uint16_t on return, with ONE conditional correction by q -- for canonical operands the canonical result, for an
// operand >= q (only a crafted proof holds one) whatever that arithmetic gives.  The verifier's comparisons on opened-party
// records (mlwe_verifier.cpp:275, :279, :306, :370, :451, :459, :487, :491) are made on these values, so they are reproduced
// bit for bit; everything that goes through gf3329_mul (`% 3329`) is simply folded.
KOSK_HD inline uint32_t ref_add_u16(uint32_t a, uint32_t b) { const int32_t s = (int32_t)a + (int32_t)b; return (uint32_t)(s < Q ? s : s - Q) & 0xFFFFu; }
KOSK_HD inline uint32_t ref_sub_u16(uint32_t a, uint32_t b) { return (uint32_t)(a < b ? (int32_t)a + Q - (int32_t)b : (int32_t)a - (int32_t)b) & 0xFFFFu; }
KOSK_HD inline uint32_t gf_fold(uint32_t x) { return x >= (uint32_t)Q ? x % (uint32_t)Q : x; }

// Interpolation through the nodes x_j = 256 + rest[j] (j < 407 resp. 813), barycentric form
//   p(k) = l(k) * sum_j (w_j y_j) / (k - x_j):
// the y are scaled by w_j when gathered, the GEMM operand is the plain Cauchy matrix 1/(k - x_j)
// (a table lookup per entry), and l(k) is applied afterwards (set 0) or not needed at all (set 1:
// p(k) == 0 <=> the Cauchy sum is 0).  The nodes are "all parties in [lo, hi] except the opened ones in
// between" (holes), so w_j and l(k) are factorials times a product over the <= 150 holes.
struct InterpArgs {
    const uint16_t *rest;    // [proof][sel_stride] unopened parties, ascending
    const uint16_t *isort;   // [proof][sel_stride] opened parties, ascending
    const uint16_t *hrange;  // [proof][4]: hole index range [h0,h1) into isort for set 0, then set 1
    int sel_stride;
    const uint16_t *inv, *fact, *invfact; // field inverses, k!, 1/k!  (k < 3329)
    const uint16_t *invlimb;  // (low limb | high limb << 8) of 1/d at index d + interp_table_off(), see k_interp_apply
    uint16_t *w;             // [proof][2][832] barycentric weights
    uint16_t *ell;           // [proof][416]   l(k) of set 0 (0 where k is a node)
    int16_t *node_of;        // [proof][416]   j if evaluation point k is node x_j, else -1
};

// opened list from the image -> I, complement, sorted I, hole ranges, MALFORMED bit (overwrites fail[])
hipError_t launch_opened_setup(const uint8_t *proof, size_t image_stride, size_t off_I, uint16_t *I, uint16_t *rest, uint16_t *isort,
                               uint16_t *hrange, size_t sel_stride, uint32_t *fail, int nproofs, hipStream_t st);
hipError_t launch_disassemble(const VerifyArgs &v, const FieldDesc *fields, const FieldPlan &plan, const int16_t *rowtab,
                              const uint8_t *proof, size_t image_stride, size_t off_tcomm, size_t off_comm,
                              uint8_t *dig1, uint8_t *dig2, const GateOffsets &go, int nproofs, hipStream_t st);
// Tcomm / view hash of the OPENED parties straight from the proof image (mlwe_verifier.cpp:23-35, :585-632):
// s, e, f, NTT f, z_s, z_e are contiguous per party there; only beta, gamma, s+r, e+r, u come from rows.
struct OpenedHashArgs {
    const uint8_t *proof;
    size_t image_stride;
    uint32_t off_s, off_e, off_f, off_nttf, off_zs, off_ze;
    const uint16_t *P;
    size_t proof_stride;
    const uint16_t *O; // opened matrix: beta, gamma, u of the opened parties
    size_t o_stride;
    RowMap rm;
    const uint16_t *opened;
    int sel_stride;
    const uint8_t *prefix; // Tcomm digest table [proof][NPARTY][32] (view hash only)
    uint8_t *out;          // digest table, written at the party's index
    uint8_t *out_compact;  // optional [proof][NOPEN][32]: the same digests in the order of the list I (what the host still needs)
};
hipError_t launch_opened_hash(const OpenedHashArgs &a, int K, bool view, int nproofs, hipStream_t st);
hipError_t launch_check_batch(const VerifyArgs &v, const uint16_t *t_pk, const uint16_t *u1, const uint16_t *u2, int nu, int nproofs,
                              hipStream_t st);
// weighted shares of both interpolations as MFMA fragment tiles: interp_y_bytes(set) bytes per proof
size_t interp_y_bytes(int set);
int interp_table_len(); // entries of InterpArgs::invlimb
int interp_table_off(); // entry d + off = limbs of 1/d
hipError_t launch_gather_frags(const uint16_t *P, size_t proof_stride, const int16_t *rows1, int nrows1, uint8_t *out1,
                               const int16_t *rows2, int nrows2, uint8_t *out2, const uint16_t *rest, int sel_stride,
                               const uint16_t *w, int nproofs, hipStream_t st);
// weights, l(k), node map (k_interp_setup), then the interpolation itself with the Cauchy operator built in registers
// (k_interp_apply): degree d: P[b][dst_rows[r]][k] = node(k) ? P[b][src_rows[r]][256 + rest[node]] : l(k) sum_j y1[b][r][j] / (k - x_j),
// k < 407; degree 2d: out2[b][r][k] = sum_j y2[b][r][j] / (k - x_j), k < 256
hipError_t launch_interp_setup(const InterpArgs &a, int nproofs, hipStream_t st);
hipError_t launch_interp_apply(const InterpArgs &a, uint16_t *P, size_t proof_stride, const int16_t *src_rows, const int16_t *dst_rows,
                               int n1, const uint8_t *y1, int n2, const uint8_t *y2, uint16_t *out2, int nproofs, hipStream_t st);
hipError_t launch_check_opened(const VerifyArgs &v, int nproofs, hipStream_t st);

// ---- key generation (kosk_keygen_kernels.hip) ----
hipError_t launch_keygen(const uint8_t *tape, size_t tape_stride, uint8_t *seeds, size_t seed_stride, int16_t *A, size_t A_stride,
                         int16_t *se, size_t se_stride, int K, int eta1, int n, hipStream_t st, XofGuard xof = XofGuard());
hipError_t launch_keygen_pack(const int16_t *A, size_t A_stride, const int16_t *sehat, size_t sehat_stride, const uint8_t *seeds,
                              size_t seed_stride, uint16_t *t_out, uint8_t *pk, size_t pk_stride, uint8_t *shat_bytes, size_t sb_stride, int K, int n,
                              hipStream_t st);
hipError_t launch_decode_pk(const uint8_t *pk, size_t pk_stride, uint16_t *t_out, int16_t *A, size_t A_stride, int K, int n, hipStream_t st,
                            XofGuard xof = XofGuard());

// the LDS-DMA staged kernel runs where the layout allows it (aligned rows, no lane map), else the plain kernel; *variant: bit 0 the DMA kernel ran
hipError_t launch_commit_hash(const HashArgs &a, int ngroups, int K, bool view, hipStream_t st, int *variant = nullptr);
hipError_t launch_sha3_msgs(const uint8_t *in, size_t in_stride, int len, uint8_t *out, size_t out_stride,
                            int outlen, int n, int domain, hipStream_t st);
// the same on the lane-pair sponge (32 messages per wave; kosk_keccak_split_dev.hpp)
hipError_t launch_sha3_msgs_pair(const uint8_t *in, size_t in_stride, int len, uint8_t *out, size_t out_stride, int outlen, int n,
                                 int domain, hipStream_t st);
// ---- Fiat-Shamir aggregation on the device (kosk_fs_kernels.hip): one wave per proof hashes the proof's digest table where the
// commitment kernel wrote it and derives the challenge from the digest (mlwe_prover.cpp:130-153, :445-474; mlwe_verifier.cpp:37-65, :634-683)
enum FsMode { FS_DIGEST = 0, FS_ALPHA = 1, FS_OPENED = 2, FS_CHECK = 3 };
struct FsArgs {
    const uint8_t *in;   // [n] messages, in_stride bytes apart; base and stride multiples of 8 (the digest tables: 1454 x 32 bytes per proof)
    size_t in_stride;
    int len;             // bytes hashed per message
    uint8_t *out_digest; // optional [n][32]: sha3_256 of every message (FS_DIGEST: the result; else h1 / ch for whoever wants them)
    // FS_ALPHA: alpha[proof][alpha_stride], J = 70 + 2K entries derived, zeros behind them up to 80
    uint16_t *alpha;
    int alpha_stride, J;
    // FS_OPENED (prover): rows of the opened-list tables (kosk_params.hpp: I, SEL_WIN, SEL_OSORT, SEL_OPOS; the ascending complement)
    uint16_t *I, *rest;
    int sel_stride;
    // FS_CHECK (verifier): the recomputed list against field I of the proof image, bit FB_OPENED_SET of fail[proof] on a mismatch
    const uint8_t *proof;
    size_t image_stride;
    uint32_t off_I;
    uint32_t *fail;
};
hipError_t launch_fs_chain(const FsArgs &A, int mode, int n, hipStream_t st);
bool copy_small_ok(const void *src, size_t src_stride, const void *dst, size_t dst_stride, size_t row_bytes);
hipError_t launch_copy_small(const void *src, size_t src_stride, void *dst, size_t dst_stride, size_t row_bytes, size_t nrows, hipStream_t st);
hipError_t launch_rows_copy(const uint16_t *src, size_t src_stride, uint16_t *dst, size_t dst_stride, int count,
                            int nrows, hipStream_t st);
// expand_f + tape randoms + witness secrets (the kernels that only read the tape / the key) in one launch
// key generation outputs when it runs as roles of the prover's first launch
// the randomness tapes of a merged call whose callers keep theirs in HBM: caller j's tapes start at ptr[j], `per` proofs each
// (the last one may hold fewer), all tape_stride apart -- read in place, no copy into one buffer (count == 0: one buffer)
struct TapeSegs {
    const uint8_t *ptr[16]; // Combiner::MAX_WIDTH callers of a merged run
    int per, count;
};
struct KeygenFront {
    uint8_t *seeds;
    size_t seed_stride;
    int16_t *A;
    size_t A_stride;
    int16_t *se;
    size_t se_stride;
    XofGuard xof;
};
hipError_t launch_prover_pre(const uint8_t *tape, size_t tape_stride, uint16_t *P, size_t proof_stride, int row_f, int M,
                             int slice0_off, const int16_t *fresh_rows, int slice_begin, int slice_end, bool expand_f,
                             int witness_mode, const int16_t *se, size_t se_stride, const RowMap &rm, int eta1, int nproofs,
                             hipStream_t st, const KeygenFront *kg = nullptr, const TapeSegs *segs = nullptr);
hipError_t launch_ntt(const NttArgs &a, hipStream_t st);
// k_ntt256 (this proof's na.npg = 2K polynomials) + k_matvec_ntt(nttsr -> nttasr) + k_copy_tails in one launch
hipError_t launch_relation_ntt(const NttArgs &na, const int16_t *A, size_t A_stride, uint16_t *P, size_t proof_stride, const RowMap &rm,
                               int nproofs, hipStream_t st);
hipError_t launch_matvec_ntt(const int16_t *A, size_t A_stride, uint16_t *P, size_t proof_stride, int v_row0, int row0, int K,
                             int nproofs, hipStream_t st);
hipError_t launch_rows_to_limbs(const LimbArgs &a, hipStream_t st);
hipError_t launch_gemm(const GemmArgs &a, hipStream_t st);
// table products (shared table, 407-wide u16 input rows) with the data rows resident in LDS; `sink` = 4 KiB of scratch
bool table_gemm_usable(const GemmArgs &a);
hipError_t launch_table_gemm(const GemmArgs &a, uint16_t *sink, hipStream_t st);
// K3 on the matrix cores (prover): beta / gamma / r / NTT_r rows of every proof from its f / NTT f rows and the alpha-power coefficient
// matrix (k_coef_limbs), plus s + r and e + r in the epilogue (k_lincomb_stream: persistent workgroups, prefetched inputs)
hipError_t launch_lincomb_stream(const uint16_t *P, size_t proof_stride, const RowMap &rm, const uint8_t *coef, uint16_t *C,
                                 const int16_t *lin_rows, int J, int nproofs, hipStream_t st);
hipError_t launch_coef_limbs(const uint16_t *alpha, int J, int M, uint8_t *B, int nproofs, hipStream_t st);
hipError_t launch_pow_table(const uint16_t *alpha, int J, int M, int32_t *pwT, int nproofs, hipStream_t st);
hipError_t launch_lincomb(const LincombArgs &a, int nproofs, hipStream_t st);
hipError_t launch_post_gates(uint16_t *P, size_t proof_stride, const RowMap &rm, int nproofs, hipStream_t st);
hipError_t launch_copy_tails(uint16_t *P, size_t proof_stride, const RowMap &rm, int nproofs, hipStream_t st);
hipError_t launch_post_relation(uint16_t *P, size_t proof_stride, const RowMap &rm, int nproofs, hipStream_t st);
// the proof wire image (k_assemble_groups: opened-party records from dense window gathers)
hipError_t launch_assemble(const AssembleArgs &a, size_t off_tcomm, size_t off_comm, size_t off_I, int nproofs, hipStream_t st);

} // namespace kosk
