// ref_layout.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// sizeof / offsetof of the reference's own structs, compiled against the reference's headers where they lie
// (kosk.hpp:13-16, ss.hpp:33-42, mlwe_prover.hpp:34-94) into oracle/_ref/libkosk_ref_k{2,3,4}.so.  Nothing of the
// reference is restated here: the numbers come out of its headers.  tests/test_oracle_vs_ref.py compares them with
// the layouts the product uses (include/kosk_mi355x.h, include/kosk_compat.hpp, csrc/kosk_params.hpp).
#include <cstddef>

#include "kosk.hpp"

namespace {
struct Entry {
    const char *name;
    size_t value;
};
#define SZ(T) {"sizeof " #T, sizeof(T)}
#define OFF(T, m) {"offsetof " #T "." #m, offsetof(T, m)}
const Entry kEntries[] = {
    {"KYBER_K", KYBER_K}, {"KYBER_ETA1", KYBER_ETA1}, {"MPCITH_N", MPCITH_N}, {"MPCITH_T", MPCITH_T}, {"MPCITH_K", MPCITH_K},
    {"MPCITH_V", MPCITH_V}, {"DEG_D", DEG_D}, {"DEG_2D", DEG_2D}, {"MPCITH_PROOF_SIZE", MPCITH_PROOF_SIZE},
    {"MPCITH_PRE_RANDOMNESS_SIZE", MPCITH_PRE_RANDOMNESS_SIZE},
    SZ(share_vec), OFF(share_vec, share_x), OFF(share_vec, share_y), SZ(secret_vec), OFF(secret_vec, secret),
    SZ(kyber_keypair), OFF(kyber_keypair, pk), OFF(kyber_keypair, sk),
    SZ(mlwe_inst), OFF(mlwe_inst, A), OFF(mlwe_inst, t), OFF(mlwe_inst, s), OFF(mlwe_inst, e),
    SZ(mpcith_randomness), OFF(mpcith_randomness, f), OFF(mpcith_randomness, NTT_f), OFF(mpcith_randomness, f_shares),
    OFF(mpcith_randomness, NTT_f_shares),
    SZ(mpcith_range_proof), OFF(mpcith_range_proof, s_eta_shares), OFF(mpcith_range_proof, e_eta_shares),
    SZ(mpcith_proof),
    OFF(mpcith_proof, f_shares), OFF(mpcith_proof, NTT_f_shares), OFF(mpcith_proof, beta_shares), OFF(mpcith_proof, gamma_shares),
    OFF(mpcith_proof, Tcomm), OFF(mpcith_proof, I), OFF(mpcith_proof, s_shares), OFF(mpcith_proof, e_shares), OFF(mpcith_proof, t_shares),
    OFF(mpcith_proof, NTT_s_shares), OFF(mpcith_proof, NTT_e_shares), OFF(mpcith_proof, NTT_Ar_shares), OFF(mpcith_proof, NTT_As_shares),
    OFF(mpcith_proof, sr_shares), OFF(mpcith_proof, er_shares), OFF(mpcith_proof, s_eta_shares), OFF(mpcith_proof, e_eta_shares),
    OFF(mpcith_proof, s_sub_eta_shares), OFF(mpcith_proof, e_sub_eta_shares), OFF(mpcith_proof, z_s_ddeg_shares),
    OFF(mpcith_proof, z_e_ddeg_shares), OFF(mpcith_proof, u_s_2ddeg_shares), OFF(mpcith_proof, u_e_2ddeg_shares), OFF(mpcith_proof, comm),
    SZ(mpcith_vp_state), OFF(mpcith_vp_state, comm), OFF(mpcith_vp_state, s_sh), OFF(mpcith_vp_state, e_sh), OFF(mpcith_vp_state, f_sh),
    OFF(mpcith_vp_state, Tf_sh), OFF(mpcith_vp_state, beta), OFF(mpcith_vp_state, gamma), OFF(mpcith_vp_state, sr_sh),
    OFF(mpcith_vp_state, er_sh), OFF(mpcith_vp_state, s_ddeg_sh), OFF(mpcith_vp_state, e_ddeg_sh), OFF(mpcith_vp_state, s_zero_sh),
    OFF(mpcith_vp_state, e_zero_sh),
};
} // namespace

extern "C" {
int ref_layout_count(void) { return (int)(sizeof kEntries / sizeof kEntries[0]); }
const char *ref_layout_name(int i) { return kEntries[i].name; }
size_t ref_layout_value(int i) { return kEntries[i].value; }
}
