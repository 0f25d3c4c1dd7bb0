// Both halves of the reference's main.cpp -- the "mlwe prover test" on the second-level API (main.cpp:18-62) and the
// "kyber verifiable keygen & KOSK" part (main.cpp:66-94) -- recompiled against
// include/kosk_compat.hpp instead of the reference's kosk.hpp.  randombytes() here is a SHAKE256 tape
// ("kosk-tape-v1:0") so that the printed digests can be compared with the recorded reference ones.
//   hipcc -std=c++17 -DKYBER_K=3 -Iinclude examples/main_like.cpp -Lmpcith_kyber_kosk_amd -lkosk_mi355x -o main_like
#include <ctime>
#include <vector>

#include "kosk_compat.hpp"

static std::vector<uint8_t> g_tape;
static size_t g_pos = 0;
extern "C" void randombytes(uint8_t *out, size_t outlen)
{
    if (g_pos + outlen > g_tape.size()) abort();
    memcpy(out, g_tape.data() + g_pos, outlen);
    g_pos += outlen;
}

static void hex(const char *name, const uint8_t *d, size_t n)
{
    uint8_t h[32];
    kosk_host_sha3_256(h, d, n);
    printf("%s sha3_256 = ", name);
    for (int i = 0; i < 32; i++) printf("%02x", h[i]);
    printf("\n");
}

int main()
{
    g_tape.resize(kosk_tape_bytes(KYBER_K));
    const char seed[] = "kosk-tape-v1:0";
    kosk_host_shake256(g_tape.data(), g_tape.size(), (const uint8_t *)seed, sizeof(seed) - 1);

    { // main.cpp:18-62: preprocessing, raw keygen, prove, verify on the reference's own structs
        std::vector<uint8_t> saved = g_tape;
        const char seed2[] = "kosk-tape-v1:main-order";
        g_tape.resize(kosk_tape_bytes(KYBER_K));
        kosk_host_shake256(g_tape.data(), g_tape.size(), (const uint8_t *)seed2, sizeof(seed2) - 1);
        g_pos = 0;
        printf("=== mlwe prover test === \n");
        static mpcith_randomness rand_;
        static mpcith_range_proof eta_shares;
        prepare_randomness(&rand_);
        prepare_range_proof(&eta_shares);
        static mlwe_inst raw_sec;
        static kyber_keypair kp;
        kyber_keygen(&kp, &raw_sec);
        static mpcith_proof pi;
        prove(&pi, &raw_sec, &rand_, &eta_shares);
        const bool res = verify(&pi, &raw_sec);
        printf(res ? "[result] mlwe verify success\n" : "[result] mlwe verify failed\n");
        static uint8_t buf[MPCITH_PRE_RANDOMNESS_SIZE];
        encode_preprocessed_randomness(buf, &rand_, &eta_shares);
        hex("pre", buf, sizeof buf);
        hex("pi(main-order)", reinterpret_cast<const uint8_t *>(&pi), sizeof pi);
        printf("[tape] consumed %zu of %zu bytes\n", g_pos, g_tape.size());
        if (!res) return 1;
        g_tape = saved;
        g_pos = 0;
    }
    printf("=== kyber verifiable keygen & KOSK === \n");
    static kyber_keypair keypair;
    static uint8_t kosk_pi[MPCITH_PROOF_SIZE] = {0};
    if (MPCITH_PROOF_SIZE != kosk_proof_bytes(KYBER_K)) { printf("proof size mismatch\n"); return 2; }

    clock_t t0 = clock();
    kyber_verifiable_keygen(&keypair, kosk_pi);
    clock_t t1 = clock();
    printf(">>> kyber verifiable keygen (keygen + preprocess + prove) time used: %f s\n", ((double)(t1 - t0)) / CLOCKS_PER_SEC);
    bool res2 = kyber_kosk_verify(kosk_pi, keypair.pk);
    clock_t t2 = clock();
    printf(res2 ? "[result] kyber kosk verify success\n" : "[result] kyber kosk verify failed\n");
    printf(">>> kyber kosk verify time used: %f s\n", ((double)(t2 - t1)) / CLOCKS_PER_SEC);
    printf("[proof size] %lu kilobytes\n", (unsigned long)(MPCITH_PROOF_SIZE / 1024));
    hex("pk", keypair.pk, sizeof keypair.pk);
    hex("sk", keypair.sk, sizeof keypair.sk);
    hex("pi", kosk_pi, MPCITH_PROOF_SIZE);
    kosk_pi[12345] ^= 1;
    printf("[tampered] verify = %d\n", (int)kyber_kosk_verify(kosk_pi, keypair.pk));
    return res2 ? 0 : 1;
}
