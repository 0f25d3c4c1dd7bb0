// The "kyber verifiable keygen & KOSK" half of the reference's main.cpp (main.cpp:66-94), recompiled against
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
