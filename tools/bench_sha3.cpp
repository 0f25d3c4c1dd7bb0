// micro-benchmark of the host's multi-buffer SHA3 (links kosk_host.cpp): ns per 8-message permutation on one core, data in cache
#include "kosk_host.hpp"
#include <chrono>
#include <cstdio>
#include <vector>
#include <cstring>
int main(int argc, char **argv)
{
    const int n = 8; const size_t len = 1454 * 32;
    std::vector<uint8_t> buf(n * len); for (size_t i = 0; i < buf.size(); i++) buf[i] = (uint8_t)(i * 2654435761u >> 11);
    const uint8_t *in[8]; for (int i = 0; i < n; i++) in[i] = buf.data() + i * len;
    uint8_t out[8 * 32];
    kosk::sha3_256_multi(out, in, len, n);
    auto t0 = std::chrono::steady_clock::now();
    const int it = 200;
    for (int k = 0; k < it; k++) kosk::sha3_256_multi(out, in, len, n);
    double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / it;
    printf("8 x %zu bytes: %.1f us per call, %.1f ns per 8-way permutation (343 per message); digest[0..3] %02x%02x%02x%02x\n", len, us, us * 1000 / 343, out[0], out[1], out[2], out[3]);
}
