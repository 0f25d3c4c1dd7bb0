// Probe (not product code): what bounds Keccak-f[1600] on gfx950.
//  (1) issue cost of single VALU opcodes from in-kernel stamps (s_memtime = shader cycles, s_memrealtime = 100 MHz),
//      at 1, 2, 4 and 8 waves per SIMD, so the figures do not depend on an assumed clock;
//  (2) the product's register-resident permutation (kosk_keccak_dev.hpp) back to back, no memory traffic:
//      Keccak-f per second, cycles per wave-instruction, the clock the chip holds.
// Build: hipcc -O3 -std=c++20 --offload-arch=gfx950 -I mpcith_kyber_kosk_amd/csrc tools/probe_keccak.hip -o tools/probe_keccak
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <utility>
#include <vector>

#include "kosk_keccak_dev.hpp"
#include "kosk_keccak_split_dev.hpp"

#define OPS(X)                                                                                                                   \
    X(0, "v_xor_b32 %0, %0, %1") X(1, "v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96") X(2, "v_alignbit_b32 %0, %0, %1, 7")            \
    X(3, "v_alignbit_b32 %0, %0, %1, %2") X(4, "v_bfi_b32 %0, %0, %1, %2") X(5, "v_and_or_b32 %0, %0, %1, %2")                   \
    X(6, "v_perm_b32 %0, %0, %1, %2") X(7, "v_lshl_or_b32 %0, %0, 3, %1") X(8, "v_lshrrev_b32 %0, 5, %0")                        \
    X(9, "v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")                                                  \
    X(10, "v_mov_b32_dpp %0, %1 row_ror:4 row_mask:0xf bank_mask:0xf") X(11, "v_xor_b32_dpp %0, %1, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf") \
    X(12, "v_mad_i32_i24 %0, %0, %1, %2") X(13, "v_mul_lo_u32 %0, %0, %1") X(14, "v_add_u32 %0, %0, %1")                          \
    X(15, "v_lshl_add_u32 %0, %0, 3, %1") X(16, "v_xad_u32 %0, %0, %1, %2") X(17, "v_or3_b32 %0, %0, %1, %2")                     \
    X(18, "v_pk_add_u16 %0, %0, %1") X(19, "v_pk_mul_lo_u16 %0, %0, %1") X(20, "v_pk_mad_u16 %0, %0, %1, %2")                     \
    X(21, "v_fma_f32 %0, %0, %1, %2") X(22, "v_mov_b32 %0, %1")
static const char *kNames[] = {"v_xor_b32", "v_bitop3_b32", "v_alignbit_b32 imm", "v_alignbit_b32 vgpr", "v_bfi_b32", "v_and_or_b32", "v_perm_b32",
                               "v_lshl_or_b32", "v_lshrrev_b32", "v_mov_b32_dpp quad_perm", "v_mov_b32_dpp row_ror", "v_xor_b32_dpp quad_perm",
                               "v_mad_i32_i24", "v_mul_lo_u32", "v_add_u32", "v_lshl_add_u32", "v_xad_u32", "v_or3_b32", "v_pk_add_u16",
                               "v_pk_mul_lo_u16", "v_pk_mad_u16", "v_fma_f32", "v_mov_b32"};
constexpr int NOPS = 23;

struct Stamp {
    unsigned long long cyc, rt;
};

template <int OP>
__global__ __launch_bounds__(64) void k_op(int *out, Stamp *st, int iters)
{
    int x[8], a = threadIdx.x * 3 + 1, b = (threadIdx.x ^ 0x5a) & 31;
    for (int i = 0; i < 8; i++) x[i] = threadIdx.x + i;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int i = 0; i < 8; i++) {
#define X(id, s) \
    if (OP == id) asm volatile(s : "+v"(x[i]) : "v"(a), "v"(b));
                OPS(X)
#undef X
            }
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    int s = 0;
    for (int i = 0; i < 8; i++) s += x[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if (threadIdx.x == 0) st[blockIdx.x] = Stamp{c1 - c0, r1 - r0};
}

__global__ __launch_bounds__(64) void k_keccak(uint32_t *out, Stamp *st, int nperm)
{
    kosk::KState s;
#pragma unroll
    for (int i = 0; i < 25; i++) { s.lo[i] = threadIdx.x * 2654435761u + i; s.hi[i] = blockIdx.x * 40503u + i * 7; }
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
#pragma unroll 1
    for (int p = 0; p < nperm; p++) kosk::keccak_f1600_dev(s);
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < 25; i++) acc ^= s.lo[i] ^ s.hi[i];
    out[blockIdx.x * 64 + threadIdx.x] = acc;
    if (threadIdx.x == 0) st[blockIdx.x] = Stamp{c1 - c0, r1 - r0};
}

// the lane-pair layout (kosk_keccak_split_dev.hpp): state i lives in lanes 2i (low halves) and 2i+1 (high halves)
__global__ __launch_bounds__(64) void k_keccak_split(uint32_t *out, Stamp *st, int nperm)
{
    kosk::KHalf s;
    const bool hi = threadIdx.x & 1;
#pragma unroll
    for (int i = 0; i < 25; i++) s.w[i] = (threadIdx.x * 2654435761u + i) ^ (blockIdx.x * 40503u + i * 7);
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
#pragma unroll 1
    for (int p = 0; p < nperm; p++) kosk::keccak_f1600_split(s, hi);
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < 25; i++) acc ^= s.w[i];
    out[blockIdx.x * 64 + threadIdx.x] = acc;
    if (threadIdx.x == 0) st[blockIdx.x] = Stamp{c1 - c0, r1 - r0};
}

// hybrid launch: blocks [0, nmain) hold 64 states in the one-lane form, blocks from nmain on 32 states in the lane-pair form
__global__ __launch_bounds__(64) void k_keccak_hybrid(uint32_t *out, int nmain, int nperm)
{
    uint32_t acc = 0;
    if ((int)blockIdx.x < nmain) {
        kosk::KState s;
#pragma unroll
        for (int i = 0; i < 25; i++) { s.lo[i] = threadIdx.x * 2654435761u + i; s.hi[i] = blockIdx.x * 40503u + i * 7; }
#pragma unroll 1
        for (int p = 0; p < nperm; p++) kosk::keccak_f1600_dev(s);
#pragma unroll
        for (int i = 0; i < 25; i++) acc ^= s.lo[i] ^ s.hi[i];
    } else {
        kosk::KHalf s;
        const bool hi = threadIdx.x & 1;
#pragma unroll
        for (int i = 0; i < 25; i++) s.w[i] = (threadIdx.x * 2654435761u + i) ^ (blockIdx.x * 40503u + i * 7);
#pragma unroll 1
        for (int p = 0; p < nperm; p++) kosk::keccak_f1600_split(s, hi);
#pragma unroll
        for (int i = 0; i < 25; i++) acc ^= s.w[i];
    }
    out[blockIdx.x * 64 + threadIdx.x] = acc;
}

// both layouts on the same 32 states per wave: out[state][50] one-lane form, out2[state][50] pair form
__global__ __launch_bounds__(64) void k_check(uint32_t *o1, uint32_t *o2, int nperm)
{
    const int state = blockIdx.x * 32 + (threadIdx.x >> 1);
    kosk::KState a;
    kosk::KHalf h;
    const bool hi = threadIdx.x & 1;
#pragma unroll
    for (int i = 0; i < 25; i++) {
        a.lo[i] = state * 2654435761u + i * 97 + 1;
        a.hi[i] = state * 40503u + i * 7919 + 5;
        h.w[i] = hi ? a.hi[i] : a.lo[i];
    }
    for (int p = 0; p < nperm; p++) { kosk::keccak_f1600_dev(a); kosk::keccak_f1600_split(h, hi); }
    for (int i = 0; i < 25; i++) {
        if (!hi) { o1[(size_t)state * 50 + 2 * i] = a.lo[i]; o1[(size_t)state * 50 + 2 * i + 1] = a.hi[i]; }
        o2[(size_t)state * 50 + 2 * i + (hi ? 1 : 0)] = h.w[i];
    }
}

static double median(std::vector<double> v)
{
    std::sort(v.begin(), v.end());
    return v[v.size() / 2];
}

template <class L>
static void timed(L &&launch, int waves, Stamp *d_st, double &ms, double &cyc, double &ghz)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    launch();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float f;
    hipEventElapsedTime(&f, e0, e1);
    ms = f;
    std::vector<Stamp> h(waves);
    hipMemcpy(h.data(), d_st, sizeof(Stamp) * waves, hipMemcpyDeviceToHost);
    std::vector<double> c, g;
    for (auto &s : h) {
        c.push_back((double)s.cyc);
        g.push_back(s.rt ? (double)s.cyc / ((double)s.rt * 10.0) : 0.0); // cycles per ns = GHz (s_memrealtime ticks at 100 MHz)
    }
    cyc = median(c);
    ghz = median(g);
    hipEventDestroy(e0);
    hipEventDestroy(e1);
}

int main()
{
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int simds = p.multiProcessorCount * 4;
    printf("# %s, %d CUs, %d SIMDs\n", p.gcnArchName, p.multiProcessorCount, simds);
    int *out;
    Stamp *st;
    hipMalloc(&out, (size_t)simds * 8 * 64 * 4);
    hipMalloc(&st, (size_t)simds * 8 * sizeof(Stamp));

    // warm the clocks
    for (int i = 0; i < 200; i++) hipLaunchKernelGGL(k_keccak, dim3(simds * 4), dim3(64), 0, 0, (uint32_t *)out, st, 64);
    hipDeviceSynchronize();

    printf("# (1) single opcodes: 8 independent chains per lane, 32 instructions per trip, 4000 trips; wave64\n");
    printf("# cycles/instr = in-kernel s_memtime delta / instructions of ONE wave (median over waves); with w waves per SIMD the SIMD issues\n");
    printf("# one instruction every (cycles/instr)/w cycles\n");
    printf("%-26s %5s %14s %18s %10s %16s\n", "opcode", "w/SIMD", "cyc/instr/wave", "SIMD cyc per instr", "clock GHz", "T lane-op/s wall");
    for (int w : {1, 2, 4, 8}) {
        const int blocks = simds * w, iters = 4000;
        double ms[NOPS], cyc[NOPS], ghz[NOPS];
#define X(id, s) timed([&] { hipLaunchKernelGGL(k_op<id>, dim3(blocks), dim3(64), 0, 0, out, st, iters); }, blocks, st, ms[id], cyc[id], ghz[id]);
        OPS(X)
#undef X
        for (int op = 0; op < NOPS; op++) {
            const double per = cyc[op] / (iters * 32.0);
            printf("%-26s %5d %14.2f %18.2f %10.3f %16.1f\n", kNames[op], w, per, per / w, ghz[op], (double)blocks * 64 * iters * 32 / (ms[op] * 1e-3) / 1e12);
        }
    }

    printf("# (2) keccak_f1600_dev in registers (csrc/kosk_keccak_dev.hpp: 180 VALU per round = 120 v_bitop3 + 58 v_alignbit + 2 v_xor), 256 permutations per lane\n");
    printf("%6s %8s %12s %16s %18s %10s %14s\n", "w/SIMD", "waves", "wall us", "G Keccak-f/s", "cyc/instr/wave", "clock GHz", "SIMD cyc/instr");
    for (int w : {1, 2, 3, 4, 6, 8}) {
        const int blocks = simds * w, nperm = 256;
        double ms, cyc, ghz;
        timed([&] { hipLaunchKernelGGL(k_keccak, dim3(blocks), dim3(64), 0, 0, (uint32_t *)out, st, nperm); }, blocks, st, ms, cyc, ghz);
        const double per = cyc / (nperm * 24.0 * 180.0);
        printf("%6d %8d %12.1f %16.2f %18.2f %10.3f %14.2f\n", w, blocks, ms * 1e3, (double)blocks * 64 * nperm / (ms * 1e-3) / 1e9, per, ghz, per / w);
    }
    // the quantisation step of the 46-proof batch: 1046 waves on 1024 SIMDs
    for (int blocks : {simds - 64, simds, simds + 22, simds + 256, simds * 2}) {
        double ms, cyc, ghz;
        timed([&] { hipLaunchKernelGGL(k_keccak, dim3(blocks), dim3(64), 0, 0, (uint32_t *)out, st, 4); }, blocks, st, ms, cyc, ghz);
        printf("# 4 permutations per lane, %5d waves: %.1f us\n", blocks, ms * 1e3);
    }
    { // lane-pair layout: bit-exact against the one-lane form, then the same sweep
        const int nb = 64;
        uint32_t *o1, *o2;
        hipMalloc(&o1, (size_t)nb * 32 * 50 * 4);
        hipMalloc(&o2, (size_t)nb * 32 * 50 * 4);
        hipLaunchKernelGGL(k_check, dim3(nb), dim3(64), 0, 0, o1, o2, 3);
        std::vector<uint32_t> h1((size_t)nb * 32 * 50), h2(h1.size());
        hipMemcpy(h1.data(), o1, h1.size() * 4, hipMemcpyDeviceToHost);
        hipMemcpy(h2.data(), o2, h2.size() * 4, hipMemcpyDeviceToHost);
        size_t bad = 0;
        for (size_t i = 0; i < h1.size(); i++) bad += h1[i] != h2[i];
        printf("# (3) lane-pair layout (one state on two lanes, 120 instructions per lane and round): %zu of %zu words differ from the one-lane form after 3 permutations\n", bad, h1.size());
        printf("%6s %8s %12s %16s %18s %10s\n", "w/SIMD", "waves", "wall us", "G Keccak-f/s", "cyc/instr/wave", "clock GHz");
        for (int w : {2, 4, 6, 8}) {
            const int blocks = simds * w, nperm = 256;
            double ms, cyc, ghz;
            timed([&] { hipLaunchKernelGGL(k_keccak_split, dim3(blocks), dim3(64), 0, 0, (uint32_t *)out, st, nperm); }, blocks, st, ms, cyc, ghz);
            printf("%6d %8d %12.1f %16.2f %18.2f %10.3f\n", w, blocks, ms * 1e3, (double)blocks * 32 * nperm / (ms * 1e-3) / 1e9, cyc / (nperm * 24.0 * 120.0), ghz);
        }
        for (int states : {65536, 66884, 131072}) {
            const int blocks = states / 32 + (states % 32 != 0);
            double ms, cyc, ghz;
            timed([&] { hipLaunchKernelGGL(k_keccak_split, dim3(blocks), dim3(64), 0, 0, (uint32_t *)out, st, 4); }, blocks, st, ms, cyc, ghz);
            printf("# 4 permutations per state, %6d states as %5d pair-waves: %.1f us\n", states, blocks, ms * 1e3);
        }
    }
    printf("# (4) hybrid launch, 4 permutations per state: <main> one-lane waves of 64 states first, then <tail> lane-pair waves of 32 states\n");
    for (auto mt : {std::pair<int, int>{1024, 0}, {1024, 43}, {1012, 92}, {1012, 0}, {960, 171}, {1024, 128}, {1024, 512}, {512, 1024}, {0, 1024}}) {
        const int blocks = mt.first + mt.second;
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        float best = 1e9f;
        for (int rep = 0; rep < 5; rep++) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k_keccak_hybrid, dim3(blocks), dim3(64), 0, 0, (uint32_t *)out, mt.first, 4);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float f;
            hipEventElapsedTime(&f, e0, e1);
            best = std::min(best, f);
        }
        printf("# main %5d + tail %5d waves = %6d states: %.1f us\n", mt.first, mt.second, mt.first * 64 + mt.second * 32, best * 1e3);
    }
    return 0;
}
