// Probe (not product code): what bounds the wire-image pair (k_assemble_fields / k_disassemble_fields)?  The access patterns of those
// kernels on a row matrix of the real shape -- 138 proofs x 435 rows x 1728 u16, a wave taking a 64-column window of ~72 rows:
// one 128-byte line per row, 3 456 bytes apart -- against a linear sweep of the same bytes, with and without the transposition
// through LDS and the sequential write of the image.   hipcc --offload-arch=gfx950 -O3 tools/probe_wire.hip -o tools/probe_wire
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int NP = 138, ROWS = 435, RS = 1728, NWIN = 23, GROUPS = 6, GR = 73; // 6 groups of <= 73 rows cover the 435 rows
constexpr size_t PSTRIDE = (size_t)ROWS * RS;

__device__ __forceinline__ void decode(int id, int &b, int &g, int &win) { win = id % NWIN; g = (id / NWIN) % GROUPS; b = id / (NWIN * GROUPS); }

// V0: the gather of k_assemble_fields: every lane one u16 of each row of its group (all loads in flight), nothing else
__global__ __launch_bounds__(64) void k_gather16(const uint16_t *P, uint32_t *sink)
{
    int b, g, win; decode(blockIdx.x, b, g, win);
    const int r0 = g * GR, n = min(GR, ROWS - r0);
    const uint16_t *src = P + (size_t)b * PSTRIDE + (size_t)r0 * RS + win * 64 + threadIdx.x;
    uint32_t acc = 0;
#pragma unroll 73
    for (int r = 0; r < GR; r++) acc ^= src[(size_t)(r < n ? r : n - 1) * RS];
    if (acc == 0xDEADBEEFu) sink[blockIdx.x] = acc;
}
// V1: the same bytes with 8-byte loads: 16 lanes per row line, four rows per instruction
__global__ __launch_bounds__(64) void k_gather64(const uint16_t *P, uint32_t *sink)
{
    int b, g, win; decode(blockIdx.x, b, g, win);
    const int r0 = g * GR, n = min(GR, ROWS - r0);
    const uint16_t *src = P + (size_t)b * PSTRIDE + (size_t)r0 * RS + win * 64 + (threadIdx.x & 15) * 4;
    uint32_t acc = 0;
#pragma unroll 19
    for (int q = 0; q < 19; q++) {
        int r = q * 4 + (threadIdx.x >> 4);
        r = r < n ? r : n - 1;
        const uint2 v = *reinterpret_cast<const uint2 *>(src + (size_t)r * RS);
        acc ^= v.x ^ v.y;
    }
    if (acc == 0xDEADBEEFu) sink[blockIdx.x] = acc;
}
// V2: two windows per wave (256 contiguous bytes per row and instruction), half the rows per wave
__global__ __launch_bounds__(64) void k_gather_2win(const uint16_t *P, uint32_t *sink)
{
    const int id = blockIdx.x; // (proof, 12 half groups, 12 window pairs (the 23rd window alone))
    const int wp = id % 12, hg = (id / 12) % 12, b = id / 144;
    const int r0 = hg * 37, n = min(37, ROWS - r0);
    const int col = wp * 128 + threadIdx.x * 2;
    const uint16_t *src = P + (size_t)b * PSTRIDE + (size_t)r0 * RS + (col < 1472 ? col : 0);
    uint32_t acc = 0;
#pragma unroll 37
    for (int r = 0; r < 37; r++) acc ^= *reinterpret_cast<const uint32_t *>(src + (size_t)(r < n ? r : n - 1) * RS);
    if (acc == 0xDEADBEEFu) sink[blockIdx.x] = acc;
}
// V3: a linear sweep of the same number of bytes (16 bytes per lane, 8 loads in flight per lane)
__global__ __launch_bounds__(256) void k_stream(const uint4 *src, uint32_t *sink, size_t n16)
{
    uint32_t acc = 0;
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 7 * stride < n16; i += 8 * stride) {
        uint4 v[8];
#pragma unroll
        for (int q = 0; q < 8; q++) v[q] = src[i + q * stride];
#pragma unroll
        for (int q = 0; q < 8; q++) acc ^= v[q].x ^ v[q].y ^ v[q].z ^ v[q].w;
    }
    if (acc == 0xDEADBEEFu) sink[blockIdx.x] = acc;
}
// V4: gather + transposition through LDS + sequential write of the party-major records (what k_assemble_fields does for the
// fields of unopened parties); V5 (PIPE): a wave takes TWO windows and has the second one's gather in flight while it writes
// the first one out (two LDS tiles)
template <bool PIPE>
__global__ __launch_bounds__(64) void k_transpose(const uint16_t *P, uint16_t *img)
{
    __shared__ uint16_t tile[PIPE ? 2 : 1][64 * GR + 8];
    const int lane = threadIdx.x;
    const int per = PIPE ? 2 : 1;
    uint32_t v[PIPE ? 2 : 1][GR];
    int bb[2], gg[2], ww[2], nn[2];
#pragma unroll
    for (int t = 0; t < per; t++) {
        decode(blockIdx.x * per + t, bb[t], gg[t], ww[t]);
        nn[t] = min(GR, ROWS - gg[t] * GR);
    }
    auto gather = [&](int t) {
        const uint16_t *src = P + (size_t)bb[t] * PSTRIDE + (size_t)(gg[t] * GR) * RS + ww[t] * 64 + lane;
#pragma unroll
        for (int r = 0; r < GR; r++) v[t][r] = src[(size_t)(r < nn[t] ? r : nn[t] - 1) * RS];
    };
    auto put = [&](int t) {
        uint16_t *tl = tile[PIPE ? t : 0] + lane * nn[t];
#pragma unroll
        for (int r = 0; r < GR; r++)
            if (r < nn[t]) tl[r] = (uint16_t)v[t][r];
    };
    auto flush = [&](int t) {
        // image: [proof][group][party][rows of the group]: the window's 64 parties are one contiguous run
        uint32_t *out = reinterpret_cast<uint32_t *>(img + ((size_t)bb[t] * ROWS * 1472) + (size_t)gg[t] * GR * 1472 + (size_t)ww[t] * 64 * nn[t]);
        const uint32_t *tw = reinterpret_cast<const uint32_t *>(tile[PIPE ? t : 0]);
        for (int q = lane; q < 32 * nn[t]; q += 64) out[q] = tw[q];
    };
    gather(0);
    if (PIPE) gather(1);
    put(0);
    __builtin_amdgcn_s_waitcnt(0xC07F); // lgkmcnt(0)
    __builtin_amdgcn_wave_barrier();
    flush(0);
    if (PIPE) {
        put(1);
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_wave_barrier();
        flush(1);
    }
}
// V6: the sequential write alone
__global__ __launch_bounds__(256) void k_write(uint4 *dst, size_t n16)
{
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride) dst[i] = make_uint4((uint32_t)i, 1, 2, 3);
}

template <class F>
static void timeit(const char *name, double bytes, F &&launch)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    launch(); launch(); hipDeviceSynchronize();
    float best = 1e9f;
    for (int rep = 0; rep < 5; rep++) {
        hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    printf("%-44s %7.1f us  %6.2f TB/s (%.1f MB)\n", name, best * 1e3, bytes / (best * 1e-3) / 1e12, bytes / 1e6);
}

int main()
{
    uint16_t *P, *img; uint32_t *sink;
    const size_t pbytes = (size_t)NP * PSTRIDE * 2, ibytes = (size_t)NP * ROWS * 1472 * 2;
    hipMalloc(&P, pbytes); hipMalloc(&img, ibytes); hipMalloc(&sink, 1 << 20);
    hipMemset(P, 0x11, pbytes);
    // a second matrix twice the size of the Infinity Cache, swept between runs so that every variant starts from HBM
    uint4 *flush; const size_t fbytes = (size_t)640 << 20; hipMalloc(&flush, fbytes);
    auto cold = [&] { hipLaunchKernelGGL(k_write, dim3(2048), dim3(256), 0, 0, flush, fbytes / 16); };
    const int nwave = NP * GROUPS * NWIN;
    const double gbytes = (double)NP * ROWS * NWIN * 128;
    printf("row matrix %.1f MB, gathered bytes %.1f MB, image %.1f MB\n", pbytes / 1e6, gbytes / 1e6, ibytes / 1e6);
    for (int pass = 0; pass < 2; pass++) {
        printf(pass ? "-- from HBM (640 MB written between launches) --\n" : "-- back to back (Infinity Cache warm where it fits) --\n");
        auto wrap = [&](auto &&f) { return [&, f] { if (pass) cold(); f(); }; };
        // (with the flush in front the flush's own time is included: subtract the 'flush alone' line)
        if (pass) timeit("flush alone (640 MB write)", 0, [&] { cold(); });
        timeit("V0 gather, 2-byte loads, 73 rows x 128 B", gbytes, wrap([&] { hipLaunchKernelGGL(k_gather16, dim3(nwave), dim3(64), 0, 0, P, sink); }));
        timeit("V1 gather, 8-byte loads, 4 rows per instr", gbytes, wrap([&] { hipLaunchKernelGGL(k_gather64, dim3(nwave), dim3(64), 0, 0, P, sink); }));
        timeit("V2 gather, two windows (256 B per row)", gbytes, wrap([&] { hipLaunchKernelGGL(k_gather_2win, dim3(NP * 144), dim3(64), 0, 0, P, sink); }));
        timeit("V3 linear sweep of the same bytes", gbytes, wrap([&] { hipLaunchKernelGGL(k_stream, dim3(4096), dim3(256), 0, 0, reinterpret_cast<const uint4 *>(P), sink, (size_t)(gbytes / 16)); }));
        timeit("V4 gather + LDS transpose + image write", gbytes + ibytes, wrap([&] { hipLaunchKernelGGL(k_transpose<false>, dim3(nwave), dim3(64), 0, 0, P, img); }));
        timeit("V5 the same, two windows pipelined per wave", gbytes + ibytes, wrap([&] { hipLaunchKernelGGL(k_transpose<true>, dim3(nwave / 2), dim3(64), 0, 0, P, img); }));
        timeit("V6 sequential write of the image alone", ibytes, wrap([&] { hipLaunchKernelGGL(k_write, dim3(4096), dim3(256), 0, 0, reinterpret_cast<uint4 *>(img), ibytes / 16); }));
    }
    return 0;
}
