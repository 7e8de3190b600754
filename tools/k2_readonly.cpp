// Upper bound for K2's access pattern: the same (job, chunk) -> wave mapping and the same three row streams
// (cur, ref = cur-2, sigma6), but the rows are only OR-ed together.  Tells how fast the memory system can feed
// the K2 mapping when arithmetic is free.  RPI = rows fetched per loop iteration (1 = K2's mapping).
// Build: hipcc --offload-arch=gfx950 -O3 tools/k2_readonly.cpp -o tools/k2_readonly
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int NDW, int RPI, int PF>
__global__ __launch_bounds__(64) void rd(const uint8_t *__restrict__ frames, const uint8_t *__restrict__ sg, int W, int H,
                                         int R, int nchunks, uint32_t *out)
{
    const int lane = threadIdx.x, unit = blockIdx.x, job = unit / nchunks, chunk = unit - job * nchunks;
    const size_t P = (size_t)W * H;
    const uint8_t *cur = frames + (size_t)(job + 2) * P, *ref = frames + (size_t)job * P;
    const int xoff = lane * 4 * NDW;
    const int y0 = chunk * R;
    int y1 = y0 + R; if (y1 > H) y1 = H;
    uint32_t acc = 0;
    uint32_t buf[PF + 1][RPI][3][NDW];
    auto load = [&](int slot, int y) {
#pragma unroll
        for (int r = 0; r < RPI; r++) {
            int yy = y + r < H ? y + r : H - 1;
            size_t o = (size_t)yy * W + xoff;
            const uint32_t *a = (const uint32_t *)(cur + o), *b = (const uint32_t *)(ref + o), *c = (const uint32_t *)(sg + o);
#pragma unroll
            for (int d = 0; d < NDW; d++) { buf[slot][r][0][d] = a[d]; buf[slot][r][1][d] = b[d]; buf[slot][r][2][d] = c[d]; }
        }
    };
#pragma unroll
    for (int k = 0; k < PF; k++) load(k, y0 + k * RPI);
    constexpr int U = PF + 1;
    for (int y = y0; y < y1; y += RPI * U) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            load((u + PF) % U, y + (u + PF) * RPI);
#pragma unroll
            for (int r = 0; r < RPI; r++)
#pragma unroll
                for (int d = 0; d < NDW; d++) acc |= buf[u][r][0][d] ^ buf[u][r][1][d] ^ buf[u][r][2][d];
        }
    }
    if (acc == 0x12345678u) out[unit] = acc;
}


// chain pattern: one wave serves K jobs of the same parity (frames f, f+2, .., f+2K: each frame row is loaded once and
// used as cur of one job and ref of the next) -> (K+2)/K row loads per job instead of 3
template <int NDW, int K>
__global__ __launch_bounds__(64) void rdchain(const uint8_t *__restrict__ frames, const uint8_t *__restrict__ sg, int W, int H,
                                              int R, int nchunks, uint32_t *out)
{
    const int lane = threadIdx.x, unit = blockIdx.x, ch = unit / nchunks, chunk = unit - ch * nchunks;
    const size_t P = (size_t)W * H;
    const int base = (ch / 2) * 2 * K + (ch & 1); // first job of the chain
    const int xoff = lane * 4 * NDW;
    const int y0 = chunk * R;
    int y1 = y0 + R; if (y1 > H) y1 = H;
    uint32_t acc = 0;
    uint32_t buf[2][K + 2][NDW];
    auto load = [&](int slot, int y) {
        int yy = y < H ? y : H - 1;
        size_t o = (size_t)yy * W + xoff;
#pragma unroll
        for (int f = 0; f <= K; f++) {
            const uint32_t *a = (const uint32_t *)(frames + (size_t)(base + 2 * f) * P + o);
#pragma unroll
            for (int d = 0; d < NDW; d++) buf[slot][f][d] = a[d];
        }
        const uint32_t *c = (const uint32_t *)(sg + o);
#pragma unroll
        for (int d = 0; d < NDW; d++) buf[slot][K + 1][d] = c[d];
    };
    load(0, y0);
    for (int y = y0; y < y1; y += 2) {
#pragma unroll
        for (int u = 0; u < 2; u++) {
            load(u ^ 1, y + u + 1);
#pragma unroll
            for (int f = 0; f < K + 2; f++)
#pragma unroll
                for (int d = 0; d < NDW; d++) acc |= buf[u][f][d] + (uint32_t)f;
        }
    }
    if (acc == 0x12345678u) out[unit] = acc;
}

// co-scheduled pattern: a workgroup of NW waves, wave w serves job base + 2w of the same chunk (K2's mapping otherwise):
// the frame shared by jobs j and j+2 is requested by two waves of the same CU at about the same time
template <int NDW, int NW>
__global__ __launch_bounds__(64 * NW) void rdco(const uint8_t *__restrict__ frames, const uint8_t *__restrict__ sg, int W, int H,
                                                int R, int nchunks, int njobs, uint32_t *out)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, unit = blockIdx.x, grp = unit / nchunks, chunk = unit - grp * nchunks;
    const int job = (grp / 2) * 2 * NW + (grp & 1) + 2 * wv;
    if (job >= njobs) return;
    const size_t P = (size_t)W * H;
    const uint8_t *cur = frames + (size_t)(job + 2) * P, *ref = frames + (size_t)job * P;
    const int xoff = lane * 4 * NDW;
    const int y0 = chunk * R;
    int y1 = y0 + R; if (y1 > H) y1 = H;
    uint32_t acc = 0;
    uint32_t buf[2][3][NDW];
    auto load = [&](int slot, int y) {
        int yy = y < H ? y : H - 1;
        size_t o = (size_t)yy * W + xoff;
        const uint32_t *a = (const uint32_t *)(cur + o), *b = (const uint32_t *)(ref + o), *c = (const uint32_t *)(sg + o);
#pragma unroll
        for (int d = 0; d < NDW; d++) { buf[slot][0][d] = a[d]; buf[slot][1][d] = b[d]; buf[slot][2][d] = c[d]; }
    };
    load(0, y0);
    for (int y = y0; y < y1; y += 2) {
#pragma unroll
        for (int u = 0; u < 2; u++) {
            load(u ^ 1, y + u + 1);
#pragma unroll
            for (int d = 0; d < NDW; d++) acc |= buf[u][0][d] ^ buf[u][1][d] ^ buf[u][2][d];
        }
    }
    if (acc == 0x12345678u) out[unit] = acc;
}

int main(int argc, char **argv)
{
    int F = argc > 1 ? atoi(argv[1]) : 2000, W = 1280, H = 1024, R = 128, reps = 5;
    size_t P = (size_t)W * H;
    uint8_t *slab, *sg; uint32_t *out;
    CK(hipMalloc(&slab, P * F)); CK(hipMalloc(&sg, P)); CK(hipMalloc(&out, 4 * (size_t)F * 8));
    CK(hipMemset(slab, 1, P * F)); CK(hipMemset(sg, 2, P));
    int njobs = F - 2, nch = (H + R - 1) / R;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
#define RUN(RPI, PF)                                                                                            \
    {                                                                                                            \
        hipLaunchKernelGGL((rd<5, RPI, PF>), dim3(njobs * nch), dim3(64), 0, 0, slab, sg, W, H, R, nch, out);    \
        CK(hipDeviceSynchronize());                                                                              \
        CK(hipEventRecord(e0));                                                                                  \
        for (int i = 0; i < reps; i++)                                                                           \
            hipLaunchKernelGGL((rd<5, RPI, PF>), dim3(njobs * nch), dim3(64), 0, 0, slab, sg, W, H, R, nch, out);\
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));                                                     \
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;                                              \
        printf("{\"rows_per_iter\": %d, \"prefetch\": %d, \"jobs\": %d, \"ms\": %.4f, \"alg_GBps\": %.1f, \"us_per_job\": %.4f}\n", RPI, PF, njobs, ms, \
               3.0 * P * njobs / ms / 1e6, 1e3 * ms / njobs);                                                                      \
    }
    RUN(1, 1) RUN(1, 2) RUN(1, 3) RUN(2, 1) RUN(2, 2) RUN(4, 1)
#define RUNC(K)                                                                                                  \
    {                                                                                                            \
        int nchains = (njobs / (2 * K)) * 2;                                                                     \
        hipLaunchKernelGGL((rdchain<5, K>), dim3(nchains * nch), dim3(64), 0, 0, slab, sg, W, H, R, nch, out);   \
        CK(hipDeviceSynchronize());                                                                              \
        CK(hipEventRecord(e0));                                                                                  \
        for (int i = 0; i < reps; i++)                                                                           \
            hipLaunchKernelGGL((rdchain<5, K>), dim3(nchains * nch), dim3(64), 0, 0, slab, sg, W, H, R, nch, out);\
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));                                                     \
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;                                              \
        printf("{\"chain\": %d, \"jobs\": %d, \"ms\": %.4f, \"us_per_job\": %.4f}\n", K, nchains * K, ms, 1e3 * ms / (nchains * K)); \
    }
    RUNC(1) RUNC(2) RUNC(4) RUNC(8)
#define RUNW(NW)                                                                                                 \
    {                                                                                                            \
        int ngrp = ((njobs + 2 * NW - 1) / (2 * NW)) * 2;                                                        \
        hipLaunchKernelGGL((rdco<5, NW>), dim3(ngrp * nch), dim3(64 * NW), 0, 0, slab, sg, W, H, R, nch, njobs, out); \
        CK(hipDeviceSynchronize());                                                                              \
        CK(hipEventRecord(e0));                                                                                  \
        for (int i = 0; i < reps; i++)                                                                           \
            hipLaunchKernelGGL((rdco<5, NW>), dim3(ngrp * nch), dim3(64 * NW), 0, 0, slab, sg, W, H, R, nch, njobs, out); \
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));                                                     \
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;                                              \
        printf("{\"waves_per_group\": %d, \"jobs\": %d, \"ms\": %.4f, \"us_per_job\": %.4f}\n", NW, njobs, ms, 1e3 * ms / njobs); \
    }
    RUNW(1) RUNW(2) RUNW(4) RUNW(8)
    return 0;
}
