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
        printf("{\"rows_per_iter\": %d, \"prefetch\": %d, \"jobs\": %d, \"ms\": %.4f, \"alg_GBps\": %.1f}\n", RPI, PF, njobs, ms, \
               3.0 * P * njobs / ms / 1e6);                                                                      \
    }
    RUN(1, 1) RUN(1, 2) RUN(1, 3) RUN(2, 1) RUN(2, 2) RUN(4, 1)
    return 0;
}
