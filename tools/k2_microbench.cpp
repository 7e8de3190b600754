// k2_microbench -- native driver of the fused ProcessFrame(+hist) kernel on a contiguous synthetic slab
// (BASELINE.json configs[2]: "Synthetic 10k-frame stack 1280x1024, fused diff+thresh+morph HBM-roofline
// microbench").  Exists so that rocprofv3 --pmc can wrap a plain native program.
//   k2_microbench [frames=2000] [reps=5] [store=0|1] [W=1280] [H=1024] [rows_per_chunk=0] [sigma=1] [chain=1]
//                 [cycle=0]
// (ABUB_K2_BOUND=0 in the environment: the plain row machine for every row, the dense-regime worst case)
// Build: hipcc --offload-arch=gfx950 -O2 tools/k2_microbench.cpp -Iinclude -Lautobub3hs_amd -labub_hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#include "abub_hip.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
#define AK(x) do { int r = (x); if (r) { fprintf(stderr, "%s: rc=%d %s\n", #x, r, abub_last_error()); return 1; } } while (0)

__device__ inline uint32_t mix(uint32_t x)
{
    x = (x ^ (x >> 16)) * 0x45d9f3bu;
    x = (x ^ (x >> 16)) * 0x45d9f3bu;
    return x ^ (x >> 16);
}
// background + sum of four U{-1,0,1} per frame, sparse bright discs drifting every 64 frames
__global__ void fill(uint8_t *slab, size_t P, int W, size_t total, int discs)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < total; i += stride) {
        size_t f = i / P, p = i % P;
        int x = (int)(p % W), y = (int)(p / W);
        uint32_t h = mix((uint32_t)p * 2654435761u ^ mix((uint32_t)f + 77u));
        int n = (int)((h & 0xff) % 3 + ((h >> 8) & 0xff) % 3 + ((h >> 16) & 0xff) % 3 + ((h >> 24) & 0xff) % 3) - 4;
        int v = 40 + (x * 60) / W + (y % 97) / 4 + n;
        int k = (int)(f % 64);
        int dx = x - (200 + (int)((f / 64) * 37 % 800)), dy = y - (300 + (int)((f / 64) * 53 % 400));
        if (discs && k >= 32 && dx * dx + dy * dy <= (2 + k - 32) * (2 + k - 32))
            v += 40;
        slab[i] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
    }
}

int main(int argc, char **argv)
{
    int F = argc > 1 ? atoi(argv[1]) : 2000, reps = argc > 2 ? atoi(argv[2]) : 5;
    int store = argc > 3 ? atoi(argv[3]) : 0, W = argc > 4 ? atoi(argv[4]) : 1280, H = argc > 5 ? atoi(argv[5]) : 1024;
    int R = argc > 6 ? atoi(argv[6]) : 0;
    int chain = argc > 8 ? atoi(argv[8]) : 1; // 1: pass the chain hint (jobs f, f+2 share a frame), 0: plain job list
    int sig = argc > 7 ? atoi(argv[7]) : 1; // model sigma: 1 -> ~3.5 supra-threshold noise pixels per row, 2 -> none
    int co = argc > 12 ? atoi(argv[12]) : 0; // K3 only: 1 = with the fused candidate list (cut 3 for every frame)
    int k3 = argc > 11 ? atoi(argv[11]) : 0; // 1: time K3 (post-trigger image + histogram, no store) over the same frames
    int discs = argc > 10 ? atoi(argv[10]) : 1; // 0: noise only (no growing discs)
    int cyc = argc > 9 ? atoi(argv[9]) : 0; // > 0: the jobs cycle through the first `cyc` frames only (L2-resident inputs:
                                            // what the kernel costs when memory is free)
    size_t P = (size_t)W * H;
    uint8_t *slab, *sigma, *sigma6, *diff = nullptr;
    uint32_t *hist;
    abub_job *jobs;
    int njobs = F - 2;
    CK(hipMalloc(&slab, P * F));
    CK(hipMalloc(&sigma, P));
    CK(hipMalloc(&sigma6, P));
    CK(hipMalloc(&hist, (size_t)njobs * 1024));
    CK(hipMalloc(&jobs, (size_t)njobs * sizeof(abub_job)));
    if (store)
        CK(hipMalloc(&diff, P * (size_t)njobs));
    hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, slab, P, W, P * F, discs);
    CK(hipMemset(sigma, sig, P));
    AK(abub_sigma6_dev(sigma, sigma6, P, nullptr));
    std::vector<abub_job> hj(njobs);
    for (int j = 0; j < njobs; j++)
        hj[j] = cyc > 0 ? abub_job{(uint32_t)((j + 2) % cyc), (uint32_t)(j % cyc), 0u, (uint32_t)j}
                        : abub_job{(uint32_t)(j + 2), (uint32_t)j, 0u, (uint32_t)j};
    CK(hipMemcpy(jobs, hj.data(), njobs * sizeof(abub_job), hipMemcpyHostToDevice));
    CK(hipDeviceSynchronize());
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    if (chain && !R && store) AK(abub_diff_hist_chained_store_dev(slab, sigma6, jobs, njobs, W, H, hist, diff, njobs, 2, nullptr)); else if (chain && !R) AK(abub_diff_hist_chained_dev(slab, sigma6, jobs, njobs, W, H, hist, njobs, 2, nullptr)); else AK(abub_diff_hist_dev(slab, sigma6, jobs, njobs, W, H, hist, diff, R, nullptr)); // warm-up
    CK(hipDeviceSynchronize());
    float best = 1e30f, sum = 0;
    for (int r = 0; r < reps; r++) {
        CK(hipEventRecord(a, 0));
        if (chain && !R && store) AK(abub_diff_hist_chained_store_dev(slab, sigma6, jobs, njobs, W, H, hist, diff, njobs, 2, nullptr)); else if (chain && !R) AK(abub_diff_hist_chained_dev(slab, sigma6, jobs, njobs, W, H, hist, njobs, 2, nullptr)); else AK(abub_diff_hist_dev(slab, sigma6, jobs, njobs, W, H, hist, diff, R, nullptr));
        CK(hipEventRecord(b, 0));
        CK(hipEventSynchronize(b));
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        sum += ms;
        if (ms < best)
            best = ms;
    }
    if (k3) { // K3: mu = frame 0 of the slab (so |f - mu| is noise plus the discs), the same sigma6
        uint8_t *mu;
        CK(hipMalloc(&mu, P));
        CK(hipMemcpy(mu, slab, P, hipMemcpyDeviceToDevice));
        int32_t *cthr = nullptr;
        uint32_t *pairs = nullptr, *pcount = nullptr;
        const uint32_t pcap = 32u << 20;
        if (co) {
            std::vector<int32_t> hc(njobs, 3);
            CK(hipMalloc(&cthr, njobs * sizeof(int32_t)));
            CK(hipMemcpy(cthr, hc.data(), njobs * sizeof(int32_t), hipMemcpyHostToDevice));
            CK(hipMalloc(&pairs, (size_t)pcap * 8));
            CK(hipMalloc(&pcount, 256));
        }
#define K3_CALL()                                                                                                   \
    if (co) {                                                                                                       \
        CK(hipMemsetAsync(pcount, 0, 4, 0));                                                                        \
        AK(abub_posttrig_compact_dev(slab, mu, sigma6, jobs, njobs, W, H, hist, nullptr, cthr, pairs, pcap, pcount, 0, nullptr)); \
    } else                                                                                                          \
        AK(abub_posttrig_dev(slab, mu, sigma6, jobs, njobs, W, H, hist, nullptr, nullptr))
        K3_CALL();
        CK(hipDeviceSynchronize());
        best = 1e30f;
        sum = 0;
        for (int r = 0; r < reps; r++) {
            CK(hipEventRecord(a, 0));
            K3_CALL();
            CK(hipEventRecord(b, 0));
            CK(hipEventSynchronize(b));
            float ms;
            CK(hipEventElapsedTime(&ms, a, b));
            sum += ms;
            if (ms < best)
                best = ms;
        }
        uint32_t npairs = 0;
        if (co)
            CK(hipMemcpy(&npairs, pcount, 4, hipMemcpyDeviceToHost));
        printf("{\"k3\": 1, \"compact\": %d, \"pairs\": %u, \"frames\": %d, \"W\": %d, \"H\": %d, \"ms_avg\": %.4f, \"ms_min\": %.4f, \"us_per_frame\": %.4f, \"compulsory_GBps\": %.1f}\n",
               co, npairs, njobs, W, H, sum / reps, best, 1e3 * sum / reps / njobs, (double)P * njobs / (sum / reps * 1e-3) / 1e9);
        return 0;
    }
    std::vector<uint32_t> hh((size_t)njobs * 256);
    CK(hipMemcpy(hh.data(), hist, hh.size() * 4, hipMemcpyDeviceToHost));
    unsigned long long nz = 0, tot = 0;
    for (int j = 0; j < njobs; j++)
        for (int k = 0; k < 256; k++) {
            tot += hh[(size_t)j * 256 + k];
            if (k)
                nz += hh[(size_t)j * 256 + k];
        }
    double ms = sum / reps;
    // compulsory bytes: every frame once (+ D once in store mode); the contract's algorithmic figure charges cur, ref, sigma6
    double bytes = (store ? 4.0 : 3.0) * P * njobs, comp = (store ? 2.0 : 1.0) * P * njobs;
    printf("{\"frames\": %d, \"W\": %d, \"H\": %d, \"store\": %d, \"chain\": %d, \"sigma\": %d, \"cycle\": %d, \"ms_avg\": %.4f, \"ms_min\": %.4f, "
           "\"frames_per_s\": %.1f, \"compulsory_GBps\": %.1f, \"frac_of_8TBps\": %.4f, \"alg_GBps\": %.1f, \"nonzero_px\": %llu, "
           "\"hist_total_ok\": %d}\n",
           njobs, W, H, store, chain, sig, cyc, ms, best, njobs / (ms * 1e-3), comp / (ms * 1e-3) / 1e9, comp / (ms * 1e-3) / 1e9 / 8000.0,
           bytes / (ms * 1e-3) / 1e9, nz, tot == (unsigned long long)P * njobs);
    return 0;
}
