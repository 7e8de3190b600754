// abub_misc.hip -- the small kernels of the hot path and the library's host-side plumbing (gfx950):
//   K1  k1_welford, k1_welford4        Trainer::CalculateMeanSigmaImageVector (float32 Welford, no FMA)
//   K1b k1b_pair_hist                  histogram of sat(f1 - f0) (training entropy veto)
//   K4  k4_compact, k4_compact_pairs   binarize + foreground index compaction (unfused fallback)
//       k_pairs_*                      counting-sort grouping of the shared candidate list by image
//       k_match_ccorr, k_subsat_hist   bellows veto (TrackAFeature terms, image subtraction)
//   k_sigma6, k_fill_stack_jobs, error messages, device info, per-stream scratch of the bound-and-verify passes
#include "abub_dev.hpp"

// ------------------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------------------
static thread_local char g_err[256] = "";

extern "C" const char *abub_last_error(void) { return g_err; }

int abub_set_err_(int code, const char *what, hipError_t e)
{
    if (e != hipSuccess)
        snprintf(g_err, sizeof g_err, "%s: %s", what, hipGetErrorString(e));
    else
        snprintf(g_err, sizeof g_err, "%s", what);
    return code;
}

extern "C" int abub_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess)
        return 0;
    return n;
}

extern "C" int abub_device_info(int device, char *name, int name_cap, int *cus, uint64_t *hbm_bytes)
{
    hipDeviceProp_t p;
    HIPCHK(hipGetDeviceProperties(&p, device));
    if (name && name_cap > 0)
        snprintf(name, name_cap, "%s (%s)", p.name, p.gcnArchName);
    if (cus)
        *cus = p.multiProcessorCount;
    if (hbm_bytes)
        *hbm_bytes = p.totalGlobalMem;
    return ABUB_OK;
}

// ------------------------------------------------------------------------------------------------
// sigma6 = min(6*sigma, 255)
// ------------------------------------------------------------------------------------------------
__global__ void k_sigma6(const uint8_t *__restrict__ s, uint8_t *__restrict__ o, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        int v = 6 * (int)s[i];
        o[i] = (uint8_t)(v > 255 ? 255 : v);
    }
}

extern "C" int abub_sigma6_dev(const uint8_t *sigma, uint8_t *sigma6, size_t n, void *stream)
{
    if (!sigma || !sigma6)
        return set_err(ABUB_E_INVALID, "abub_sigma6_dev: null pointer");
    if (n == 0)
        return ABUB_OK;
    int blocks = (int)((n + 255) / 256);
    if (blocks > 4096)
        blocks = 4096;
    hipLaunchKernelGGL(k_sigma6, dim3(blocks), dim3(256), 0, (hipStream_t)stream, sigma, sigma6, n);
    HIPCHK(hipGetLastError());
    return ABUB_OK;
}

// ------------------------------------------------------------------------------------------------
// job list for regular stacks
// ------------------------------------------------------------------------------------------------
__global__ void k_fill_stack_jobs(abub_job *jobs, int nstacks, int F, int first, int count, int off,
                                  int nmodels)
{
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= nstacks * count)
        return;
    int s = j / count, n = j % count;
    int i = first + n;
    int r = i - off;
    if (r < 0)
        r = 0;
    abub_job jb;
    jb.cur = (uint32_t)(s * F + i);
    jb.ref = (uint32_t)(s * F + r);
    jb.model = (uint32_t)(s % nmodels);
    jb.out = (uint32_t)j;
    jobs[j] = jb;
}

extern "C" int abub_fill_stack_jobs_dev(abub_job *jobs, int nstacks, int F, int first, int count,
                                        int ref_offset, int nmodels, void *stream)
{
    if (!jobs || nstacks < 0 || F <= 0 || first < 0 || count < 0 || first + count > F ||
        ref_offset < 0 || nmodels <= 0)
        return set_err(ABUB_E_INVALID, "abub_fill_stack_jobs_dev: bad arguments");
    int n = nstacks * count;
    if (n == 0)
        return ABUB_OK;
    hipLaunchKernelGGL(k_fill_stack_jobs, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       jobs, nstacks, F, first, count, ref_offset, nmodels);
    HIPCHK(hipGetLastError());
    return ABUB_OK;
}

// Per-stream scratch of the bound-and-verify pass (grow-only; launches on one stream are ordered, launches on
// different streams get different buffers).  Growth frees the old buffer only after the stream has drained.
struct K2ScratchBuf {
    void *p = nullptr;
    size_t n = 0;
};
static std::mutex g_scratchMu;
static std::map<std::pair<int, hipStream_t>, K2ScratchBuf> g_scratch;

// Returns the stream's buffer with g_scratchMu HELD by `hold`: the caller keeps it until the whole launch sequence that
// uses the buffer is enqueued.  Several host threads may launch on one stream (the run pipeline's stack groups do):
// their sequences must not interleave, or one would reset the counters and overwrite the list of the other between
// its kernels.  (Enqueueing takes microseconds; execution is ordered by the stream.)
void *abub_k2_scratch(hipStream_t st, size_t bytes, std::unique_lock<std::mutex> &hold)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess)
        return nullptr;
    hold = std::unique_lock<std::mutex>(g_scratchMu);
    K2ScratchBuf &b = g_scratch[std::make_pair(dev, st)];
    if (b.n < bytes) {
        if (b.p) {
            (void)hipStreamSynchronize(st);
            (void)hipFree(b.p);
            b.p = nullptr;
            b.n = 0;
        }
        const size_t want = bytes + bytes / 4;
        if (hipMalloc(&b.p, want) != hipSuccess) {
            b.p = nullptr;
            return nullptr;
        }
        b.n = want;
    }
    return b.p;
}

extern "C" int abub_bound_counts_dev(void *stream, uint32_t counts[2])
{
    if (!counts)
        return set_err(ABUB_E_INVALID, "abub_bound_counts_dev: null pointer");
    counts[0] = counts[1] = 0;
    int dev = 0;
    HIPCHK(hipGetDevice(&dev));
    hipStream_t st = (hipStream_t)stream;
    std::lock_guard<std::mutex> lock(g_scratchMu);
    auto it = g_scratch.find(std::make_pair(dev, st));
    if (it == g_scratch.end() || !it->second.p)
        return ABUB_OK;
    uint32_t c[64];
    HIPCHK(hipStreamSynchronize(st));
    HIPCHK(hipMemcpy(c, it->second.p, sizeof c, hipMemcpyDeviceToHost));
    counts[0] = c[0];  // handed-over pieces
    counts[1] = c[32]; // entries of the global suspect list
    return ABUB_OK;
}

extern "C" int abub_scratch_release(void *stream)
{
    int dev = 0;
    HIPCHK(hipGetDevice(&dev));
    hipStream_t st = (hipStream_t)stream;
    std::lock_guard<std::mutex> lock(g_scratchMu);
    auto it = g_scratch.find(std::make_pair(dev, st));
    if (it == g_scratch.end())
        return ABUB_OK;
    (void)hipStreamSynchronize(st);
    if (it->second.p)
        (void)hipFree(it->second.p);
    g_scratch.erase(it);
    return ABUB_OK;
}

// ------------------------------------------------------------------------------------------------
// K1: float32 Welford, exactly the reference recurrence (Trainer.cpp:180-196); this TU is compiled
// with -ffp-contract=off, and hipcc's default correctly-rounded fp32 divide / sqrt stays on.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k1_welford(const uint8_t *__restrict__ frames,
                                                  const uint32_t *__restrict__ idx, int N, size_t P,
                                                  uint8_t *__restrict__ mu, uint8_t *__restrict__ sigma)
{
    size_t px = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (px >= P)
        return;
    float mean = 0.f, m2 = 0.f;
    for (int k = 0; k < N; k++) {
        size_t f = idx ? idx[k] : (uint32_t)k;
        float x = (float)frames[f * P + px];
        float delta = x - mean;
        mean = mean + delta / (float)(k + 1);
        m2 = m2 + delta * (x - mean);
    }
    float var = m2 / (float)(N - 1);
    float sd = sqrtf(var);
    int isd = (sd != sd) ? 0 : (int)sd;
    sigma[px] = (uint8_t)isd;
    mu[px] = (uint8_t)(int)mean;
}

// 4 pixels per thread (dword loads) when P % 4 == 0
__global__ __launch_bounds__(256) void k1_welford4(const uint8_t *__restrict__ frames,
                                                   const uint32_t *__restrict__ idx, int N, size_t P,
                                                   uint8_t *__restrict__ mu,
                                                   uint8_t *__restrict__ sigma)
{
    size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; // dword index
    if (q * 4 >= P)
        return;
    float mean[4] = {0.f, 0.f, 0.f, 0.f}, m2[4] = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < N; k++) {
        size_t f = idx ? idx[k] : (uint32_t)k;
        uint32_t w = reinterpret_cast<const uint32_t *>(frames + f * P)[q];
        float kk = (float)(k + 1);
#pragma unroll
        for (int b = 0; b < 4; b++) {
            float x = (float)((w >> (8 * b)) & 0xffu);
            float delta = x - mean[b];
            mean[b] = mean[b] + delta / kk;
            m2[b] = m2[b] + delta * (x - mean[b]);
        }
    }
    uint32_t om = 0, os = 0;
    float nm1 = (float)(N - 1);
#pragma unroll
    for (int b = 0; b < 4; b++) {
        float sd = sqrtf(m2[b] / nm1);
        int isd = (sd != sd) ? 0 : (int)sd;
        os |= (uint32_t)(isd & 0xff) << (8 * b);
        om |= (uint32_t)((int)mean[b] & 0xff) << (8 * b);
    }
    reinterpret_cast<uint32_t *>(mu)[q] = om;
    reinterpret_cast<uint32_t *>(sigma)[q] = os;
}

extern "C" int abub_train_dev(const uint8_t *frames, const uint32_t *idx, int N, int W, int H,
                              uint8_t *mu, uint8_t *sigma, void *stream)
{
    if (!frames || !mu || !sigma || N <= 0 || W <= 0 || H <= 0)
        return set_err(ABUB_E_INVALID, "abub_train_dev: bad arguments");
    size_t P = (size_t)W * H;
    hipStream_t st = (hipStream_t)stream;
    if ((P & 3) == 0) {
        size_t nq = P / 4;
        hipLaunchKernelGGL(k1_welford4, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, st, frames,
                           idx, N, P, mu, sigma);
    } else {
        hipLaunchKernelGGL(k1_welford, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, st, frames, idx,
                           N, P, mu, sigma);
    }
    HIPCHK(hipGetLastError());
    return ABUB_OK;
}

// ------------------------------------------------------------------------------------------------
// K1b: 256-bin histogram of sat(f1 - f0), grid (blocks_per_pair, npairs)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k1b_pair_hist(const uint8_t *__restrict__ frames,
                                                     const abub_job *__restrict__ pairs, size_t P,
                                                     uint32_t *__restrict__ hist)
{
    __shared__ uint32_t lh[256];
    const abub_job jb = pairs[blockIdx.y];
    const uint8_t *f1 = frames + (size_t)jb.cur * P;
    const uint8_t *f0 = frames + (size_t)jb.ref * P;
    lh[threadIdx.x] = 0;
    __syncthreads();
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < P; i += stride) {
        int d = (int)f1[i] - (int)f0[i];
        if (d > 0)
            atomicAdd(&lh[d], 1u);
    }
    __syncthreads();
    uint32_t v = lh[threadIdx.x];
    if (v && threadIdx.x)
        atomicAdd(&hist[(size_t)jb.out * 256 + threadIdx.x], v);
}

extern "C" int abub_pair_hist_dev(const uint8_t *frames, const abub_job *pairs, int npairs, int W,
                                  int H, uint32_t *hist, void *stream)
{
    if (!frames || !pairs || !hist || npairs < 0 || W <= 0 || H <= 0)
        return set_err(ABUB_E_INVALID, "abub_pair_hist_dev: bad arguments");
    if (npairs == 0)
        return ABUB_OK;
    if (npairs > 65535)
        return set_err(ABUB_E_INVALID, "abub_pair_hist_dev: npairs > 65535");
    hipStream_t st = (hipStream_t)stream;
    size_t P = (size_t)W * H;
    HIPCHK(hipMemsetAsync(hist, 0, (size_t)npairs * 256 * sizeof(uint32_t), st));
    int bx = (int)((P + 256 * 16 - 1) / (256 * 16));
    if (bx < 1)
        bx = 1;
    hipLaunchKernelGGL(k1b_pair_hist, dim3(bx, npairs), dim3(256), 0, st, frames, pairs, P, hist);
    HIPCHK(hipGetLastError());
    hipLaunchKernelGGL(k_hist_bin0, dim3(npairs), dim3(64), 0, st, hist, (uint32_t)P);
    HIPCHK(hipGetLastError());
    return ABUB_OK;
}

// ------------------------------------------------------------------------------------------------
// K4: binarize + foreground compaction.  grid (blocks, nimg); 16 pixels per thread per step when
// the image size allows it.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k4_compact(const uint8_t *__restrict__ img, size_t P,
                                                  const int32_t *__restrict__ thr,
                                                  uint32_t *__restrict__ idx, int cap,
                                                  uint32_t *__restrict__ count)
{
    const int k = blockIdx.y;
    const uint8_t *im = img + (size_t)k * P;
    const int t = thr[k];
    uint32_t *cnt = count + k;
    uint32_t *out = idx + (size_t)k * cap;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    if ((P & 15) == 0) {
        size_t nv = P / 16;
        for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < nv; q += stride) {
            uint4 w = reinterpret_cast<const uint4 *>(im)[q];
            uint32_t ww[4] = {w.x, w.y, w.z, w.w};
            // v > t  for any byte?  quick reject: all bytes <= t
#pragma unroll
            for (int d = 0; d < 4; d++) {
                if (ww[d] == 0)
                    continue;
#pragma unroll
                for (int b = 0; b < 4; b++) {
                    int v = (ww[d] >> (8 * b)) & 0xff;
                    if (v > t) {
                        uint32_t pos = atomicAdd(cnt, 1u);
                        if (pos < (uint32_t)cap)
                            out[pos] = (uint32_t)(q * 16 + d * 4 + b);
                    }
                }
            }
        }
    } else {
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < P; i += stride) {
            if ((int)im[i] > t) {
                uint32_t pos = atomicAdd(cnt, 1u);
                if (pos < (uint32_t)cap)
                    out[pos] = (uint32_t)i;
            }
        }
    }
}

extern "C" int abub_fg_compact_dev(const uint8_t *img, int nimg, int W, int H, const int32_t *thr,
                                   uint32_t *idx, int cap, uint32_t *count, void *stream)
{
    if (!img || !thr || !idx || !count || nimg < 0 || W <= 0 || H <= 0 || cap <= 0)
        return set_err(ABUB_E_INVALID, "abub_fg_compact_dev: bad arguments");
    if (nimg == 0)
        return ABUB_OK;
    if (nimg > 65535)
        return set_err(ABUB_E_INVALID, "abub_fg_compact_dev: nimg > 65535");
    hipStream_t st = (hipStream_t)stream;
    size_t P = (size_t)W * H;
    HIPCHK(hipMemsetAsync(count, 0, (size_t)nimg * sizeof(uint32_t), st));
    int bx = (int)((P / 16 + 255) / 256);
    if (bx < 1)
        bx = 1;
    if (bx > 512)
        bx = 512;
    hipLaunchKernelGGL(k4_compact, dim3(bx, nimg), dim3(256), 0, st, img, P, thr, idx, cap, count);
    HIPCHK(hipGetLastError());
    return ABUB_OK;
}

// K4 batched: shared (image, index) pair list
__global__ __launch_bounds__(256) void k4_compact_pairs(const uint8_t *__restrict__ img, size_t P,
                                                        const int32_t *__restrict__ thr,
                                                        uint32_t *__restrict__ pairs, uint32_t cap,
                                                        uint32_t *__restrict__ count)
{
    const uint32_t k = blockIdx.y;
    const uint8_t *im = img + (size_t)k * P;
    const int t = thr[k];
    size_t stride = (size_t)gridDim.x * blockDim.x;
    if ((P & 15) == 0) {
        size_t nv = P / 16;
        for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < nv; q += stride) {
            uint4 w = reinterpret_cast<const uint4 *>(im)[q];
            uint32_t ww[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
            for (int d = 0; d < 4; d++) {
                if (ww[d] == 0)
                    continue;
#pragma unroll
                for (int b = 0; b < 4; b++) {
                    int v = (ww[d] >> (8 * b)) & 0xff;
                    if (v > t) {
                        uint32_t pos = atomicAdd(count, 1u);
                        if (pos < cap) {
                            pairs[2 * (size_t)pos] = k | ((uint32_t)v << 24);
                            pairs[2 * (size_t)pos + 1] = (uint32_t)(q * 16 + d * 4 + b);
                        }
                    }
                }
            }
        }
    } else {
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < P; i += stride) {
            if ((int)im[i] > t) {
                uint32_t pos = atomicAdd(count, 1u);
                if (pos < cap) {
                    pairs[2 * (size_t)pos] = k | ((uint32_t)im[i] << 24);
                    pairs[2 * (size_t)pos + 1] = (uint32_t)i;
                }
            }
        }
    }
}

extern "C" int abub_fg_compact_pairs_dev(const uint8_t *img, int nimg, int W, int H, const int32_t *thr,
                                         uint32_t *pairs, uint32_t cap, uint32_t *count, void *stream)
{
    if (!img || !thr || !pairs || !count || nimg < 0 || W <= 0 || H <= 0 || cap == 0)
        return set_err(ABUB_E_INVALID, "abub_fg_compact_pairs_dev: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    HIPCHK(hipMemsetAsync(count, 0, sizeof(uint32_t), st));
    if (nimg == 0)
        return ABUB_OK;
    size_t P = (size_t)W * H;
    int bx = (int)((P / 16 + 255) / 256);
    if (bx < 1)
        bx = 1;
    if (bx > 64)
        bx = 64;
    for (int base = 0; base < nimg; base += 65535) {
        int n = nimg - base < 65535 ? nimg - base : 65535;
        if (base != 0)
            return set_err(ABUB_E_INVALID, "abub_fg_compact_pairs_dev: nimg > 65535");
        hipLaunchKernelGGL(k4_compact_pairs, dim3(bx, n), dim3(256), 0, st, img, P, thr, pairs, cap, count);
    }
    HIPCHK(hipGetLastError());
    return ABUB_OK;
}

// ------------------------------------------------------------------------------------------------
// Group the shared candidate list by slot on the device (counting sort), so that the host receives
// every image's pixels as one contiguous run: count per slot -> exclusive scan -> scatter.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pairs_count(const uint32_t *__restrict__ pairs,
                                                     const uint32_t *__restrict__ count, uint32_t cap,
                                                     uint32_t nslots, uint32_t *__restrict__ slotcount)
{
    uint32_t n = *count;
    if (n > cap)
        n = cap;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        uint32_t s = pairs[2 * (size_t)i] & 0x00ffffffu;
        if (s < nslots)
            atomicAdd(&slotcount[s], 1u);
    }
}

// single block: offsets[s] = sum_{t<s} slotcount[t], offsets[nslots] = total; cursor = offsets
__global__ __launch_bounds__(1024) void k_pairs_scan(const uint32_t *__restrict__ slotcount, uint32_t nslots,
                                                     uint32_t *__restrict__ offsets, uint32_t *__restrict__ cursor)
{
    __shared__ uint32_t part[1024];
    const uint32_t t = threadIdx.x;
    const uint32_t per = (nslots + 1023) / 1024;
    const uint32_t lo = t * per, hi = lo + per < nslots ? lo + per : nslots;
    uint32_t sum = 0;
    for (uint32_t i = lo; i < hi; i++)
        sum += slotcount[i];
    part[t] = sum;
    __syncthreads();
    for (uint32_t o = 1; o < 1024; o <<= 1) {
        uint32_t v = t >= o ? part[t - o] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    uint32_t run = part[t] - sum;
    for (uint32_t i = lo; i < hi; i++) {
        offsets[i] = run;
        cursor[i] = run;
        run += slotcount[i];
    }
    if (t == 1023)
        offsets[nslots] = part[1023];
}

__global__ __launch_bounds__(256) void k_pairs_scatter(const uint32_t *__restrict__ pairs,
                                                       const uint32_t *__restrict__ count, uint32_t cap,
                                                       uint32_t nslots, uint32_t *__restrict__ cursor,
                                                       uint32_t *__restrict__ idx_out, uint8_t *__restrict__ val_out)
{
    uint32_t n = *count;
    if (n > cap)
        n = cap;
    const uint32_t stride = gridDim.x * blockDim.x;
    // whole waves iterate together (the loop bound is rounded up per wave): the shuffles below need all lanes
    const uint32_t nround = (n + 63u) & ~63u;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < nround; i += stride) {
        const bool live = i < n;
        uint32_t w0 = 0, w1 = 0, s = 0xffffffffu;
        if (live) {
            w0 = pairs[2 * (size_t)i];
            w1 = pairs[2 * (size_t)i + 1];
            s = w0 & 0x00ffffffu;
            if (s >= nslots)
                s = 0xffffffffu;
        }
        // one atomic per RUN of equal slots inside the wave (a producer wave-row appends its entries as one
        // contiguous run): the run's first lane reserves for all of it, the others take their rank in the run
        const int lane = threadIdx.x & 63;
        const uint32_t prev = __shfl_up(s, 1);
        const bool head = lane == 0 || s != prev;
        const unsigned long long heads = __builtin_amdgcn_ballot_w64(head);
        const unsigned long long upto = heads & (~0ull >> (63 - lane));   // heads at or below this lane
        const int leader = 63 - __builtin_clzll(upto);                    // lane 0 is always a head
        const unsigned long long above = leader < 63 ? heads >> (leader + 1) : 0ull;
        const int len = above ? __builtin_ctzll(above) + 1 : 64 - leader; // run length
        uint32_t base = 0;
        if (head && s != 0xffffffffu)
            base = atomicAdd(&cursor[s], (uint32_t)len);
        base = __shfl(base, leader);
        const uint32_t pos = base + (uint32_t)(lane - leader);
        if (s != 0xffffffffu && pos < cap) { // (offsets come from full counts: stay inside the buffers on overflow)
            idx_out[pos] = w1;
            val_out[pos] = (uint8_t)(w0 >> 24);
        }
    }
}

// counts per slot straight from the histograms the producing kernels already made: the number of
// list entries of slot s is the number of its pixels with value > cthr[s]
__global__ __launch_bounds__(64) void k_slot_counts_from_hist(const uint32_t *__restrict__ hist,
                                                              const int32_t *__restrict__ cthr, uint32_t nslots,
                                                              uint32_t *__restrict__ slotcount)
{
    const uint32_t sl = blockIdx.x;
    if (sl >= nslots)
        return;
    const uint32_t *h = hist + (size_t)sl * 256;
    const int t = cthr[sl];
    const int l = threadIdx.x;
    uint32_t c = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        int b = l + 64 * k;
        if (b > t)
            c += h[b];
    }
    for (int o = 32; o > 0; o >>= 1)
        c += __shfl_xor(c, o);
    if (l == 0)
        slotcount[sl] = c;
}

static int pairs_group_impl(const uint32_t *pairs, const uint32_t *count, uint32_t cap, int nslots,
                            uint32_t *scratch, uint32_t *offsets, uint32_t *idx_out, uint8_t *val_out,
                            const uint32_t *hist, const int32_t *cthr, void *stream)
{
    if (!pairs || !count || !scratch || !offsets || !idx_out || !val_out || nslots <= 0 || cap == 0)
        return set_err(ABUB_E_INVALID, "abub_pairs_group_dev: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    uint32_t *slotcount = scratch, *cursor = scratch + nslots;
    if (hist && cthr) {
        hipLaunchKernelGGL(k_slot_counts_from_hist, dim3(nslots), dim3(64), 0, st, hist, cthr, (uint32_t)nslots, slotcount);
    } else {
        HIPCHK(hipMemsetAsync(slotcount, 0, (size_t)nslots * sizeof(uint32_t), st));
        hipLaunchKernelGGL(k_pairs_count, dim3(512), dim3(256), 0, st, pairs, count, cap, (uint32_t)nslots, slotcount);
    }
    hipLaunchKernelGGL(k_pairs_scan, dim3(1), dim3(1024), 0, st, slotcount, (uint32_t)nslots, offsets, cursor);
    hipLaunchKernelGGL(k_pairs_scatter, dim3(512), dim3(256), 0, st, pairs, count, cap, (uint32_t)nslots, cursor,
                       idx_out, val_out);
    HIPCHK(hipGetLastError());
    return ABUB_OK;
}

extern "C" int abub_pairs_group_dev(const uint32_t *pairs, const uint32_t *count, uint32_t cap, int nslots,
                                    uint32_t *scratch /* [2*nslots] */, uint32_t *offsets /* [nslots+1] */,
                                    uint32_t *idx_out /* [cap] */, uint8_t *val_out /* [cap] */, void *stream)
{
    return pairs_group_impl(pairs, count, cap, nslots, scratch, offsets, idx_out, val_out, nullptr, nullptr, stream);
}

extern "C" int abub_pairs_group_hist_dev(const uint32_t *pairs, const uint32_t *count, uint32_t cap, int nslots,
                                         uint32_t *scratch, uint32_t *offsets, uint32_t *idx_out, uint8_t *val_out,
                                         const uint32_t *hist, const int32_t *cthr, void *stream)
{
    if (!hist || !cthr)
        return set_err(ABUB_E_INVALID, "abub_pairs_group_hist_dev: bad arguments");
    return pairs_group_impl(pairs, count, cap, nslots, scratch, offsets, idx_out, val_out, hist, cthr, stream);
}

// ------------------------------------------------------------------------------------------------
// Bellows veto support (L3Localizer::TrackAFeature, L3Localizer.cpp:473-543): raw terms of
// cv::matchTemplate(CV_TM_CCORR_NORMED) -- for every placement (x,y) of the template the exact integer
// cross-correlation sum(T*I) and window energy sum(I*I).  The double-precision normalisation, the min-max
// normalise and the sub-pixel centre of mass are host work on the (small) result matrix.
// One thread = 4 adjacent x placements (sliding 4-byte window), one workgroup row = one result row.
// Per template row the partial sums stay in u32 (<= 2048*65025*... guarded by tw <= 4096), then widen.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_match_ccorr(const uint8_t *__restrict__ img, int W, int H,
                                                    const uint8_t *__restrict__ tmpl, int tw, int th, int rw, int rh,
                                                    unsigned long long *__restrict__ num,
                                                    unsigned long long *__restrict__ wsum2)
{
    const int y = blockIdx.y;
    const int x0 = (blockIdx.x * 64 + threadIdx.x) * 4;
    if (x0 >= rw)
        return;
    unsigned long long acc[4] = {0, 0, 0, 0}, sq[4] = {0, 0, 0, 0};
    for (int r = 0; r < th; r++) {
        const uint8_t *irow = img + (size_t)(y + r) * W;
        const uint8_t *trow = tmpl + (size_t)r * tw;
        uint32_t a[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
        // window bytes f[k] = I(x0 + c + k); positions beyond the row end are clamped (their results are discarded)
        uint32_t f0 = irow[min(x0, W - 1)], f1 = irow[min(x0 + 1, W - 1)], f2 = irow[min(x0 + 2, W - 1)];
        for (int c = 0; c < tw; c++) {
            const uint32_t f3 = irow[min(x0 + c + 3, W - 1)];
            const uint32_t t = trow[c];
            a[0] += t * f0;
            a[1] += t * f1;
            a[2] += t * f2;
            a[3] += t * f3;
            q[0] += f0 * f0;
            q[1] += f1 * f1;
            q[2] += f2 * f2;
            q[3] += f3 * f3;
            f0 = f1;
            f1 = f2;
            f2 = f3;
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            acc[k] += a[k];
            sq[k] += q[k];
        }
    }
#pragma unroll
    for (int k = 0; k < 4; k++)
        if (x0 + k < rw) {
            num[(size_t)y * rw + x0 + k] = acc[k];
            wsum2[(size_t)y * rw + x0 + k] = sq[k];
        }
}

extern "C" int abub_match_ccorr_dev(const uint8_t *img, int W, int H, const uint8_t *tmpl, int tw, int th,
                                    unsigned long long *num, unsigned long long *wsum2, void *stream)
{
    if (!img || !tmpl || !num || !wsum2 || W <= 0 || H <= 0 || tw <= 0 || th <= 0 || tw > W || th > H || tw > 4096)
        return set_err(ABUB_E_INVALID, "abub_match_ccorr_dev: bad arguments");
    const int rw = W - tw + 1, rh = H - th + 1;
    dim3 grid((rw + 255) / 256, rh), block(64);
    if (rh > 65535)
        return set_err(ABUB_E_INVALID, "abub_match_ccorr_dev: image too tall");
    hipLaunchKernelGGL(k_match_ccorr, grid, block, 0, (hipStream_t)stream, img, W, H, tmpl, tw, th, rw, rh, num, wsum2);
    HIPCHK(hipGetLastError());
    return ABUB_OK;
}

// img = sat(img - sub) in place + 256-bin histogram (`overTheSigma -= diff_frame`, L3Localizer.cpp:362)
__global__ __launch_bounds__(256) void k_subsat_hist(uint8_t *__restrict__ img, const uint8_t *__restrict__ sub, size_t n,
                                                     uint32_t *__restrict__ hist)
{
    __shared__ uint32_t lh[256];
    lh[threadIdx.x] = 0;
    __syncthreads();
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        int v = (int)img[i] - (int)sub[i];
        v = v < 0 ? 0 : v;
        img[i] = (uint8_t)v;
        if (v)
            atomicAdd(&lh[v], 1u);
    }
    __syncthreads();
    uint32_t c = lh[threadIdx.x];
    if (c && threadIdx.x)
        atomicAdd(&hist[threadIdx.x], c);
}

extern "C" int abub_subsat_hist_dev(uint8_t *img, const uint8_t *sub, int W, int H, uint32_t *hist, void *stream)
{
    if (!img || !sub || !hist || W <= 0 || H <= 0)
        return set_err(ABUB_E_INVALID, "abub_subsat_hist_dev: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    size_t n = (size_t)W * H;
    HIPCHK(hipMemsetAsync(hist, 0, 256 * sizeof(uint32_t), st));
    int blocks = (int)((n + 256 * 16 - 1) / (256 * 16));
    hipLaunchKernelGGL(k_subsat_hist, dim3(blocks < 1 ? 1 : blocks), dim3(256), 0, st, img, sub, n, hist);
    HIPCHK(hipGetLastError());
    hipLaunchKernelGGL(k_hist_bin0, dim3(1), dim3(64), 0, st, hist, (uint32_t)n);
    HIPCHK(hipGetLastError());
    return ABUB_OK;
}
