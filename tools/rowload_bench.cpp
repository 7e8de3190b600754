// rowload_bench -- how fast the memory system feeds one-wave-per-row-chunk kernels under different lane->byte
// mappings of a row, with the arithmetic removed (rows are only OR-ed together).  Decides the lane mapping of K2/K3.
//   MAP 0  blocked: lane L owns bytes [4*NDW*L, 4*NDW*(L+1))  (round-1 mapping: dwordx4 + dword at a 20-byte stride)
//   MAP 1  split:   segment A = dwordx4 at 16*L, then dwordx2 / dword segments for the rest of the row (every
//                   load instruction covers one contiguous span of the row)
//   MAP 2  dwords:  NDW dword loads at 4*(L + 64*k)
// Patterns: single = 3 row streams per job (cur, ref, sigma6); chain2 = one wave serves 2 jobs sharing a frame
// (3 frames + sigma6 per step); chain2+store = the same plus two zero rows stored per step (store mode).
// Build: hipcc --offload-arch=gfx950 -O3 tools/rowload_bench.cpp -o tools/rowload_bench
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int W, int MAP>
struct RowMap;

// ---- W = 1280 -------------------------------------------------------------------------------------------------
template <>
struct RowMap<1280, 0> {
    static constexpr int N = 5;
    static __device__ __forceinline__ void load(uint32_t (&r)[N], const uint8_t *row, int lane)
    {
        const uint32_t *p = (const uint32_t *)(row + 20 * lane);
#pragma unroll
        for (int d = 0; d < 5; d++) r[d] = p[d];
    }
    static __device__ __forceinline__ void store0(uint8_t *row, int lane)
    {
        uint32_t *p = (uint32_t *)(row + 20 * lane);
#pragma unroll
        for (int d = 0; d < 5; d++) p[d] = 0;
    }
    static __device__ __forceinline__ void store0nt(uint8_t *row, int lane)
    {
        uint32_t *p = (uint32_t *)(row + 20 * lane);
#pragma unroll
        for (int d = 0; d < 5; d++) __builtin_nontemporal_store(0u, &p[d]);
    }
};
template <>
struct RowMap<1280, 1> {
    static constexpr int N = 5;
    static __device__ __forceinline__ void load(uint32_t (&r)[N], const uint8_t *row, int lane)
    {
        const uint4 a = *(const uint4 *)(row + 16 * lane);
        r[0] = a.x; r[1] = a.y; r[2] = a.z; r[3] = a.w;
        r[4] = *(const uint32_t *)(row + 1024 + 4 * lane);
    }
    static __device__ __forceinline__ void store0(uint8_t *row, int lane)
    {
        *(uint4 *)(row + 16 * lane) = make_uint4(0, 0, 0, 0);
        *(uint32_t *)(row + 1024 + 4 * lane) = 0;
    }
    static __device__ __forceinline__ void store0nt(uint8_t *row, int lane)
    {
        { uint4 *q_ = (uint4 *)(row + 16 * lane); __builtin_nontemporal_store(0u, &q_->x); __builtin_nontemporal_store(0u, &q_->y); __builtin_nontemporal_store(0u, &q_->z); __builtin_nontemporal_store(0u, &q_->w); }
        __builtin_nontemporal_store(0u, (uint32_t *)(row + 1024 + 4 * lane));
    }
};
template <>
struct RowMap<1280, 2> {
    static constexpr int N = 5;
    static __device__ __forceinline__ void load(uint32_t (&r)[N], const uint8_t *row, int lane)
    {
#pragma unroll
        for (int d = 0; d < 5; d++) r[d] = *(const uint32_t *)(row + 256 * d + 4 * lane);
    }
    static __device__ __forceinline__ void store0(uint8_t *row, int lane)
    {
#pragma unroll
        for (int d = 0; d < 5; d++) *(uint32_t *)(row + 256 * d + 4 * lane) = 0;
    }
    static __device__ __forceinline__ void store0nt(uint8_t *row, int lane)
    {
#pragma unroll
        for (int d = 0; d < 5; d++) __builtin_nontemporal_store(0u, (uint32_t *)(row + 256 * d + 4 * lane));
    }
};
// ---- W = 1680 -------------------------------------------------------------------------------------------------
template <>
struct RowMap<1680, 0> { // 60 lanes x 28 bytes
    static constexpr int N = 7;
    static __device__ __forceinline__ void load(uint32_t (&r)[N], const uint8_t *row, int lane)
    {
        const int l = lane < 60 ? lane : 0;
        const uint32_t *p = (const uint32_t *)(row + 28 * l);
#pragma unroll
        for (int d = 0; d < 7; d++) r[d] = p[d];
    }
    static __device__ __forceinline__ void store0(uint8_t *row, int lane)
    {
        if (lane < 60) {
            uint32_t *p = (uint32_t *)(row + 28 * lane);
#pragma unroll
            for (int d = 0; d < 7; d++) p[d] = 0;
        }
    }
    static __device__ __forceinline__ void store0nt(uint8_t *row, int lane)
    {
        if (lane < 60) {
            uint32_t *p = (uint32_t *)(row + 28 * lane);
#pragma unroll
            for (int d = 0; d < 7; d++) __builtin_nontemporal_store(0u, &p[d]);
        }
    }
};
template <>
struct RowMap<1680, 1> { // 64 x 16 + 64 x 8 + 36 x 4
    static constexpr int N = 7;
    static __device__ __forceinline__ void load(uint32_t (&r)[N], const uint8_t *row, int lane)
    {
        const uint4 a = *(const uint4 *)(row + 16 * lane);
        r[0] = a.x; r[1] = a.y; r[2] = a.z; r[3] = a.w;
        const uint2 b = *(const uint2 *)(row + 1024 + 8 * lane);
        r[4] = b.x; r[5] = b.y;
        const int l = lane < 36 ? lane : 0;
        r[6] = *(const uint32_t *)(row + 1536 + 4 * l);
    }
    static __device__ __forceinline__ void store0(uint8_t *row, int lane)
    {
        *(uint4 *)(row + 16 * lane) = make_uint4(0, 0, 0, 0);
        *(uint2 *)(row + 1024 + 8 * lane) = make_uint2(0, 0);
        if (lane < 36) *(uint32_t *)(row + 1536 + 4 * lane) = 0;
    }
    static __device__ __forceinline__ void store0nt(uint8_t *row, int lane)
    {
        { uint4 *q_ = (uint4 *)(row + 16 * lane); __builtin_nontemporal_store(0u, &q_->x); __builtin_nontemporal_store(0u, &q_->y); __builtin_nontemporal_store(0u, &q_->z); __builtin_nontemporal_store(0u, &q_->w); }
        { uint2 *q_ = (uint2 *)(row + 1024 + 8 * lane); __builtin_nontemporal_store(0u, &q_->x); __builtin_nontemporal_store(0u, &q_->y); }
        if (lane < 36) __builtin_nontemporal_store(0u, (uint32_t *)(row + 1536 + 4 * lane));
    }
};
template <>
struct RowMap<1680, 2> { // 64 x 16 + 41 x 16
    static constexpr int N = 8;
    static __device__ __forceinline__ void load(uint32_t (&r)[N], const uint8_t *row, int lane)
    {
        const uint4 a = *(const uint4 *)(row + 16 * lane);
        r[0] = a.x; r[1] = a.y; r[2] = a.z; r[3] = a.w;
        const int l = lane < 41 ? lane : 0;
        const uint4 b = *(const uint4 *)(row + 1024 + 16 * l);
        r[4] = b.x; r[5] = b.y; r[6] = b.z; r[7] = b.w;
    }
    static __device__ __forceinline__ void store0(uint8_t *row, int lane)
    {
        *(uint4 *)(row + 16 * lane) = make_uint4(0, 0, 0, 0);
        if (lane < 41) *(uint4 *)(row + 1024 + 16 * lane) = make_uint4(0, 0, 0, 0);
    }
    static __device__ __forceinline__ void store0nt(uint8_t *row, int lane)
    {
        { uint4 *q_ = (uint4 *)(row + 16 * lane); __builtin_nontemporal_store(0u, &q_->x); __builtin_nontemporal_store(0u, &q_->y); __builtin_nontemporal_store(0u, &q_->z); __builtin_nontemporal_store(0u, &q_->w); }
        if (lane < 41) { uint4 *q_ = (uint4 *)(row + 1024 + 16 * lane); __builtin_nontemporal_store(0u, &q_->x); __builtin_nontemporal_store(0u, &q_->y); __builtin_nontemporal_store(0u, &q_->z); __builtin_nontemporal_store(0u, &q_->w); }
    }
};

// single: job j reads frames j+2 (cur), j (ref) and sigma6
template <int W, int MAP>
__global__ __launch_bounds__(64) void rd_single(const uint8_t *__restrict__ frames, const uint8_t *__restrict__ sg, int H, int R,
                                                int nchunks, uint32_t *out)
{
    using M = RowMap<W, MAP>;
    const int lane = threadIdx.x, unit = blockIdx.x, job = unit / nchunks, chunk = unit - job * nchunks;
    const size_t P = (size_t)W * H;
    const uint8_t *cur = frames + (size_t)(job + 2) * P, *ref = frames + (size_t)job * P;
    const int y0 = chunk * R;
    int y1 = y0 + R; if (y1 > H) y1 = H;
    uint32_t acc = 0;
    uint32_t buf[2][3][M::N];
    auto load = [&](int slot, int y) {
        const size_t o = (size_t)(y < H ? y : H - 1) * W;
        M::load(buf[slot][0], cur + o, lane); M::load(buf[slot][1], ref + o, lane); M::load(buf[slot][2], sg + o, lane);
    };
    load(0, y0);
    for (int y = y0; y < y1; y += 2) {
#pragma unroll
        for (int u = 0; u < 2; u++) {
            load(u ^ 1, y + u + 1);
#pragma unroll
            for (int d = 0; d < M::N; d++) acc |= buf[u][0][d] ^ buf[u][1][d] ^ buf[u][2][d];
        }
    }
    if (acc == 0x12345678u) out[unit] = acc;
}

// chain of K jobs per wave: frames base, base+2, .., base+2K (+ sigma6); STORE: K zero rows stored per step
template <int W, int MAP, int K, int STORE>
__global__ __launch_bounds__(64) void rd_chain(const uint8_t *__restrict__ frames, const uint8_t *__restrict__ sg, int H, int R,
                                               int nchunks, uint32_t *out, uint8_t *__restrict__ diff)
{
    using M = RowMap<W, MAP>;
    const int lane = threadIdx.x, unit = blockIdx.x, ch = unit / nchunks, chunk = unit - ch * nchunks;
    const size_t P = (size_t)W * H;
    const int base = (ch / 2) * 2 * K + (ch & 1); // first job of the chain
    const int y0 = chunk * R;
    int y1 = y0 + R; if (y1 > H) y1 = H;
    uint32_t acc = 0;
    uint32_t buf[2][K + 2][M::N];
    auto load = [&](int slot, int y) {
        const size_t o = (size_t)(y < H ? y : H - 1) * W;
#pragma unroll
        for (int f = 0; f <= K; f++) M::load(buf[slot][f], frames + (size_t)(base + 2 * f) * P + o, lane);
        M::load(buf[slot][K + 1], sg + o, lane);
    };
    load(0, y0);
    for (int y = y0; y < y1; y += 2) {
#pragma unroll
        for (int u = 0; u < 2; u++) {
            load(u ^ 1, y + u + 1);
#pragma unroll
            for (int f = 0; f < K + 2; f++)
#pragma unroll
                for (int d = 0; d < M::N; d++) acc |= buf[u][f][d] + (uint32_t)f;
            if (STORE && y + u < y1) {
#pragma unroll
                for (int f = 0; f < K; f++) {
                    uint8_t *row = diff + (size_t)(base + 2 * f) * P + (size_t)(y + u) * W;
                    if (STORE == 2) M::store0nt(row, lane); else M::store0(row, lane);
                }
            }
        }
    }
    if (acc == 0x12345678u) out[unit] = acc;
}

// the chain2 pattern at a fixed occupancy (waves per SIMD), to separate "not enough loads in flight" from bandwidth;
// VALU = extra dependent-free VALU instructions per loaded dword (stand-in for the scan's arithmetic)
template <int W, int MAP, int OCC, int VALU>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(OCC, OCC))) void rd_chain_occ(const uint8_t *__restrict__ frames, const uint8_t *__restrict__ sg, int H, int R,
                                               int nchunks, uint32_t *out)
{
    using M = RowMap<W, MAP>;
    constexpr int K = 2;
    const int lane = threadIdx.x, unit = blockIdx.x, ch = unit / nchunks, chunk = unit - ch * nchunks;
    const size_t P = (size_t)W * H;
    const int base = (ch / 2) * 2 * K + (ch & 1);
    const int y0 = chunk * R;
    int y1 = y0 + R; if (y1 > H) y1 = H;
    uint32_t acc = 0;
    uint32_t buf[2][K + 2][M::N];
    auto load = [&](int slot, int y) {
        const size_t o = (size_t)(y < H ? y : H - 1) * W;
#pragma unroll
        for (int f = 0; f <= K; f++) M::load(buf[slot][f], frames + (size_t)(base + 2 * f) * P + o, lane);
        M::load(buf[slot][K + 1], sg + o, lane);
    };
    load(0, y0);
    for (int y = y0; y < y1; y += 2) {
#pragma unroll
        for (int u = 0; u < 2; u++) {
            load(u ^ 1, y + u + 1);
#pragma unroll
            for (int f = 0; f < K + 2; f++)
#pragma unroll
                for (int d = 0; d < M::N; d++) {
                    uint32_t v = buf[u][f][d];
#pragma unroll
                    for (int q = 0; q < VALU; q++) v = __builtin_amdgcn_perm(v, acc, 0x07020500u + q); // slow-rate op
                    acc |= v + (uint32_t)f;
                }
        }
    }
    if (acc == 0x12345678u) out[unit] = acc;
}

static hipEvent_t e0, e1;
template <typename F>
static float timeit(F launch, int reps)
{
    launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; i++) launch();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

template <int W, int MAP>
static void run(const uint8_t *slab, const uint8_t *sg, uint32_t *out, uint8_t *diff, int F, int H, int R)
{
    if (R <= 0) R = (H + 7) / 8; // 8 chunks per frame: chunk id == XCD id, like the kernels' default
    const int njobs = F - 2, nch = (H + R - 1) / R, reps = 5;
    const double P = (double)W * H;
    float ms = timeit([&] { hipLaunchKernelGGL((rd_single<W, MAP>), dim3(njobs * nch), dim3(64), 0, 0, slab, sg, H, R, nch, out); }, reps);
    printf("{\"W\": %d, \"map\": %d, \"pattern\": \"single\", \"jobs\": %d, \"ms\": %.4f, \"us_per_job\": %.4f, \"compulsory_TBps\": %.3f}\n", W, MAP,
           njobs, ms, 1e3 * ms / njobs, P * njobs / ms / 1e9);
    const int nchains = (njobs / 4) * 2;
    ms = timeit([&] { hipLaunchKernelGGL((rd_chain<W, MAP, 2, 0>), dim3(nchains * nch), dim3(64), 0, 0, slab, sg, H, R, nch, out, diff); }, reps);
    printf("{\"W\": %d, \"map\": %d, \"pattern\": \"chain2\", \"jobs\": %d, \"ms\": %.4f, \"us_per_job\": %.4f, \"compulsory_TBps\": %.3f}\n", W, MAP,
           nchains * 2, ms, 1e3 * ms / (nchains * 2), P * nchains * 2 / ms / 1e9);
    ms = timeit([&] { hipLaunchKernelGGL((rd_chain<W, MAP, 2, 1>), dim3(nchains * nch), dim3(64), 0, 0, slab, sg, H, R, nch, out, diff); }, reps);
    printf("{\"W\": %d, \"map\": %d, \"pattern\": \"chain2+store\", \"jobs\": %d, \"ms\": %.4f, \"us_per_job\": %.4f, \"compulsory_TBps\": %.3f}\n", W,
           MAP, nchains * 2, ms, 1e3 * ms / (nchains * 2), 2 * P * nchains * 2 / ms / 1e9);
    ms = timeit([&] { hipLaunchKernelGGL((rd_chain<W, MAP, 2, 2>), dim3(nchains * nch), dim3(64), 0, 0, slab, sg, H, R, nch, out, diff); }, reps);
    printf("{\"W\": %d, \"map\": %d, \"pattern\": \"chain2+ntstore\", \"jobs\": %d, \"ms\": %.4f, \"us_per_job\": %.4f, \"compulsory_TBps\": %.3f}\n", W,
           MAP, nchains * 2, ms, 1e3 * ms / (nchains * 2), 2 * P * nchains * 2 / ms / 1e9);
    const int nchains3 = (njobs / 6) * 2;
    ms = timeit([&] { hipLaunchKernelGGL((rd_chain<W, MAP, 3, 0>), dim3(nchains3 * nch), dim3(64), 0, 0, slab, sg, H, R, nch, out, diff); }, reps);
    printf("{\"W\": %d, \"map\": %d, \"pattern\": \"chain3\", \"jobs\": %d, \"ms\": %.4f, \"us_per_job\": %.4f, \"compulsory_TBps\": %.3f}\n", W, MAP,
           nchains3 * 3, ms, 1e3 * ms / (nchains3 * 3), P * nchains3 * 3 / ms / 1e9);
    ms = timeit([&] { hipLaunchKernelGGL((rd_chain<W, MAP, 3, 1>), dim3(nchains3 * nch), dim3(64), 0, 0, slab, sg, H, R, nch, out, diff); }, reps);
    printf("{\"W\": %d, \"map\": %d, \"pattern\": \"chain3+store\", \"jobs\": %d, \"ms\": %.4f, \"us_per_job\": %.4f, \"compulsory_TBps\": %.3f}\n", W, MAP,
           nchains3 * 3, ms, 1e3 * ms / (nchains3 * 3), 2 * P * nchains3 * 3 / ms / 1e9);
    const int nchains4 = (njobs / 8) * 2;
    ms = timeit([&] { hipLaunchKernelGGL((rd_chain<W, MAP, 4, 0>), dim3(nchains4 * nch), dim3(64), 0, 0, slab, sg, H, R, nch, out, diff); }, reps);
    printf("{\"W\": %d, \"map\": %d, \"pattern\": \"chain4\", \"jobs\": %d, \"ms\": %.4f, \"us_per_job\": %.4f, \"compulsory_TBps\": %.3f}\n", W, MAP,
           nchains4 * 4, ms, 1e3 * ms / (nchains4 * 4), P * nchains4 * 4 / ms / 1e9);
}

// The chip's own streaming ceilings, for scale: every lane reads (and, in the copy, writes) 16 contiguous bytes per
// step, the grid strides over the whole slab -- no rows, no chunks, no reuse.
__global__ __launch_bounds__(256) void rd_flat(const uint4 *__restrict__ src, size_t n16, uint32_t *__restrict__ out)
{
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
        const uint4 v = src[i];
        acc.x ^= v.x;
        acc.y ^= v.y;
        acc.z ^= v.z;
        acc.w ^= v.w;
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u)
        out[0] = 1;
}
__global__ __launch_bounds__(256) void copy_flat(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n16)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256)
        dst[i] = src[i];
}

// The copy in the shape MI355X_MICROARCH.md quotes for its 6.29 TB/s figure: UNR independent 16-byte loads per lane in flight
// before the first store, a grid of a few waves per SIMD striding over contiguous UNR x 4 KiB tiles.  NT bit 0:
// nontemporal loads, bit 1: nontemporal stores.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <int UNR, int NT>
__global__ __launch_bounds__(256) void copy_guide(const u32x4 *__restrict__ src, u32x4 *__restrict__ dst, size_t n16)
{
    const size_t tile = (size_t)UNR * 256;
    for (size_t base = (size_t)blockIdx.x * tile; base + tile <= n16; base += (size_t)gridDim.x * tile) {
        u32x4 v[UNR];
#pragma unroll
        for (int u = 0; u < UNR; u++)
            v[u] = (NT & 1) ? __builtin_nontemporal_load(&src[base + (size_t)u * 256 + threadIdx.x]) : src[base + (size_t)u * 256 + threadIdx.x];
#pragma unroll
        for (int u = 0; u < UNR; u++) {
            if (NT & 2)
                __builtin_nontemporal_store(v[u], &dst[base + (size_t)u * 256 + threadIdx.x]);
            else
                dst[base + (size_t)u * 256 + threadIdx.x] = v[u];
        }
    }
}
// write-only and read-only streams in the same shape (what each direction gives alone)
template <int UNR, int NT>
__global__ __launch_bounds__(256) void fill_guide(u32x4 *__restrict__ dst, size_t n16)
{
    const size_t tile = (size_t)UNR * 256;
    const u32x4 z = {0u, 0u, 0u, 0u};
    for (size_t base = (size_t)blockIdx.x * tile; base + tile <= n16; base += (size_t)gridDim.x * tile) {
#pragma unroll
        for (int u = 0; u < UNR; u++) {
            if (NT & 2)
                __builtin_nontemporal_store(z, &dst[base + (size_t)u * 256 + threadIdx.x]);
            else
                dst[base + (size_t)u * 256 + threadIdx.x] = z;
        }
    }
}
template <int UNR>
__global__ __launch_bounds__(256) void read_guide(const u32x4 *__restrict__ src, size_t n16, uint32_t *__restrict__ out)
{
    const size_t tile = (size_t)UNR * 256;
    u32x4 acc = {0u, 0u, 0u, 0u};
    for (size_t base = (size_t)blockIdx.x * tile; base + tile <= n16; base += (size_t)gridDim.x * tile) {
        u32x4 v[UNR];
#pragma unroll
        for (int u = 0; u < UNR; u++)
            v[u] = src[base + (size_t)u * 256 + threadIdx.x];
#pragma unroll
        for (int u = 0; u < UNR; u++)
            acc ^= v[u];
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u)
        out[0] = 1;
}

static void copy_ceiling(const uint8_t *slab, uint8_t *diff, uint32_t *out, int F)
{
    const size_t n16 = (size_t)1280 * 1024 * F / 16 / 2048 * 2048;
    const double GB = n16 * 16 / 1e9;
#define CG(UNR, NT, BLOCKS)                                                                                                     \
    {                                                                                                                           \
        float ms = timeit([&] { hipLaunchKernelGGL((copy_guide<UNR, NT>), dim3(BLOCKS), dim3(256), 0, 0, (const u32x4 *)slab, (u32x4 *)diff, n16); }, 5); \
        printf("{\"pattern\": \"guide copy\", \"loads_in_flight\": %d, \"nt\": %d, \"blocks\": %d, \"GB_read\": %.2f, \"ms\": %.4f, \"TBps_read_plus_write\": %.3f}\n", UNR, NT, BLOCKS, GB, ms, 2 * GB / ms); \
    }
    CG(4, 0, 1024) CG(4, 0, 2048) CG(4, 0, 4096) CG(4, 0, 8192) CG(4, 0, 16384)
    CG(8, 0, 1024) CG(8, 0, 2048) CG(8, 0, 4096)
    CG(2, 0, 4096) CG(1, 0, 8192)
    CG(4, 2, 1024) CG(4, 2, 2048) CG(4, 2, 4096) CG(4, 2, 8192)
    CG(4, 1, 2048) CG(4, 3, 2048) CG(4, 3, 4096) CG(8, 3, 2048) CG(8, 2, 2048)
#undef CG
    for (int blocks : {2048, 4096}) {
        float ms = timeit([&] { hipLaunchKernelGGL((fill_guide<4, 0>), dim3(blocks), dim3(256), 0, 0, (u32x4 *)diff, n16); }, 5);
        printf("{\"pattern\": \"guide fill\", \"nt\": 0, \"blocks\": %d, \"GB\": %.2f, \"ms\": %.4f, \"TBps\": %.3f}\n", blocks, GB, ms, GB / ms);
        ms = timeit([&] { hipLaunchKernelGGL((fill_guide<4, 2>), dim3(blocks), dim3(256), 0, 0, (u32x4 *)diff, n16); }, 5);
        printf("{\"pattern\": \"guide fill\", \"nt\": 2, \"blocks\": %d, \"GB\": %.2f, \"ms\": %.4f, \"TBps\": %.3f}\n", blocks, GB, ms, GB / ms);
        ms = timeit([&] { hipLaunchKernelGGL((read_guide<4>), dim3(blocks), dim3(256), 0, 0, (const u32x4 *)slab, n16, out); }, 5);
        printf("{\"pattern\": \"guide read\", \"blocks\": %d, \"GB\": %.2f, \"ms\": %.4f, \"TBps\": %.3f}\n", blocks, GB, ms, GB / ms);
    }
    {
        float ms = timeit([&] { CK(hipMemcpyAsync(diff, slab, n16 * 16, hipMemcpyDeviceToDevice, 0)); }, 5);
        printf("{\"pattern\": \"hipMemcpy D2D\", \"GB_read\": %.2f, \"ms\": %.4f, \"TBps_read_plus_write\": %.3f}\n", GB, ms, 2 * GB / ms);
        ms = timeit([&] { CK(hipMemsetAsync(diff, 0, n16 * 16, 0)); }, 5);
        printf("{\"pattern\": \"hipMemset\", \"GB\": %.2f, \"ms\": %.4f, \"TBps\": %.3f}\n", GB, ms, GB / ms);
    }
}

int main(int argc, char **argv)
{
    const int F = argc > 1 ? atoi(argv[1]) : 2000, R = argc > 2 ? atoi(argv[2]) : 0;
    const size_t Pmax = (size_t)1680 * 1050;
    uint8_t *slab, *sg, *diff;
    uint32_t *out;
    CK(hipMalloc(&slab, Pmax * F));
    CK(hipMalloc(&diff, Pmax * F));
    CK(hipMalloc(&sg, Pmax));
    CK(hipMalloc(&out, 4 * (size_t)F * 16));
    CK(hipMemset(slab, 1, Pmax * F));
    CK(hipMemset(sg, 2, Pmax));
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    if (argc > 3 && !strcmp(argv[3], "copy")) { // only the chip's copy / fill / read ceilings
        copy_ceiling(slab, diff, out, F);
        return 0;
    }
    {
        const int H = 1024, Rr = R > 0 ? R : 128, nch = (H + Rr - 1) / Rr, njobs = F - 2, nchains = (njobs / 4) * 2;
#define OCCRUN(MAP, OCC, VALU)                                                                                                   \
    {                                                                                                                            \
        float ms = timeit([&] { hipLaunchKernelGGL((rd_chain_occ<1280, MAP, OCC, VALU>), dim3(nchains * nch), dim3(64), 0, 0, slab, sg, H, Rr, nch, out); }, 5); \
        printf("{\"W\": 1280, \"map\": %d, \"pattern\": \"chain2@occ\", \"occ\": %d, \"valu_per_dword\": %d, \"ms\": %.4f, \"us_per_job\": %.4f}\n", MAP, OCC, VALU, ms, 1e3 * ms / (nchains * 2)); \
    }
        OCCRUN(0, 2, 0) OCCRUN(0, 3, 0) OCCRUN(0, 4, 0) OCCRUN(0, 6, 0) OCCRUN(0, 8, 0)
        OCCRUN(1, 2, 0) OCCRUN(1, 3, 0) OCCRUN(1, 4, 0) OCCRUN(1, 6, 0) OCCRUN(1, 8, 0)
        OCCRUN(0, 4, 2) OCCRUN(0, 4, 4) OCCRUN(0, 4, 6) OCCRUN(0, 4, 8) OCCRUN(0, 8, 4) OCCRUN(0, 8, 6)
        OCCRUN(1, 4, 4) OCCRUN(1, 4, 6)
    }
    {
        const size_t n16 = (size_t)1280 * 1024 * F / 16;
        for (int blocks : {2048, 8192, 32768}) {
            float ms = timeit([&] { hipLaunchKernelGGL(rd_flat, dim3(blocks), dim3(256), 0, 0, (const uint4 *)slab, n16, out); }, 5);
            printf("{\"pattern\": \"flat read\", \"blocks\": %d, \"GB\": %.2f, \"ms\": %.4f, \"TBps\": %.3f}\n", blocks, n16 * 16 / 1e9, ms, n16 * 16 / ms / 1e9);
            ms = timeit([&] { hipLaunchKernelGGL(copy_flat, dim3(blocks), dim3(256), 0, 0, (const uint4 *)slab, (uint4 *)diff, n16); }, 5);
            printf("{\"pattern\": \"flat copy\", \"blocks\": %d, \"GB_read\": %.2f, \"ms\": %.4f, \"TBps_read_plus_write\": %.3f}\n", blocks, n16 * 16 / 1e9, ms,
                   2 * n16 * 16 / ms / 1e9);
        }
    }
    if (argc > 3) return 0;
    run<1280, 0>(slab, sg, out, diff, F, 1024, R);
    run<1280, 1>(slab, sg, out, diff, F, 1024, R);
    run<1280, 2>(slab, sg, out, diff, F, 1024, R);
    run<1680, 0>(slab, sg, out, diff, F, 1050, R);
    run<1680, 1>(slab, sg, out, diff, F, 1050, R);
    run<1680, 2>(slab, sg, out, diff, F, 1050, R);
    return 0;
}
