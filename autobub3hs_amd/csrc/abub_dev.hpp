// abub_dev.hpp -- shared device helpers, small structs and host-side plumbing of the gfx950 kernels
// (abub_k2.hip: ProcessFrame + histogram; abub_k3.hip: post-trigger images; abub_misc.hip: training, compaction,
// bellows terms, error plumbing, per-stream scratch).  Everything here is static / inline: each TU gets its own copy.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <map>
#include <mutex>
#include <utility>

#include "../../include/abub_hip.h"

// ------------------------------------------------------------------------------------------------
// error plumbing (the thread-local message buffer lives in abub_misc.hip)
// ------------------------------------------------------------------------------------------------
int abub_set_err_(int code, const char *what, hipError_t e);
static int set_err(int code, const char *what, hipError_t e = hipSuccess) { return abub_set_err_(code, what, e); }
#define HIPCHK(x)                                   \
    do {                                            \
        hipError_t e_ = (x);                        \
        if (e_ != hipSuccess)                       \
            return set_err(ABUB_E_HIP, #x, e_);     \
    } while (0)

// ------------------------------------------------------------------------------------------------
// small device helpers
// ------------------------------------------------------------------------------------------------
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
typedef short s16x2 __attribute__((ext_vector_type(2)));

// cv::borderInterpolate(BORDER_REFLECT_101)
__device__ __forceinline__ int reflect101(int p, int len)
{
    if (len == 1)
        return 0;
    while (p < 0 || p >= len)
        p = p < 0 ? -p : 2 * len - 2 - p;
    return p;
}

// two u16 lanes, saturating unsigned subtract  (v_pk_sub_u16 ... clamp)
__device__ __forceinline__ uint32_t pk_subsat(uint32_t a, uint32_t b)
{
    u16x2 x = __builtin_bit_cast(u16x2, a), y = __builtin_bit_cast(u16x2, b);
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_sub_sat(x, y));
}
// two i16 lanes, |a-b|  (2x v_pk_sub_i16 + v_pk_max_i16)
__device__ __forceinline__ uint32_t pk_absdiff(uint32_t a, uint32_t b)
{
    s16x2 x = __builtin_bit_cast(s16x2, a), y = __builtin_bit_cast(s16x2, b);
    s16x2 d = x - y, e = y - x;
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(d, e));
}
// two u16 lanes: a*K + b  (v_pk_mad_u16 with an inline constant)
template <int K>
__device__ __forceinline__ uint32_t pk_madk(uint32_t a, uint32_t b)
{
    u16x2 x = __builtin_bit_cast(u16x2, a), y = __builtin_bit_cast(u16x2, b);
    u16x2 k = {K, K};
    return __builtin_bit_cast(uint32_t, (u16x2)(x * k + y));
}

// (a << 2) + b in one VALU op.  Written as asm because the compiler would CSE the shift of two such
// expressions sharing `a` into shift + 2 adds (3 ops instead of 2).  No u16 lane overflows into its
// neighbour here (all lanes <= 65408 after the add), so the 32-bit form is exact for both lanes.
__device__ __forceinline__ uint32_t lshl2_add(uint32_t a, uint32_t b)
{
    uint32_t r;
    asm("v_lshl_add_u32 %0, %1, 2, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// bytes (b0,b1) / (b2,b3) of a dword widened to two u16 lanes  (v_perm_b32)
__device__ __forceinline__ uint32_t widen_lo(uint32_t w) { return __builtin_amdgcn_perm(0u, w, 0x0c010c00u); }
__device__ __forceinline__ uint32_t widen_hi(uint32_t w) { return __builtin_amdgcn_perm(0u, w, 0x0c030c02u); }

// hist[slot][0] = P - sum(hist[slot][1..255])  (the kernels only count non-zero pixels)
// ---- packed-byte thresholds for the v_sad_u8 bound scans (K2: k2_sad_chain, K3: k3_bound_scan) ------------------------
__device__ __forceinline__ uint32_t pk_addsat(uint32_t a, uint32_t b)
{
    u16x2 x = __builtin_bit_cast(u16x2, a), y = __builtin_bit_cast(u16x2, b);
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_add_sat(x, y));
}
// bytes (b0,b1) / (b2,b3) of a dword into the HIGH bytes of two u16 lanes: a saturating u16 add / sub of two such
// values leaves min(x + y, 255) / max(x - y, 0) in the high byte
__device__ __forceinline__ uint32_t widen8_lo(uint32_t w) { return __builtin_amdgcn_perm(0u, w, 0x010c000cu); }
__device__ __forceinline__ uint32_t widen8_hi(uint32_t w) { return __builtin_amdgcn_perm(0u, w, 0x030c020cu); }
// the high bytes of the lanes of (a: pixels 0,1; b: pixels 2,3) packed back into one dword
__device__ __forceinline__ uint32_t pack8(uint32_t a, uint32_t b) { return __builtin_amdgcn_perm(b, a, 0x07050301u); }


static __global__ __launch_bounds__(64) void k_hist_bin0(uint32_t *hist, uint32_t P, const uint8_t *__restrict__ only = nullptr)
{
    if (only && !only[blockIdx.x]) // (deferred pieces: only the slots whose rows were just completed)
        return;
    uint32_t *h = hist + (size_t)blockIdx.x * 256;
    int l = threadIdx.x;
    uint32_t s = h[l + 64] + h[l + 128] + h[l + 192] + (l ? h[l] : 0u);
    for (int o = 32; o > 0; o >>= 1)
        s += __shfl_xor(s, o);
    if (l == 0)
        h[0] = P - s;
}

#define DPP_WAVE_SHL1 0x130 /* lane i <- lane i+1 */
#define DPP_WAVE_SHR1 0x138 /* lane i <- lane i-1 */

template <int NDW>
struct RowIn {
    uint32_t c[NDW], r[NDW], s[NDW];
};

template <int NDW>
__device__ __forceinline__ void k2_load_row(RowIn<NDW> &R, const uint8_t *__restrict__ cur,
                                            const uint8_t *__restrict__ ref,
                                            const uint8_t *__restrict__ sg, int y, int W, int xoff)
{
    // xoff is clamped to a valid column for idle lanes by the caller: no branch, no exec masking
    size_t o = (size_t)y * W + xoff;
    const uint32_t *pc = reinterpret_cast<const uint32_t *>(cur + o);
    const uint32_t *pr = reinterpret_cast<const uint32_t *>(ref + o);
    const uint32_t *ps = reinterpret_cast<const uint32_t *>(sg + o);
#pragma unroll
    for (int d = 0; d < NDW; d++) {
        R.c[d] = pc[d];
        R.r[d] = pr[d];
        R.s[d] = ps[d];
    }
}

// Optional fused compaction: pixels with value > thr are appended to one shared list as
// (slot | value << 24, raster index).  Lives entirely in the rare non-zero path.
struct Compact {
    uint32_t *pairs;
    uint32_t *count;
    uint32_t cap;
    uint32_t slot;
    int thr;
};
// One atomicAdd per wave and row: every lane brings its candidate count `c`, gets back the position of
// its first entry.  Must be called by all 64 lanes (wave-uniform control flow).  A single shared
// counter serialises at ~90 atomics/us on MI355X, so per-pixel reservations would dominate the pass.
__device__ __forceinline__ uint32_t compact_reserve(const Compact &cp, uint32_t c)
{
    const int lane = threadIdx.x;
    uint32_t inc = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        uint32_t t = __shfl_up(inc, o);
        if (lane >= o)
            inc += t;
    }
    const uint32_t total = __shfl(inc, 63);
    uint32_t base = 0;
    if (total) {
        if (lane == 0)
            base = atomicAdd(cp.count, total);
        base = __shfl(base, 0);
    }
    return base + inc - c;
}
__device__ __forceinline__ void compact_put(const Compact &cp, uint32_t &pos, uint32_t v, uint32_t idx)
{
    if ((int)v > cp.thr) {
        if (pos < cp.cap) {
            cp.pairs[2 * (size_t)pos] = cp.slot | (v << 24);
            cp.pairs[2 * (size_t)pos + 1] = idx;
        }
        ++pos;
    }
}

#define K2B_PEND 512 /* suspects per (job, chunk) kept in LDS; also >= the groups of one row (W/4 <= 512) */
#ifndef K2B_SUB
#define K2B_SUB 32 /* rows per handed-over piece */
#endif
#ifndef K2B_GS
#define K2B_GS 2 /* 4-pixel groups per recurrence group of the bound scans */
#endif
#define G_TW 32 /* tile of the generic (any size / ROI) kernels */
#define G_TH 8

// hand the rows [y, y1) of `unit` to the row machine, in pieces of K2B_SUB rows (capacity: see launch_k2_rows)
__device__ __forceinline__ void k2b_hand_over(uint2 *__restrict__ units, uint32_t *__restrict__ nunits, uint32_t unit, int y, int y1,
                                              int lane, uint8_t *__restrict__ incomplete = nullptr, uint32_t job = 0)
{
    const int np = (y1 - y + K2B_SUB - 1) / K2B_SUB;
    if (np <= 0)
        return;
    uint32_t base = 0;
    if (lane == 0) {
        base = atomicAdd(nunits, (uint32_t)np);
        if (incomplete) // deferred pieces (abub_diff_hist_chained_deferred_dev): the job's histogram is not final yet
            incomplete[job] = 1;
    }
    base = __builtin_amdgcn_readfirstlane(base);
    for (int i = lane; i < np; i += 64) {
        const int a = y + i * K2B_SUB, b = a + K2B_SUB < y1 ? a + K2B_SUB : y1;
        units[base + i] = make_uint2(unit, (uint32_t)a | ((uint32_t)b << 16));
    }
}

// ---- the launch's global suspect list (K2 and K3 scans) -----------------------------------------------------------
// The scanning waves only MOVE their LDS suspect lists to a global list {job, group code}; sus_tail_list evaluates it
// afterwards with the whole chip (see there for why).  SusList.list == nullptr: no global list, the waves evaluate
// their suspects themselves.
struct SusList {
    uint2 *list;
    uint32_t *count;
    uint32_t cap;
};
#define SUSL_UB 4 /* entries per lane and block iteration of sus_tail_list, at most */
__device__ __forceinline__ void sus_copy_out(const uint32_t *pend, uint32_t n, uint32_t job, uint2 *__restrict__ glist, uint32_t base,
                                             int lane)
{
    for (uint32_t i = lane; i < n; i += 64)
        glist[base + i] = make_uint2(job, pend[i]);
}
// reserves n entries; false (and the slots it did get are marked empty) when the list cannot take them
__device__ __forceinline__ bool sus_reserve(uint32_t n, uint2 *__restrict__ glist, uint32_t *__restrict__ gcount, uint32_t gcap,
                                            uint32_t &base, int lane)
{
    if (!glist)
        return false;
    uint32_t b = 0;
    if (lane == 0)
        b = atomicAdd(gcount, n);
    b = (uint32_t)__builtin_amdgcn_readfirstlane((int)b);
    base = b;
    if (b < gcap && gcap - b >= n)
        return true;
    for (uint32_t i = b + lane; i < gcap && i - b < n; i += 64) // (b may already lie beyond the capacity)
        glist[i] = make_uint2(0xffffffffu, 0u);
    return false;
}

// wave-level ordering of LDS traffic (a list belongs to one wave; LDS executes a wave's instructions in order)
__device__ __forceinline__ void wave_lds_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// D for the four pixels of 4-pixel group `code` (= y * W/4 + x0/4) of a job, straight from the definition
// (AnalyzerUnit.cpp:351-370): returns the values packed one per byte.  Interior groups read their 12-byte window as
// three aligned dwords per array and row; the two edge groups (reflected columns) take the byte path.
__device__ __forceinline__ uint32_t k2_exact_group(const uint8_t *__restrict__ cur, const uint8_t *__restrict__ ref,
                                                   const uint8_t *__restrict__ sg, int y, int x0, int W, int H)
{
    const bool interior = x0 >= 4 && x0 + 8 <= W;
    int xs[8];
#pragma unroll
    for (int j = 0; j < 8; j++)
        xs[j] = reflect101(x0 - 2 + j, W);
    int Sp[4] = {0, 0, 0, 0}, Sn[4] = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 5; i++) {
        const size_t ro = (size_t)reflect101(y - 2 + i, H) * W;
        int pp[8], nn[8];
        if (interior) {
            const uint32_t *pc = reinterpret_cast<const uint32_t *>(cur + ro + x0 - 4);
            const uint32_t *pr = reinterpret_cast<const uint32_t *>(ref + ro + x0 - 4);
            const uint32_t *ps = reinterpret_cast<const uint32_t *>(sg + ro + x0 - 4);
            const uint32_t cw[3] = {pc[0], pc[1], pc[2]}, rw[3] = {pr[0], pr[1], pr[2]}, sw[3] = {ps[0], ps[1], ps[2]};
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int q = (2 + j) >> 2, sh = 8 * ((2 + j) & 3);
                const int c = (cw[q] >> sh) & 0xff, r = (rw[q] >> sh) & 0xff, s6 = (sw[q] >> sh) & 0xff;
                int a = c - r - s6, b = r - c - s6;
                pp[j] = a > 0 ? a : 0;
                nn[j] = b > 0 ? b : 0;
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int c = cur[ro + xs[j]], r = ref[ro + xs[j]], s6 = sg[ro + xs[j]];
                int a = c - r - s6, b = r - c - s6;
                pp[j] = a > 0 ? a : 0;
                nn[j] = b > 0 ? b : 0;
            }
        }
        const int wv = (i == 0 || i == 4) ? 1 : (i == 2 ? 6 : 4);
#pragma unroll
        for (int k = 0; k < 4; k++) {
            Sp[k] += wv * (pp[k] + 4 * pp[k + 1] + 6 * pp[k + 2] + 4 * pp[k + 3] + pp[k + 4]);
            Sn[k] += wv * (nn[k] + 4 * nn[k + 1] + 6 * nn[k + 2] + 4 * nn[k + 3] + nn[k + 4]);
        }
    }
    uint32_t packed = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int a = (Sp[k] + 128) >> 8, b = (Sn[k] + 128) >> 8;
        const int d = a > b ? a - b : b - a;
        packed |= (uint32_t)d << (8 * k);
    }
    return packed;
}

// exact O for the four pixels of 4-pixel group (y, x0) (L3Localizer.cpp:779-785), packed one value per byte.
// Interior groups read their 6-pixel window as three aligned dwords per array and row (the pixels x0-1 .. x0+4 are
// bytes 3 .. 8 of the 12 bytes from x0-4); the two edge groups (reflected columns) go byte by byte.
__device__ __forceinline__ uint32_t k3_exact_group(const uint8_t *__restrict__ f, const uint8_t *__restrict__ m,
                                                   const uint8_t *__restrict__ sg, int y, int x0, int W, int H)
{
    const bool interior = x0 >= 4 && x0 + 8 <= W;
    int S[4] = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const size_t ro = (size_t)reflect101(y - 1 + i, H) * W;
        int o[6];
        if (interior) {
            const uint32_t *pf = reinterpret_cast<const uint32_t *>(f + ro + x0 - 4);
            const uint32_t *pm = reinterpret_cast<const uint32_t *>(m + ro + x0 - 4);
            const uint32_t *ps = reinterpret_cast<const uint32_t *>(sg + ro + x0 - 4);
            const uint32_t fw[3] = {pf[0], pf[1], pf[2]}, mw[3] = {pm[0], pm[1], pm[2]}, sw[3] = {ps[0], ps[1], ps[2]};
#pragma unroll
            for (int j = 0; j < 6; j++) {
                const int q = (3 + j) >> 2, sh = 8 * ((3 + j) & 3);
                int a = (int)((fw[q] >> sh) & 0xff) - (int)((mw[q] >> sh) & 0xff);
                a = a < 0 ? -a : a;
                a -= (int)((sw[q] >> sh) & 0xff);
                o[j] = a < 0 ? 0 : a;
            }
        } else {
#pragma unroll
            for (int j = 0; j < 6; j++) {
                const int x = reflect101(x0 - 1 + j, W);
                int a = (int)f[ro + x] - (int)m[ro + x];
                a = a < 0 ? -a : a;
                a -= (int)sg[ro + x];
                o[j] = a < 0 ? 0 : a;
            }
        }
#pragma unroll
        for (int q = 0; q < 4; q++)
            S[q] += o[q] + o[q + 1] + o[q + 2];
    }
    uint32_t packed = 0;
#pragma unroll
    for (int q = 0; q < 4; q++)
        packed |= ((uint32_t)(S[q] + 4) / 9u) << (8 * q);
    return packed;
}

// ---- sus_tail_list: the second kernel of a bound-and-verify launch ------------------------------------------------
// The exact evaluation of a suspect group is a chain of dependent latencies (LDS code -> window loads -> values ->
// candidate-list reservation -> stores): done by the scanning wave itself it holds a wave slot for about as long as
// the scan did -- every tracking frame has its bubble -- and with the fused candidate list it costs one reservation
// on the shared counter per 64 groups (that counter serialises at ~90 atomics / us).  So the scanning waves only MOVE
// their LDS lists to the global list (one reservation per wave, or per overflowing LDS list), and this kernel
// evaluates all of it with the whole chip: one lane per group, up to four consecutive groups per lane, one
// candidate-list reservation per block iteration.  KIND 2: D of ProcessFrame (mu unused), KIND 3: O of the tracking
// frames.  When the global list is full the scanning wave evaluates its groups itself (k2b_tail / k3s_tail).
template <int KIND, bool COMPACT, bool STORE>
__global__ __launch_bounds__(256) void sus_tail_list(const uint8_t *__restrict__ frames, const uint8_t *__restrict__ mu,
                                                     const uint8_t *__restrict__ sigma6, const abub_job *__restrict__ jobs,
                                                     int W, int H, uint32_t *__restrict__ hist, uint8_t *__restrict__ img,
                                                     const uint2 *__restrict__ glist, const uint32_t *__restrict__ gcount,
                                                     uint32_t gcap, const int32_t *__restrict__ cthr, uint32_t *pairs,
                                                     uint32_t pcap, uint32_t *pcount, uint32_t slot_base)
{
    __shared__ uint32_t wtot[4], bbase;
    // histogram counts of the wave's leading job are gathered in LDS and added to memory once per bin: a bubble's
    // pixels share a few values, and thousands of same-address atomics per frame serialise in the L2
    __shared__ uint32_t lh[4][256];
    uint32_t n = *gcount;
    if (n > gcap)
        n = gcap;
    const size_t P = (size_t)W * H;
    const uint32_t ngroups = (uint32_t)W / 4;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // entries per lane and block iteration: as few as keeps every block busy (short lists: more blocks in flight, shorter
    // latency chains), at most SUSL_UB (long lists: few reservations on the shared candidate list)
    uint32_t KE = (n + gridDim.x * 256u - 1u) / (gridDim.x * 256u);
    KE = KE < 1u ? 1u : (KE > SUSL_UB ? (uint32_t)SUSL_UB : KE);
    const uint32_t PER = 256u * KE;
#pragma unroll
    for (int q = 0; q < 4; q++)
        lh[wv][4 * lane + q] = 0;
    for (uint32_t e0 = blockIdx.x * PER; e0 < n; e0 += gridDim.x * PER) {
        uint32_t Ov[SUSL_UB], pix0[SUSL_UB], slot[SUSL_UB];
        int thr[SUSL_UB];
        uint32_t c = 0;
        // the wave's leading job = the job of its first entry (most of the wave's consecutive entries belong to it)
        uint32_t job0 = 0xffffffffu;
        {
            const uint32_t ef = e0 + (uint32_t)(wv * 64) * KE;
            if (ef < n)
                job0 = glist[ef].x;
            job0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)job0);
        }
        // entries and their job records first (independent loads), then the windows
        uint2 ens[SUSL_UB];
        abub_job jbs[SUSL_UB];
#pragma unroll
        for (int u = 0; u < SUSL_UB; u++) {
            // consecutive entries per lane: the candidate list keeps the producers' runs of equal slots (k_pairs_scatter
            // reserves once per run)
            const uint32_t e = e0 + (uint32_t)tid * KE + (uint32_t)u;
            ens[u] = make_uint2(0xffffffffu, 0u);
            if ((uint32_t)u < KE && e < n)
                ens[u] = glist[e];
        }
#pragma unroll
        for (int u = 0; u < SUSL_UB; u++)
            jbs[u] = jobs[ens[u].x != 0xffffffffu ? ens[u].x : 0u];
#pragma unroll
        for (int u = 0; u < SUSL_UB; u++) {
            Ov[u] = 0;
            pix0[u] = 0;
            slot[u] = 0;
            thr[u] = 255;
            const uint2 en = ens[u];
            if (en.x != 0xffffffffu) {
                const abub_job jb = jbs[u];
                const int y = (int)(en.y / ngroups), x0 = (int)(en.y % ngroups) * 4;
                uint32_t packed;
                if (KIND == 2)
                    packed = k2_exact_group(frames + (size_t)jb.cur * P, frames + (size_t)jb.ref * P,
                                            sigma6 + (size_t)jb.model * P, y, x0, W, H);
                else
                    packed = k3_exact_group(frames + (size_t)jb.cur * P, mu + (size_t)jb.model * P,
                                            sigma6 + (size_t)jb.model * P, y, x0, W, H);
                if (COMPACT)
                    thr[u] = cthr[jb.out];
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const uint32_t v = (packed >> (8 * q)) & 0xffu;
                    if (v) {
                        if (en.x == job0)
                            atomicAdd(&lh[wv][v], 1u);
                        else
                            atomicAdd(&hist[(size_t)jb.out * 256 + v], 1u);
                    }
                    c += COMPACT && (int)v > thr[u];
                }
                if (STORE) // (the launcher cleared the image; an aligned dword)
                    *reinterpret_cast<uint32_t *>(img + (size_t)jb.out * P + (size_t)y * W + x0) = packed;
                Ov[u] = packed;
                pix0[u] = (uint32_t)(y * W + x0);
                slot[u] = jb.out + slot_base;
            }
        }
        if (job0 != 0xffffffffu) { // the gathered counts: four bins per lane
            wave_lds_fence();
            const size_t hb = (size_t)jobs[job0].out * 256;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const uint32_t cnt = lh[wv][4 * lane + q];
                if (cnt) {
                    atomicAdd(&hist[hb + 4 * lane + q], cnt);
                    lh[wv][4 * lane + q] = 0;
                }
            }
            wave_lds_fence();
        }
        if (COMPACT) { // one reservation on the shared candidate list per block iteration
            uint32_t inc = c;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t t = __shfl_up(inc, o);
                if (lane >= o)
                    inc += t;
            }
            if (lane == 63)
                wtot[wv] = inc;
            __syncthreads();
            if (tid == 0) {
                const uint32_t tot = wtot[0] + wtot[1] + wtot[2] + wtot[3];
                bbase = tot ? atomicAdd(pcount, tot) : 0u;
            }
            __syncthreads();
            uint32_t pos = bbase + inc - c;
            for (int w = 0; w < wv; w++)
                pos += wtot[w];
#pragma unroll
            for (int u = 0; u < SUSL_UB; u++)
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const uint32_t v = (Ov[u] >> (8 * q)) & 0xffu;
                    if ((int)v > thr[u]) {
                        if (pos < pcap) {
                            pairs[2 * (size_t)pos] = slot[u] | (v << 24);
                            pairs[2 * (size_t)pos + 1] = pix0[u] + q;
                        }
                        ++pos;
                    }
                }
            __syncthreads(); // wtot / bbase are rewritten by the next iteration
        }
    }
}

static inline int pick_ndw(int W)
{
    if (W < 4 || (W & 3))
        return 0;
    int nd = W / 4;
    for (int ndw = 1; ndw <= 8; ndw++)
        if (nd % ndw == 0 && nd / ndw <= 64)
            return ndw;
    return 0;
}

// Per-stream scratch of the bound-and-verify passes (abub_misc.hip): returns the stream's grow-only buffer with the
// scratch mutex HELD by `hold` until the caller has enqueued its whole launch sequence.
void *abub_k2_scratch(hipStream_t st, size_t bytes, std::unique_lock<std::mutex> &hold);
static inline void *k2_scratch(hipStream_t st, size_t bytes, std::unique_lock<std::mutex> &hold) { return abub_k2_scratch(st, bytes, hold); }

struct CompactArgs {
    const int32_t *cthr;
    uint32_t *pairs;
    uint32_t cap;
    uint32_t *count;
    uint32_t slot_base; // added to job.out in the list entries (several launches share one list)
    int chain_len = 0, chain_stride = 0; // trigger-only hint: blocks of chain_len jobs, job q refs the cur of job q - stride
    // deferred pieces (trigger-only): the handed-over row ranges go to the CALLER's list instead of the row machine, and
    // incomplete[job] is set for every job that has some; abub_diff_hist_pieces_dev() evaluates them later, per job
    uint2 *df_pieces = nullptr;
    uint32_t df_cap = 0;
    uint32_t *df_count = nullptr;
    uint8_t *df_incomplete = nullptr;
};
