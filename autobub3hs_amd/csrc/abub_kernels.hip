// abub_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the bubble-detection hot path
// and their stateless C-ABI launchers (include/abub_hip.h, layer A).
//
// Kernel inventory (DESIGN.md section "Kernels"):
//   K2  k2_rows<NDW,STORE,PF,COMPACT>  fused AnalyzerUnit::ProcessFrame + 256-bin histogram, register-rolling rows
//                                      (optional stored image / fused candidate list; list mode for handed-over rows)
//       k2_bound_chain / k2_bound_scan trigger-only: proves rows of D zero from a bound, lists the rest
//       k2_exact_groups                the listed 4-pixel groups, straight from the definition
//   K2g k2_generic                     same arithmetic, any size / ROI, LDS tile (also the ROI overload)
//   K1  k1_welford, k1_welford4        Trainer::CalculateMeanSigmaImageVector (float32 Welford, no FMA)
//   K1b k1b_pair_hist                  histogram of sat(f1 - f0) (training entropy veto)
//   K3  k3_rows<NDW,STORE>, k3_generic post-trigger |f-mu|-6sigma, 3x3 box, histogram (+ candidate list)
//   K4  k4_compact, k4_compact_pairs   binarize + foreground index compaction (unfused fallback)
//       k_pairs_*                      counting-sort grouping of the shared candidate list by image
//       k_match_ccorr, k_subsat_hist   bellows veto (TrackAFeature terms, image subtraction)
// No MFMA anywhere: this is integer/byte pixel work (see DESIGN.md "Why no MFMA").
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
// error plumbing
// ------------------------------------------------------------------------------------------------
static thread_local char g_err[256] = "";
int abub_set_err_(int code, const char *what, hipError_t e);

extern "C" const char *abub_last_error(void) { return g_err; }

int abub_set_err_(int code, const char *what, hipError_t e)
{
    if (e != hipSuccess)
        snprintf(g_err, sizeof g_err, "%s: %s", what, hipGetErrorString(e));
    else
        snprintf(g_err, sizeof g_err, "%s", what);
    return code;
}
static int set_err(int code, const char *what, hipError_t e = hipSuccess) { return abub_set_err_(code, what, e); }
#define HIPCHK(x)                                   \
    do {                                            \
        hipError_t e_ = (x);                        \
        if (e_ != hipSuccess)                       \
            return set_err(ABUB_E_HIP, #x, e_);     \
    } while (0)

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

// hist[slot][0] = P - sum(hist[slot][1..255])  (the kernels only count non-zero pixels)
__global__ __launch_bounds__(64) void k_hist_bin0(uint32_t *hist, uint32_t P)
{
    uint32_t *h = hist + (size_t)blockIdx.x * 256;
    int l = threadIdx.x;
    uint32_t s = h[l + 64] + h[l + 128] + h[l + 192] + (l ? h[l] : 0u);
    for (int o = 32; o > 0; o >>= 1)
        s += __shfl_xor(s, o);
    if (l == 0)
        h[0] = P - s;
}

// ------------------------------------------------------------------------------------------------
// K2 fast: register-rolling rows.
//
// One wave (64-thread workgroup) owns output rows [y0,y1) of one job.  Lane L holds 4*NDW consecutive
// pixels of a row (blocked mapping), the wave spans the whole row: W == 4*NDW*nl, nl <= 64 lanes.
// Per input row: 3*NDW dword loads/lane (cur, ref, sigma6) -> saturating differences on u16 pairs
// (two pixels per 32-bit register, pos plane and neg plane) -> horizontal 1-4-6-4-1 with the two
// neighbour pairs fetched from the adjacent lanes by DPP wave shifts (reflect-101 in-lane at the
// image edges) -> vertical 1-4-6-4-1 as four in-place accumulators per pair (no ring rotation)
// -> (S+128)>>8 via byte permute, |pos-neg|, LDS histogram of the (rare) non-zero pixels, optional store.
// Every u16 lane stays < 65536: H <= 4080+8, V <= 16*4088 = 65408.
// ------------------------------------------------------------------------------------------------
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

// persistent per-wave state of the vertical pass: four in-place accumulators per u16 pair and plane
// out = a0 + X ; a0 = a1 + 4X ; a1 = a2 + 6X ; a2 = Xprev + 4X.  The previous row's X lives in a second
// register set that alternates with the current one (K2Row), so nothing is copied at the loop back-edge.
template <int NDW>
struct K2Acc {
    uint32_t pa0[2 * NDW], pa1[2 * NDW], pa2[2 * NDW];
    uint32_t na0[2 * NDW], na1[2 * NDW], na2[2 * NDW];
};
template <int NDW>
struct K2Row {
    uint32_t hp[2 * NDW], hn[2 * NDW]; // horizontally filtered row, both planes
};

// pos / neg planes of one row as u16 pairs (AnalyzerUnit.cpp:351-352)
template <int NDW>
__device__ __forceinline__ void k2_planes(const RowIn<NDW> &in, uint32_t (&Xp)[2 * NDW], uint32_t (&Xn)[2 * NDW])
{
#pragma unroll
    for (int d = 0; d < NDW; d++) {
        uint32_t c0 = widen_lo(in.c[d]), c1 = widen_hi(in.c[d]);
        uint32_t r0 = widen_lo(in.r[d]), r1 = widen_hi(in.r[d]);
        uint32_t s0 = widen_lo(in.s[d]), s1 = widen_hi(in.s[d]);
        Xp[2 * d] = pk_subsat(c0, r0 + s0);
        Xn[2 * d] = pk_subsat(r0, c0 + s0);
        Xp[2 * d + 1] = pk_subsat(c1, r1 + s1);
        Xn[2 * d + 1] = pk_subsat(r1, c1 + s1);
    }
}
// horizontal 1-4-6-4-1 of one row, both planes (AnalyzerUnit.cpp:359-360; the +128 rounding is applied at the end)
template <int NDW>
__device__ __forceinline__ void k2_hpass(const uint32_t (&Xp)[2 * NDW], const uint32_t (&Xn)[2 * NDW], bool first_lane,
                                         bool last_lane, uint32_t (&Hp)[2 * NDW], uint32_t (&Hn)[2 * NDW])
{
    constexpr int NP = 2 * NDW;
    // left pair (p[-2],p[-1]) and right pair (p[n],p[n+1]): neighbour lanes, reflect-101 at the edges
    uint32_t reflLp = __builtin_amdgcn_perm(Xp[0], Xp[1], 0x07060100u); // (X1.lo, X0.hi) = (p2,p1)
    uint32_t reflLn = __builtin_amdgcn_perm(Xn[0], Xn[1], 0x07060100u);
    uint32_t reflRp = __builtin_amdgcn_perm(Xp[NP - 2], Xp[NP - 1], 0x07060100u); // (p[n-2],p[n-3])
    uint32_t reflRn = __builtin_amdgcn_perm(Xn[NP - 2], Xn[NP - 1], 0x07060100u);
    uint32_t Lp = __builtin_amdgcn_update_dpp(0u, Xp[NP - 1], DPP_WAVE_SHR1, 0xf, 0xf, false);
    uint32_t Ln = __builtin_amdgcn_update_dpp(0u, Xn[NP - 1], DPP_WAVE_SHR1, 0xf, 0xf, false);
    uint32_t Rp = __builtin_amdgcn_update_dpp(0u, Xp[0], DPP_WAVE_SHL1, 0xf, 0xf, false);
    uint32_t Rn = __builtin_amdgcn_update_dpp(0u, Xn[0], DPP_WAVE_SHL1, 0xf, 0xf, false);
    Lp = first_lane ? reflLp : Lp;
    Ln = first_lane ? reflLn : Ln;
    Rp = last_lane ? reflRp : Rp;
    Rn = last_lane ? reflRn : Rn;
    uint32_t am1p = __builtin_amdgcn_alignbit(Xp[0], Lp, 16); // (p[-1], p[0])
    uint32_t am1n = __builtin_amdgcn_alignbit(Xn[0], Ln, 16);
#pragma unroll
    for (int j = 0; j < NP; j++) {
        uint32_t xm1p = j ? Xp[j - 1] : Lp, xp1p = j + 1 < NP ? Xp[j + 1] : Rp;
        uint32_t xm1n = j ? Xn[j - 1] : Ln, xp1n = j + 1 < NP ? Xn[j + 1] : Rn;
        uint32_t ap1p = __builtin_amdgcn_alignbit(xp1p, Xp[j], 16); // (p[2j+1], p[2j+2])
        uint32_t ap1n = __builtin_amdgcn_alignbit(xp1n, Xn[j], 16);
        uint32_t sp = am1p + ap1p, sn = am1n + ap1n;
        uint32_t tp = xm1p + xp1p, tn = xm1n + xp1n;
        Hp[j] = pk_madk<6>(Xp[j], (sp << 2) + tp); // every u16 lane <= 4080
        Hn[j] = pk_madk<6>(Xn[j], (sn << 2) + tn);
        am1p = ap1p;
        am1n = ap1n;
    }
}

// one input row -> one output row (valid once 5 rows went in)
template <int NDW, bool STORE, bool COMPACT>
__device__ __forceinline__ void k2_row(const RowIn<NDW> &in, K2Acc<NDW> &A, K2Row<NDW> &Hcur,
                                       const K2Row<NDW> &Hprev, bool emit, bool active,
                                       bool first_lane, bool last_lane, uint32_t *lh,
                                       uint32_t *__restrict__ po, const Compact &cp, uint32_t pix0, int &zrun)
{
    constexpr int NP = 2 * NDW;
    // sat(sat(c - r) - s) == sat(c - (r + s)) for s >= 0, and r + s <= 510 fits the u16 lane: one plain
    // 32-bit add (VOP2) + one saturating packed subtract per plane instead of two packed subtracts
    uint32_t Xp[NP], Xn[NP];
    k2_planes<NDW>(in, Xp, Xn);

    // ---- zero-run shortcut (wave-uniform) ------------------------------------------------------
    // A row whose pos and neg planes vanish on every lane filters to H = 0, and after four such rows the
    // whole vertical state (a0,a1,a2 and both H sets) is zero: further zero rows leave it untouched and emit
    // D = 0, so everything below is skipped.  Static-camera frames spend most of their rows here.
    {
        uint32_t nz = 0;
#pragma unroll
        for (int j = 0; j < NP; j++)
            nz |= Xp[j] | Xn[j];
        const bool rowzero = __builtin_amdgcn_ballot_w64(nz != 0) == 0;
        if (rowzero && zrun >= 4) {
            if (STORE && emit && active) {
#pragma unroll
                for (int d = 0; d < NDW; d++)
                    po[d] = 0;
            }
            return;
        }
        zrun = rowzero ? zrun + 1 : 0;
    }

    // ---- horizontal 1-4-6-4-1 (AnalyzerUnit.cpp:359-360; the +128 rounding is applied at the end) ----
    uint32_t(&Hp)[NP] = Hcur.hp;
    uint32_t(&Hn)[NP] = Hcur.hn;
    k2_hpass<NDW>(Xp, Xn, first_lane, last_lane, Hp, Hn);

    // ---- vertical 1-4-6-4-1, (S+128)>>8, absdiff (AnalyzerUnit.cpp:370) -----------------------
    uint32_t Vp[NP], Vn[NP];
    uint32_t big = 0; // OR of all sums S: if no u16 lane reaches 128 every (S+128)>>8, hence D, is 0
#pragma unroll
    for (int j = 0; j < NP; j++) {
        Vp[j] = A.pa0[j] + Hp[j];
        Vn[j] = A.na0[j] + Hn[j];
        const uint32_t p4 = Hp[j] << 2, n4 = Hn[j] << 2; // shared by a0 and a2 (plain VOP2 shift + adds)
        A.pa0[j] = A.pa1[j] + p4;
        A.na0[j] = A.na1[j] + n4;
        A.pa1[j] = pk_madk<6>(Hp[j], A.pa2[j]);
        A.na1[j] = pk_madk<6>(Hn[j], A.na2[j]);
        A.pa2[j] = Hprev.hp[j] + p4;
        A.na2[j] = Hprev.hn[j] + n4;
        big |= Vp[j] | Vn[j];
    }
    uint32_t Dp[NP];
    uint32_t any = 0;
    // wave-uniform shortcut: for almost every row of almost every frame all sums stay below 128
    const bool quiet = __builtin_amdgcn_ballot_w64((big & 0xff80ff80u) != 0) == 0;
    if (quiet) {
#pragma unroll
        for (int j = 0; j < NP; j++)
            Dp[j] = 0;
    } else {
#pragma unroll
        for (int j = 0; j < NP; j++) {
            // (S+128)>>8: byte1 / byte3 of the u16 lanes after the rounding add (S <= 65280: no lane overflow)
            uint32_t rp = __builtin_amdgcn_perm(0u, Vp[j] + 0x00800080u, 0x0c030c01u);
            uint32_t rn = __builtin_amdgcn_perm(0u, Vn[j] + 0x00800080u, 0x0c030c01u);
            Dp[j] = pk_absdiff(rp, rn);
            any |= Dp[j];
        }
    }

    if (emit) {
        const bool mine = any && active;
        if (__builtin_amdgcn_ballot_w64(mine)) { // rare and wave-uniform: D is zero for almost every pixel
            uint32_t pos = 0;
            if (COMPACT) {
                uint32_t c = 0;
                if (mine) {
#pragma unroll
                    for (int j = 0; j < NP; j++)
                        c += ((int)(Dp[j] & 0xffffu) > cp.thr) + ((int)(Dp[j] >> 16) > cp.thr);
                }
                pos = compact_reserve(cp, c);
            }
            if (mine) {
#pragma unroll
                for (int j = 0; j < NP; j++) {
                    uint32_t lo = Dp[j] & 0xffffu, hi = Dp[j] >> 16;
                    if (lo) {
                        atomicAdd(&lh[lo], 1u);
                        if (COMPACT)
                            compact_put(cp, pos, lo, pix0 + 2 * j);
                    }
                    if (hi) {
                        atomicAdd(&lh[hi], 1u);
                        if (COMPACT)
                            compact_put(cp, pos, hi, pix0 + 2 * j + 1);
                    }
                }
            }
        }
        if (STORE && active) {
#pragma unroll
            for (int d = 0; d < NDW; d++)
                po[d] = __builtin_amdgcn_perm(Dp[2 * d + 1], Dp[2 * d], 0x06040200u);
        }
    }
}

#ifndef K2_WAVES_PER_EU
#define K2_WAVES_PER_EU 1
#endif
template <int NDW, bool STORE, int PF, bool COMPACT>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(K2_WAVES_PER_EU))) void k2_rows(const uint8_t *__restrict__ frames,
                                              const uint8_t *__restrict__ sigma6,
                                              const abub_job *__restrict__ jobs, int W, int H,
                                              int rows_per_chunk, int nchunks,
                                              uint32_t *__restrict__ hist, uint8_t *__restrict__ diff,
                                              const int32_t *__restrict__ cthr, uint32_t *pairs,
                                              uint32_t pcap, uint32_t *pcount, uint32_t slot_base,
                                              const uint2 *__restrict__ unit_list,
                                              const uint32_t *__restrict__ unit_count)
{
    constexpr int NP = 2 * NDW; // u16-pair registers per plane per lane
    __shared__ uint32_t lh[256];

    const int lane = threadIdx.x;
    // list mode (bound-and-verify hand-over): the grid strides over the listed units; otherwise unit = block
    const uint32_t nunits_ = unit_list ? *unit_count : gridDim.x;
    for (uint32_t ui = blockIdx.x; ui < nunits_; ui += gridDim.x) {
    const int unit = unit_list ? (int)unit_list[ui].x : (int)ui;
    const int job = unit / nchunks;
    const int chunk = unit - job * nchunks; // chunk fastest: unit % 8 == chunk % 8 when nchunks % 8 == 0
    const abub_job jb = jobs[job];
    const size_t P = (size_t)W * H;
    const uint8_t *cur = frames + (size_t)jb.cur * P;
    const uint8_t *ref = frames + (size_t)jb.ref * P;
    const uint8_t *sg = sigma6 + (size_t)jb.model * P;
    const int nl = W / (4 * NDW);
    const bool active = lane < nl;
    const bool first_lane = lane == 0;
    const bool last_lane = lane == nl - 1;
    const int xoff = active ? lane * 4 * NDW : 0; // idle lanes shadow lane 0 (results unused)

    // list mode: a handed-over piece, rows [y & 0xffff, y >> 16)
    const int y0 = unit_list ? (int)(unit_list[ui].y & 0xffffu) : chunk * rows_per_chunk;
    int y1 = unit_list ? (int)(unit_list[ui].y >> 16) : (chunk + 1) * rows_per_chunk;
    if (y1 > H)
        y1 = H;
    const int T = y1 - y0 + 4; // input rows y0-2 .. y1+1 (reflected)

    lh[lane] = 0;
    lh[lane + 64] = 0;
    lh[lane + 128] = 0;
    lh[lane + 192] = 0;
    __syncthreads();

    K2Acc<NDW> A;
#pragma unroll
    for (int j = 0; j < NP; j++)
        A.pa0[j] = A.pa1[j] = A.pa2[j] = A.na0[j] = A.na1[j] = A.na2[j] = 0;
    K2Row<NDW> HR[2];
#pragma unroll
    for (int j = 0; j < NP; j++)
        HR[0].hp[j] = HR[0].hn[j] = HR[1].hp[j] = HR[1].hn[j] = 0;

    uint8_t *dbase = STORE ? diff + (size_t)jb.out * P + xoff : nullptr;
    Compact cp;
    cp.pairs = COMPACT ? pairs : nullptr;
    cp.count = pcount;
    cp.cap = pcap;
    cp.slot = jb.out + slot_base;
    cp.thr = COMPACT ? cthr[jb.out] : 255;

    // Software prefetch PF rows ahead through a ring of PF+1 row buffers.  The loop is unrolled by
    // U = lcm(PF+1, 2) so that ring slots are compile-time registers and neither the ring nor the
    // vertical accumulators (ping-pong period 2) need register moves at the back-edge.
    constexpr int RING = PF + 1;
    constexpr int U = (RING % 2 == 0) ? RING : 2 * RING;
    RowIn<NDW> ring[RING];
#pragma unroll
    for (int k = 0; k < PF; k++) {
        int tk = k < T ? k : T - 1;
        k2_load_row<NDW>(ring[k], cur, ref, sg, reflect101(y0 - 2 + tk, H), W, xoff);
    }
    // T is rounded up to a multiple of U: the (at most U-1) extra rows re-read the last input row and emit
    // nothing, which keeps the unrolled body free of guards (no phi copies of the ring / accumulators)
    const int Tpad = (T + U - 1) / U * U;
    int zrun = 4; // the vertical state starts out all zero
    for (int t = 0; t < Tpad; t += U) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int tt = t + u;
            int tn = tt + PF < T ? tt + PF : T - 1;
            k2_load_row<NDW>(ring[(u + PF) % RING], cur, ref, sg, reflect101(y0 - 2 + tn, H), W, xoff);
            int y = y0 + tt - 4;
            k2_row<NDW, STORE, COMPACT>(ring[u % RING], A, HR[u & 1], HR[(u & 1) ^ 1], tt >= 4 && tt < T, active,
                                        first_lane, last_lane, lh,
                                        reinterpret_cast<uint32_t *>(dbase + (ptrdiff_t)y * W), cp,
                                        (uint32_t)(y * W + xoff), zrun);
        }
    }

    __syncthreads();
    uint32_t *gh = hist + (size_t)jb.out * 256;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        uint32_t v = lh[lane + 64 * k];
        if (v && (lane + 64 * k))
            atomicAdd(&gh[lane + 64 * k], v);
    }
    __syncthreads(); // lh is zeroed again by the next unit
    } // units
}

// ------------------------------------------------------------------------------------------------
// K2 "bound and verify".  D(y,x) != 0 needs a 5x5 weighted sum S >= 128 in one plane.  With X = pos + neg (disjoint
// supports, so both plane sums are <= the sum over X) and the lane's pixels in groups of four columns:
//     S(y,x) <= sum_i w_i * 6 * M_g(y+i),   M_g(r) = m_{g-1}(r) + m_g(r) + m_{g+1}(r),   m_g = mass of X in group g
// (every tap of the horizontal filter is <= 6 and reaches at most the neighbouring group; at the image edges the
// reflected columns fall into the edge group itself, which the edge-replicated m_{-1} = m_0 counts a second time).
//   k2_bound_scan / k2_bound_chain
//       one wave per (job, chunk) -- or per K chained jobs -- carries only this bound down the rows: the same
//       1-4-6-4-1 recurrence, but on packed group masses instead of 4*NDW filtered pairs in two planes, and proves
//       "D is zero" for whole rows with one ballot.  Groups it cannot prove are remembered in LDS (at most K2B_PEND
//       per job and chunk) and computed exactly BY THE SAME WAVE once its scan is over (k2b_tail: the four pixels
//       straight from the definition, one lane per group) -- no list in global memory, no second kernel.  A chunk
//       whose rows keep exceeding 32 suspects (one row of the row machine costs about 30 exact groups), or that
//       would overflow its LDS list, hands its REMAINING rows over.  (Handing over only the next K2B_SUB rows and
//       scanning on behind them was measured equal in time and cost a wave of occupancy at W = 1680.)
//   k2_rows (list mode)
//       the full row machine on the handed-over row ranges, cut into pieces of K2B_SUB rows so that the few of them
//       spread over the chip instead of serialising behind one wave each.
// Store mode: the scan writes the rows it is responsible for as zeros, the tail overwrites its groups' dwords (same
// wave, later in program order), the row machine writes the handed-over rows.
// Every pixel is either proven zero or computed with the reference arithmetic, never twice: histograms and D are
// bit-identical to the plain k2_rows pass.
// Packed halves: a register holds (mass of columns 0,2 | mass of columns 1,3) of a group; all recurrences are linear
// and stay < 65536 per half (<= 16 * 4 * 2 * 255 with paired groups); the row test folds max-of-halves over the
// lane's groups, which can only over-estimate, the per-group test on a suspicious row folds exactly.
// ------------------------------------------------------------------------------------------------
#define K2B_PEND 512 /* suspects per (job, chunk) kept in LDS; also >= the groups of one row (W/4 <= 512) */
#ifndef K2B_SUB
#define K2B_SUB 32 /* rows per handed-over piece */
#endif

// Lane -> pixel mapping of the scan kernels.  A row is cut into segments; in segment s every active lane owns segK(s)
// consecutive dwords (4-pixel groups): lane L the groups [gbase[s] + segK(s) * L, + segK(s)).
//   blocked (SPLIT = false): one segment of NDW dwords per lane -- the row machine's mapping; a lane's dwordx4 + dword
//                            loads then sit at a 4 * NDW byte stride, so every load instruction touches every line
//                            of the row partially;
//   split   (SPLIT = true):  two segments of 4 and NDW - 4 dwords per lane (NDW = 5 .. 7) over the same nl = W / (4 NDW)
//                            lanes: the first 16 * nl bytes of the row go out as one dwordx4 per lane, the rest as one
//                            dword / dwordx2 / dwordx3 per lane -- each load instruction covers ONE contiguous span of
//                            the row with whole pieces per lane.  tools/rowload_bench.cpp: the chained scan's access
//                            pattern is 3 % cheaper at W = 1280 and 12 % at W = 1680 this way.
// The suspect codes are global group indices and hand-overs are row ranges, so the row machine (always blocked) and
// the exact tails do not care which mapping the scan used.
template <int NDW, bool SPLIT>
struct ScanMap {
    static_assert(!SPLIT || (NDW >= 5 && NDW <= 7), "the split mapping is a dwordx4 plus 1 .. 3 dwords per lane");
    static constexpr int NSEG = SPLIT ? 2 : 1;
    static __device__ __host__ constexpr int segK(int s) { return !SPLIT ? NDW : (s == 0 ? 4 : NDW - 4); }
    static __device__ __host__ constexpr int segD0(int s) { return !SPLIT ? 0 : (s == 0 ? 0 : 4); }
    static __device__ __host__ constexpr int segOf(int d) { return !SPLIT ? 0 : (d < 4 ? 0 : 1); }
    int nl;          // active lanes (the same in every segment)
    int gbase[NSEG]; // first 4-pixel group of the segment
    __device__ __forceinline__ void init(int W)
    {
        nl = W / (4 * NDW);
#pragma unroll
        for (int s = 0; s < NSEG; s++)
            gbase[s] = nl * segD0(s);
    }
    // byte offset of the lane's piece of segment s in a row (idle lanes shadow lane 0: valid address, results unused)
    __device__ __forceinline__ int byteoff(int s, int lane) const { return 4 * (gbase[s] + segK(s) * (lane < nl ? lane : 0)); }
    __device__ __forceinline__ void load(uint32_t (&r)[NDW], const uint8_t *__restrict__ row, int lane) const
    {
#pragma unroll
        for (int s = 0; s < NSEG; s++) {
            const uint32_t *p = reinterpret_cast<const uint32_t *>(row + byteoff(s, lane));
#pragma unroll
            for (int d = 0; d < segK(s); d++)
                r[segD0(s) + d] = p[d];
        }
    }
    __device__ __forceinline__ void store_zero(uint8_t *__restrict__ row, int lane) const
    {
        if (lane < nl) {
#pragma unroll
            for (int s = 0; s < NSEG; s++) {
                uint32_t *p = reinterpret_cast<uint32_t *>(row + byteoff(s, lane));
#pragma unroll
                for (int d = 0; d < segK(s); d++)
                    p[d] = 0;
            }
        }
    }
};

template <int NDW>
struct K2BoundJob { // per-job state of the bound recurrence and of its suspect list (all wave-uniform but b*/Mprev)
    // The recurrence runs on PAIRS of 4-pixel groups (8 columns, the last one alone when NDW is odd): the taps of a
    // column still reach at most the neighbouring 4-pixel group on either side, so M = left group + own pair + right
    // group bounds every column of the pair; suspects are listed as their 4-pixel groups.
#ifndef K2B_GS
#define K2B_GS 2
#endif
    static constexpr int GS = K2B_GS > NDW ? NDW : K2B_GS; // 4-pixel groups per recurrence group
    static constexpr int NG = (NDW + GS - 1) / GS;
    // vertical 1-4-6-4-1 of the group masses as four cascaded two-tap sums (binomial = (1 + z^-1)^4): per group four
    // plain 32-bit adds and no shift / multiply; P[k][parity] = output of stage k at the previous row of that parity
    // (the row loops are unrolled by two, so nothing is ever copied).  Wide rows (NDW >= 6) keep the state in four
    // in-place accumulators instead (P[k][0]): 16 instead of 32 registers per job there, which is a wave of occupancy.
    static constexpr bool CASCADE = NDW <= 5;
    uint32_t P[4][2][NG];
    uint32_t npend, hot, jidx;
    int handover; // < 0: scanning; >= 0: first output row left to the row machine (or "nothing to do")
};

// hand the rows [y, y1) of `unit` to the row machine, in pieces of K2B_SUB rows (capacity: see launch_k2_rows)
__device__ __forceinline__ void k2b_hand_over(uint2 *__restrict__ units, uint32_t *__restrict__ nunits, uint32_t unit, int y, int y1,
                                              int lane)
{
    const int np = (y1 - y + K2B_SUB - 1) / K2B_SUB;
    if (np <= 0)
        return;
    uint32_t base = 0;
    if (lane == 0)
        base = atomicAdd(nunits, (uint32_t)np);
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
template <int KIND, bool COMPACT, bool STORE>
__global__ __launch_bounds__(256) void sus_tail_list(const uint8_t *__restrict__ frames, const uint8_t *__restrict__ mu,
                              const uint8_t *__restrict__ sigma6, const abub_job *__restrict__ jobs, int W, int H,
                              uint32_t *__restrict__ hist, uint8_t *__restrict__ img, const uint2 *__restrict__ glist,
                              const uint32_t *__restrict__ gcount, uint32_t gcap, const int32_t *__restrict__ cthr,
                              uint32_t *pairs, uint32_t pcap, uint32_t *pcount, uint32_t slot_base);
// moves `n` codes of one job to the global list at `base` (reserved by the caller); slots beyond the capacity are dropped
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

// one input row of one job: group masses m[] -> bound of output row y; suspects go to the job's LDS list.
// Everything that steers control flow is read through SGPRs (ballots, s_bcnt1), so the scan loops compile to scalar
// branches.
template <int NDW, bool SPLIT>
__device__ __forceinline__ void k2b_row(K2BoundJob<NDW> &J, const int par, uint32_t (&m)[NDW], bool emit, int y,
                                        const ScanMap<NDW, SPLIT> &map, int lane, uint32_t ngroups, uint32_t budget,
                                        uint32_t *pend, const SusList &gl)
{
    using Map = ScanMap<NDW, SPLIT>;
    constexpr int NSEG = Map::NSEG;
    // masses of the groups left of each segment's first and right of its last group: neighbour lanes inside a
    // segment (DPP), the adjacent segment's end at lane 0 / the last active lane (one v_readlane), the replicated own
    // edge group at the image border (the reflected column lies inside it).  Idle lanes (lane >= nl) carry lane 0's
    // pixels: nothing reads their masses (the last active lane takes its right neighbour from the edge value) and
    // their bounds are masked out of the row test below.
    const bool act = lane < map.nl;
    const int lastLane = map.nl - 1;
    uint32_t mLs[NSEG], mRs[NSEG];
#pragma unroll
    for (int s = 0; s < NSEG; s++) {
        const int dF = Map::segD0(s), dL = Map::segD0(s) + Map::segK(s) - 1;
        uint32_t l = __builtin_amdgcn_update_dpp(0u, m[dL], DPP_WAVE_SHR1, 0xf, 0xf, false);
        uint32_t r = __builtin_amdgcn_update_dpp(0u, m[dF], DPP_WAVE_SHL1, 0xf, 0xf, false);
        uint32_t edgeL = m[dF], edgeR = m[dL];
        if (s > 0)
            edgeL = __builtin_amdgcn_readlane(m[Map::segD0(s - 1) + Map::segK(s - 1) - 1], lastLane);
        if (s + 1 < NSEG)
            edgeR = __builtin_amdgcn_readlane(m[Map::segD0(s + 1)], 0);
        mLs[s] = lane == 0 ? edgeL : l;
        mRs[s] = lane == lastLane ? edgeR : r;
    }
    constexpr int NG = K2BoundJob<NDW>::NG;
    constexpr int GS = K2BoundJob<NDW>::GS;
    static_assert(GS == 2 || NDW == 1, "the segment tables assume pairs of groups");
    uint32_t B[NG];
    uint32_t worst = 0; // OR of the bounds: each half >= that half of every group's bound (cheaper than a packed max)
#pragma unroll
    for (int g = 0; g < NG; g++) {
        const int g0 = GS * g, g1 = GS * g + GS - 1 < NDW ? GS * g + GS - 1 : NDW - 1; // first and last 4-pixel group
        const int sg = Map::segOf(g0);
        uint32_t own = m[g0];
#pragma unroll
        for (int q = g0 + 1; q <= g1; q++)
            own += m[q];
        const bool segStart = g0 == Map::segD0(sg), segEnd = g1 == Map::segD0(sg) + Map::segK(sg) - 1;
        const uint32_t M = (segStart ? mLs[sg] : m[g0 - 1]) + own + (segEnd ? mRs[sg] : m[g1 + 1]);
        if (K2BoundJob<NDW>::CASCADE) {
            const uint32_t s1 = M + J.P[0][par ^ 1][g];
            const uint32_t s2 = s1 + J.P[1][par ^ 1][g];
            const uint32_t s3 = s2 + J.P[2][par ^ 1][g];
            B[g] = s3 + J.P[3][par ^ 1][g];
            J.P[0][par][g] = M;
            J.P[1][par][g] = s1;
            J.P[2][par][g] = s2;
            J.P[3][par][g] = s3;
        } else { // four in-place accumulators (b0, b1, b2, previous M): half the registers, a shift and a multiply more
            B[g] = J.P[0][0][g] + M;
            const uint32_t M4 = M << 2;
            J.P[0][0][g] = J.P[1][0][g] + M4;
            J.P[1][0][g] = pk_madk<6>(M, J.P[2][0][g]);
            J.P[2][0][g] = J.P[3][0][g] + M4;
            J.P[3][0][g] = M;
        }
        worst |= B[g];
    }
    const bool unsure = act && ((worst & 0xffffu) + (worst >> 16)) > 21u; // 6 * (lo + hi) < 128 <=> lo + hi <= 21
    if (!(emit && __builtin_amdgcn_ballot_w64(unsure)))
        return;
    // ---- rare: some group of this row cannot be proven zero -----------------------------------------------
    unsigned long long bm[NG]; // lanes whose recurrence group g (GS 4-pixel groups) is suspect
    bool mine[NG];
    uint32_t total = 0;
#pragma unroll
    for (int g = 0; g < NG; g++) {
        mine[g] = act && ((B[g] & 0xffffu) + (B[g] >> 16)) > 21u;
        bm[g] = __builtin_amdgcn_ballot_w64(mine[g]);
        const int nq = GS * g + GS <= NDW ? GS : NDW - GS * g; // 4-pixel groups of this recurrence group
        total += (uint32_t)nq * (uint32_t)__builtin_popcountll(bm[g]);
    }
    J.hot += total > 32u; // one row of the row machine costs about as much as 30 exact groups
    // (A full LDS list means a chunk with a lot of structure: for K2 the row machine is the cheaper way through such
    // rows -- an exact group costs 45 window loads and two 5x5 sums --, so the list is NOT flushed to the global suspect
    // list to make room, as K3 does; measured: flushing made the tail kernel 2.4x longer than the pieces it saved.)
    if (J.hot >= 4u || J.npend + total > budget) {
        J.handover = y; // dense rows (or the LDS list is full): the rest of the chunk goes to the row machine
        return;
    }
    uint32_t base = J.npend;
#pragma unroll
    for (int g = 0; g < NG; g++) {
        const unsigned long long b = bm[g];
        if (b) {
            const int nq = GS * g + GS <= NDW ? GS : NDW - GS * g;
            const int sg = Map::segOf(GS * g);
            // global index of the recurrence group's first 4-pixel group
            const uint32_t code0 = (uint32_t)y * ngroups + (uint32_t)(map.gbase[sg] + Map::segK(sg) * lane + (GS * g - Map::segD0(sg)));
            const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, 0u));
            if (mine[g]) {
#pragma unroll
                for (int q = 0; q < nq; q++)
                    pend[base + (uint32_t)nq * below + q] = code0 + q;
            }
            base += (uint32_t)nq * (uint32_t)__builtin_popcountll(b);
        }
    }
    J.npend = base;
}

// Store mode of the bound-and-verify pass: a row the scan is responsible for (everything above the hand-over row) is
// written as zeros by the scan itself -- proven rows ARE zero, and the few suspect groups of a row are overwritten by
// the wave's own tail.
template <int NDW>
__device__ __forceinline__ void k2b_store_zero_row(uint8_t *__restrict__ row)
{
    uint32_t *po = reinterpret_cast<uint32_t *>(row);
#pragma unroll
    for (int d = 0; d < NDW; d++)
        po[d] = 0;
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

// The wave's own tail (no global list, or it is full): D for the four pixels of every remembered group, one lane per
// group; histogram by global atomics (rare), optional store / candidates.
template <bool COMPACT, bool STORE>
__device__ __forceinline__ void k2b_tail(const uint32_t *pend, uint32_t npend, const abub_job jb, const uint8_t *__restrict__ frames,
                                         const uint8_t *__restrict__ sigma6, int W, int H, uint32_t *__restrict__ hist,
                                         uint8_t *__restrict__ diff, const Compact &cp, int lane)
{
    if (npend == 0)
        return;
    wave_lds_fence(); // orders the scan's LDS writes before the reads below
    const size_t P = (size_t)W * H;
    const uint32_t ngroups = (uint32_t)W / 4;
    const uint8_t *cur = frames + (size_t)jb.cur * P;
    const uint8_t *ref = frames + (size_t)jb.ref * P;
    const uint8_t *sg = sigma6 + (size_t)jb.model * P;
    // with the fused candidate list the whole wave iterates together (the reservation shuffles)
    const uint32_t nloop = COMPACT ? (npend + 63u) & ~63u : npend;
#pragma unroll 1
    for (uint32_t e = lane; e < nloop; e += 64) {
        uint32_t packed = 0, pix0 = 0;
        if (e < npend) {
            const uint32_t code = pend[e];
            const int y = (int)(code / ngroups), x0 = (int)(code % ngroups) * 4;
            packed = k2_exact_group(cur, ref, sg, y, x0, W, H);
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t d = (packed >> (8 * k)) & 0xffu;
                if (d)
                    atomicAdd(&hist[(size_t)jb.out * 256 + d], 1u);
            }
            if (STORE) // (the scan wrote this row as zeros; x0 is a multiple of 4 and W % 4 == 0: an aligned dword)
                *reinterpret_cast<uint32_t *>(diff + (size_t)jb.out * P + (size_t)y * W + x0) = packed;
            pix0 = (uint32_t)(y * W + x0);
        }
        if (COMPACT) { // candidates (value > cut) of the wave's groups: one reservation per wave
            uint32_t c = 0;
#pragma unroll
            for (int k = 0; k < 4; k++)
                c += (int)((packed >> (8 * k)) & 0xffu) > cp.thr;
            uint32_t pos = compact_reserve(cp, c);
#pragma unroll
            for (int k = 0; k < 4; k++)
                compact_put(cp, pos, (packed >> (8 * k)) & 0xffu, pix0 + k);
        }
    }
}

template <int NDW, bool STORE, bool COMPACT>
__global__ __launch_bounds__(64) void k2_bound_scan(const uint8_t *__restrict__ frames,
                                                    const uint8_t *__restrict__ sigma6,
                                                    const abub_job *__restrict__ jobs, int W, int H,
                                                    int rows_per_chunk, int nchunks, uint32_t budget,
                                                    uint2 *__restrict__ units, uint32_t *__restrict__ nunits,
                                                    uint32_t *__restrict__ hist, uint8_t *__restrict__ diff,
                                                    const int32_t *__restrict__ cthr, uint32_t *pairs, uint32_t pcap,
                                                    uint32_t *pcount, uint32_t slot_base, SusList gl)
{
    constexpr int NP = 2 * NDW;
    __shared__ uint32_t pend[K2B_PEND];
    const int lane = threadIdx.x;
    const int unit = blockIdx.x;
    const int job = unit / nchunks;
    const int chunk = unit - job * nchunks;
    const abub_job jb = jobs[job];
    const size_t P = (size_t)W * H;
    const uint8_t *cur = frames + (size_t)jb.cur * P;
    const uint8_t *ref = frames + (size_t)jb.ref * P;
    const uint8_t *sg = sigma6 + (size_t)jb.model * P;
    const int nl = W / (4 * NDW);
    const bool active = lane < nl;
    const int xoff = active ? lane * 4 * NDW : 0;
    const int y0 = chunk * rows_per_chunk;
    int y1 = y0 + rows_per_chunk;
    if (y1 > H)
        y1 = H;
    const int T = y1 - y0 + 4; // input rows y0-2 .. y1+1 (reflected); step tt bounds output row y0+tt-4
    const uint32_t ngroups = (uint32_t)W / 4;

    ScanMap<NDW, false> map;
    map.init(W);
    K2BoundJob<NDW> J;
#pragma unroll
    for (int g = 0; g < K2BoundJob<NDW>::NG; g++)
#pragma unroll
        for (int q = 0; q < 4; q++)
            J.P[q][0][g] = J.P[q][1][g] = 0;
    J.npend = J.hot = 0;
    J.jidx = (uint32_t)job;
    J.handover = -1;
    uint8_t *dbase = STORE ? diff + (size_t)jb.out * P + xoff : nullptr;

    RowIn<NDW> ring[2];
    k2_load_row<NDW>(ring[0], cur, ref, sg, reflect101(y0 - 2, H), W, xoff);
    const int Tpad = (T + 1) & ~1;
    for (int t = 0; t < Tpad && J.handover < 0; t += 2) {
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int tt = t + u;
            int tn = tt + 1 < T ? tt + 1 : T - 1;
            k2_load_row<NDW>(ring[u ^ 1], cur, ref, sg, reflect101(y0 - 2 + tn, H), W, xoff);
            if (J.handover < 0) {
                uint32_t Xp[NP], Xn[NP];
                k2_planes<NDW>(ring[u], Xp, Xn);
                uint32_t m[NDW];
#pragma unroll
                for (int g = 0; g < NDW; g++)
                    m[g] = (Xp[2 * g] + Xn[2 * g]) + (Xp[2 * g + 1] + Xn[2 * g + 1]);
                k2b_row<NDW, false>(J, u, m, tt >= 4 && tt < T, y0 + tt - 4, map, lane, ngroups, budget, pend, gl);
                if (STORE && tt >= 4 && tt < T && J.handover < 0 && active)
                    k2b_store_zero_row<NDW>(dbase + (ptrdiff_t)(y0 + tt - 4) * W);
            }
        }
    }
    if (J.handover >= 0)
        k2b_hand_over(units, nunits, (uint32_t)unit, J.handover, y1, lane);
    if (J.npend) { // the suspects go to the launch's global list; if that is full the wave evaluates them itself
        wave_lds_fence();
        uint32_t gb = 0;
        if (sus_reserve(J.npend, gl.list, gl.count, gl.cap, gb, lane)) {
            sus_copy_out(pend, J.npend, J.jidx, gl.list, gb, lane);
            return;
        }
    }
    Compact cp;
    cp.pairs = COMPACT ? pairs : nullptr;
    cp.count = pcount;
    cp.cap = pcap;
    cp.slot = jb.out + slot_base;
    cp.thr = COMPACT ? cthr[jb.out] : 255;
    k2b_tail<COMPACT, STORE>(pend, J.npend, jb, frames, sigma6, W, H, hist, diff, cp, lane);
}

// ------------------------------------------------------------------------------------------------
// k2_bound_chain: the bound scan for job lists with the trigger search's structure -- blocks of `L` consecutive jobs
// in which job q takes the cur frame of job q - S as its ref (FindTriggerFrame: S = 2, or 1 for small training sets).
// One wave serves up to K jobs of one such chain for one chunk: every frame row is loaded once and used as the cur
// row of one job and the ref row of the next, sigma6 once for all -- (K + 2) / K row loads per job instead of 3.
// The chain property is only a hint: the wave checks it on the job records and hands units it cannot chain to the row
// machine, so any job list gives the same histograms as k2_bound_scan / k2_rows.
// ------------------------------------------------------------------------------------------------
template <int NDW, int K, bool STORE, bool SPLIT>
__global__ __launch_bounds__(64) void k2_bound_chain(const uint8_t *__restrict__ frames,
                                                     const uint8_t *__restrict__ sigma6,
                                                     const abub_job *__restrict__ jobs, int L, int S, int nslot, int W,
                                                     int H, int rows_per_chunk, int nchunks, uint32_t budget,
                                                     uint2 *__restrict__ units, uint32_t *__restrict__ nunits,
                                                     uint32_t *__restrict__ hist, uint8_t *__restrict__ diff, SusList gl)
{
    __shared__ uint32_t pend[K][K2B_PEND];
    const int lane = threadIdx.x;
    const int chunk = blockIdx.x % nchunks;
    const int bs = blockIdx.x / nchunks; // (block of L jobs, slot)
    const int blk = bs / nslot;
    int slot = bs - blk * nslot;
    // slot -> (residue r, segment q of that residue's chain)
    int r = 0, nr = 0;
    for (r = 0; r < S; r++) {
        nr = (L - r + S - 1) / S; // jobs of residue r in the block
        const int ns = (nr + K - 1) / K;
        if (slot < ns)
            break;
        slot -= ns;
    }
    const int k = nr - slot * K < K ? nr - slot * K : K; // jobs of this wave (>= 1)
    K2BoundJob<NDW> J[K];
    abub_job jb[K];
#pragma unroll
    for (int t = 0; t < K; t++) {
        const int tc = t < k ? t : k - 1;
        J[t].jidx = (uint32_t)(blk * L + r + S * (slot * K + tc));
        jb[t] = jobs[J[t].jidx];
    }
    const int y0 = chunk * rows_per_chunk;
    int y1 = y0 + rows_per_chunk;
    if (y1 > H)
        y1 = H;
    bool chained = true;
#pragma unroll
    for (int t = 1; t < K; t++)
        if (t < k && (jb[t].ref != jb[t - 1].cur || jb[t].model != jb[0].model))
            chained = false;
    if (!chained) { // not the structure promised: every unit goes to the row machine whole
#pragma unroll
        for (int t = 0; t < K; t++)
            if (t < k)
                k2b_hand_over(units, nunits, J[t].jidx * (uint32_t)nchunks + (uint32_t)chunk, y0, y1, lane);
        return;
    }
    const size_t P = (size_t)W * H;
    const uint8_t *fp[K + 1];
    fp[0] = frames + (size_t)jb[0].ref * P;
#pragma unroll
    for (int t = 0; t < K; t++)
        fp[t + 1] = frames + (size_t)jb[t].cur * P;
    const uint8_t *sg = sigma6 + (size_t)jb[0].model * P;
    ScanMap<NDW, SPLIT> map;
    map.init(W);
    uint8_t *dbase[K];
#pragma unroll
    for (int t = 0; t < K; t++)
        dbase[t] = STORE ? diff + (size_t)jb[t].out * P : nullptr;
    const int T = y1 - y0 + 4;
    const uint32_t ngroups = (uint32_t)W / 4;
#pragma unroll
    for (int t = 0; t < K; t++) {
#pragma unroll
        for (int g = 0; g < K2BoundJob<NDW>::NG; g++)
#pragma unroll
            for (int q = 0; q < 4; q++)
                J[t].P[q][0][g] = J[t].P[q][1][g] = 0;
        J[t].npend = J[t].hot = 0;
        J[t].handover = t < k ? -1 : 0x7fffffff; // (jobs beyond k do nothing and report nothing)
    }

    // rows are fetched one step ahead through two register sets ([K + 1] = sigma6); the loop is unrolled by two so
    // that each set -- and each parity of the recurrence state -- is a fixed set of registers.  (Fetching two steps
    // ahead was measured equal: the scan is bound by VALU issue at the clock the chip holds under HBM load.)
    constexpr int PF = 1, U = 2;
    uint32_t raw[U][K + 2][NDW];
#define K2C_LOAD(SL, Y)                                                                       \
    {                                                                                         \
        const size_t o_ = (size_t)(Y) * W;                                                    \
        _Pragma("unroll") for (int f = 0; f <= K; f++) map.load(raw[SL][f], fp[f] + o_, lane); \
        map.load(raw[SL][K + 1], sg + o_, lane);                                              \
    }
#pragma unroll
    for (int q = 0; q < PF; q++) {
        const int tq = q < T ? q : T - 1;
        K2C_LOAD(q, reflect101(y0 - 2 + tq, H));
    }
    const int Tpad = (T + U - 1) / U * U;
    for (int t0 = 0; t0 < Tpad; t0 += U) {
        bool all_done = true;
#pragma unroll
        for (int t = 0; t < K; t++)
            all_done = all_done && J[t].handover >= 0;
        if (all_done)
            break;
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int tt = t0 + u;
            const int tn = tt + PF < T ? tt + PF : T - 1;
            K2C_LOAD((u + PF) % U, reflect101(y0 - 2 + tn, H));
            // widened frame rows (pw / cw) and frame + sigma6 (ps / cs): each is computed once per frame and serves the
            // job that has the frame as cur and the job that has it as ref
            uint32_t sw[2 * NDW], pw[2 * NDW], ps[2 * NDW];
#pragma unroll
            for (int d = 0; d < NDW; d++) {
                sw[2 * d] = widen_lo(raw[u][K + 1][d]);
                sw[2 * d + 1] = widen_hi(raw[u][K + 1][d]);
                pw[2 * d] = widen_lo(raw[u][0][d]);
                pw[2 * d + 1] = widen_hi(raw[u][0][d]);
                ps[2 * d] = pw[2 * d] + sw[2 * d];
                ps[2 * d + 1] = pw[2 * d + 1] + sw[2 * d + 1];
            }
#pragma unroll
            for (int t = 0; t < K; t++) {
                uint32_t cw[2 * NDW], cs[2 * NDW];
#pragma unroll
                for (int d = 0; d < NDW; d++) {
                    cw[2 * d] = widen_lo(raw[u][t + 1][d]);
                    cw[2 * d + 1] = widen_hi(raw[u][t + 1][d]);
                    cs[2 * d] = cw[2 * d] + sw[2 * d];
                    cs[2 * d + 1] = cw[2 * d + 1] + sw[2 * d + 1];
                }
                if (J[t].handover < 0) {
                    uint32_t m[NDW];
#pragma unroll
                    for (int g = 0; g < NDW; g++) // sat(c - (r + s)) + sat(r - (c + s)), both pairs of the group
                        m[g] = (pk_subsat(cw[2 * g], ps[2 * g]) + pk_subsat(pw[2 * g], cs[2 * g])) +
                               (pk_subsat(cw[2 * g + 1], ps[2 * g + 1]) + pk_subsat(pw[2 * g + 1], cs[2 * g + 1]));
                    k2b_row<NDW, SPLIT>(J[t], u, m, tt >= 4 && tt < T, y0 + tt - 4, map, lane, ngroups, budget, pend[t], gl);
                    if (STORE && tt >= 4 && tt < T && J[t].handover < 0)
                        map.store_zero(dbase[t] + (ptrdiff_t)(y0 + tt - 4) * W, lane);
                }
#pragma unroll
                for (int j = 0; j < 2 * NDW; j++) {
                    pw[j] = cw[j];
                    ps[j] = cs[j];
                }
            }
        }
    }
#undef K2C_LOAD
    uint32_t tot = 0;
#pragma unroll
    for (int t = 0; t < K; t++) {
        if (t < k) {
            if (J[t].handover >= 0)
                k2b_hand_over(units, nunits, J[t].jidx * (uint32_t)nchunks + (uint32_t)chunk, J[t].handover, y1, lane);
            tot += J[t].npend;
        }
    }
    if (tot == 0)
        return;
    wave_lds_fence();
    uint32_t gb = 0;
    if (sus_reserve(tot, gl.list, gl.count, gl.cap, gb, lane)) { // the whole wave's suspects in one reservation
#pragma unroll
        for (int t = 0; t < K; t++)
            if (t < k) {
                sus_copy_out(pend[t], J[t].npend, J[t].jidx, gl.list, gb, lane);
                gb += J[t].npend;
            }
        return;
    }
    Compact cp; // no room in the global list: the wave evaluates its suspects itself
    cp.pairs = nullptr;
    cp.count = nullptr;
    cp.cap = 0;
    cp.slot = 0;
    cp.thr = 255;
#pragma unroll
    for (int t = 0; t < K; t++)
        if (t < k)
            k2b_tail<false, STORE>(pend[t], J[t].npend, jb[t], frames, sigma6, W, H, hist, diff, cp, lane);
}

// ------------------------------------------------------------------------------------------------
// K2 generic: any W,H and the ROI overload.  256-thread workgroup, 32x8 output tile, LDS tile of
// packed (pos | neg<<16) with a 2-pixel halo; borders reflect inside the ROI.
// ------------------------------------------------------------------------------------------------
#define G_TW 32
#define G_TH 8
__global__ __launch_bounds__(256) void k2_generic(const uint8_t *__restrict__ frames,
                                                  const uint8_t *__restrict__ sigma6,
                                                  const abub_job *__restrict__ jobs,
                                                  abub_job single, int use_single, int W, int H,
                                                  int rx, int ry, int rw, int rh,
                                                  uint32_t *__restrict__ hist,
                                                  uint8_t *__restrict__ diff)
{
    __shared__ uint32_t tile[(G_TH + 4) * (G_TW + 4)];
    __shared__ uint32_t lh[256];
    const abub_job jb = use_single ? single : jobs[blockIdx.z];
    const size_t P = (size_t)W * H;
    const uint8_t *cur = frames + (size_t)jb.cur * P;
    const uint8_t *ref = frames + (size_t)jb.ref * P;
    const uint8_t *sg = sigma6 + (size_t)jb.model * P;
    const int tid = threadIdx.x;
    lh[tid] = 0;
    const int tx0 = blockIdx.x * G_TW, ty0 = blockIdx.y * G_TH; // in ROI coordinates
    for (int i = tid; i < (G_TH + 4) * (G_TW + 4); i += 256) {
        int ly = i / (G_TW + 4), lx = i - ly * (G_TW + 4);
        int x = reflect101(tx0 + lx - 2, rw) + rx;
        int y = reflect101(ty0 + ly - 2, rh) + ry;
        size_t o = (size_t)y * W + x;
        int c = cur[o], r = ref[o], s6 = sg[o];
        int pos = c - r;
        pos = pos < 0 ? 0 : pos;
        pos -= s6;
        pos = pos < 0 ? 0 : pos;
        int neg = r - c;
        neg = neg < 0 ? 0 : neg;
        neg -= s6;
        neg = neg < 0 ? 0 : neg;
        tile[i] = (uint32_t)pos | ((uint32_t)neg << 16);
    }
    __syncthreads();
    const int lx = tid % G_TW, ly = tid / G_TW;
    const int x = tx0 + lx, y = ty0 + ly;
    if (x < rw && y < rh) {
        const uint32_t w[5] = {1, 4, 6, 4, 1};
        uint32_t acc = 0;
#pragma unroll
        for (int i = 0; i < 5; i++) {
            uint32_t rowacc = 0;
#pragma unroll
            for (int j = 0; j < 5; j++)
                rowacc += w[j] * tile[(ly + i) * (G_TW + 4) + lx + j];
            acc += w[i] * rowacc;
        }
        acc += 0x00800080u;
        int a = (acc >> 8) & 0xff, b = acc >> 24;
        int d = a > b ? a - b : b - a;
        if (diff)
            diff[(size_t)jb.out * P + (size_t)(y + ry) * W + (x + rx)] = (uint8_t)d;
        if (d)
            atomicAdd(&lh[d], 1u);
    }
    __syncthreads();
    uint32_t v = lh[tid];
    if (v && tid)
        atomicAdd(&hist[(size_t)jb.out * 256 + tid], v);
}

static int pick_ndw(int W)
{
    if (W < 4 || (W & 3))
        return 0;
    int nd = W / 4;
    for (int ndw = 1; ndw <= 8; ndw++)
        if (nd % ndw == 0 && nd / ndw <= 64)
            return ndw;
    return 0;
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
static void *k2_scratch(hipStream_t st, size_t bytes, std::unique_lock<std::mutex> &hold)
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

struct CompactArgs {
    const int32_t *cthr;
    uint32_t *pairs;
    uint32_t cap;
    uint32_t *count;
    uint32_t slot_base; // added to job.out in the list entries (several launches share one list)
    int chain_len = 0, chain_stride = 0; // trigger-only hint: blocks of chain_len jobs, job q refs the cur of job q - stride
};

template <int NDW, int PF>
static void launch_k2_rows_pf(const uint8_t *frames, const uint8_t *sigma6, const abub_job *jobs,
                              int njobs, int W, int H, int R, int nchunks, uint32_t *hist, uint8_t *diff,
                              const CompactArgs &ca, hipStream_t st)
{
    dim3 grid((unsigned)njobs * nchunks), block(64);
#define K2_LAUNCH(ST, CO)                                                                                        \
    hipLaunchKernelGGL((k2_rows<NDW, ST, PF, CO>), grid, block, 0, st, frames, sigma6, jobs, W, H, R, nchunks, \
                       hist, diff, ca.cthr, ca.pairs, ca.cap, ca.count, ca.slot_base, nullptr, nullptr)
    if (ca.cthr) {
        if (diff)
            K2_LAUNCH(true, true);
        else
            K2_LAUNCH(false, true);
    } else {
        if (diff)
            K2_LAUNCH(true, false);
        else
            K2_LAUNCH(false, false);
    }
#undef K2_LAUNCH
}

// Tuning knobs of the K2 launchers.  Defaults come from the environment once (ABUB_K2_BOUND, ABUB_K2_CHAIN,
// ABUB_K2_BUDGET, ABUB_K2_PF); abub_k2_set_option() overrides them at run time (tests and benches switch between
// the bound-and-verify pass and the plain row machine inside one process).
struct K2Options {
    int bound = 1;     // 0: always the full row machine (k2_rows); 1: bound-and-verify (trigger-only AND store mode)
    int chain = -1;    // jobs per wave in the chained scan: 2 or 3; 0 = never chain; -1 = automatic (3 for rows of up to
                       // 5 dwords per lane -- measured 3 % faster at W = 1280 --, 2 for wider rows: registers)
    int budget = 512;  // suspects a chunk may remember (LDS) before it hands its rows over (<= K2B_PEND)
    int pf = 1;        // software-prefetch depth of the row machine in rows (1 or 2)
    int split = 1;     // chained scan: "split" lane mapping where the row width allows it (0: always blocked)
    int list = 0;      // 1: suspects go to a global list that a second kernel evaluates (sus_tail_list), 0: the scanning
                       // waves evaluate their own.  K2's exact groups are expensive (45 window loads, two 5x5 sums) and few
                       // per wave: measured on the bench's trigger pass, in-wave 2.36 ms vs 2.47 ms through the list (K3,
                       // where every frame has its bubble and a group costs a 3x3 box, is the other way round: list always)
    bool loaded = false;
};
static K2Options g_k2opt;
static std::mutex g_k2optMu;
static K2Options k2_options()
{
    std::lock_guard<std::mutex> lock(g_k2optMu);
    if (!g_k2opt.loaded) {
        if (const char *e = getenv("ABUB_K2_BOUND"))
            g_k2opt.bound = atoi(e);
        if (const char *e = getenv("ABUB_K2_CHAIN"))
            g_k2opt.chain = atoi(e);
        if (const char *e = getenv("ABUB_K2_BUDGET"))
            if (atoi(e) > 0)
                g_k2opt.budget = atoi(e);
        if (const char *e = getenv("ABUB_K2_PF"))
            g_k2opt.pf = atoi(e);
        if (const char *e = getenv("ABUB_K2_SPLIT"))
            g_k2opt.split = atoi(e);
        if (const char *e = getenv("ABUB_K2_LIST"))
            g_k2opt.list = atoi(e);
        g_k2opt.loaded = true;
    }
    return g_k2opt;
}

extern "C" int abub_k2_set_option(const char *name, int value)
{
    if (!name)
        return set_err(ABUB_E_INVALID, "abub_k2_set_option: null name");
    (void)k2_options();
    std::lock_guard<std::mutex> lock(g_k2optMu);
    if (!strcmp(name, "bound"))
        g_k2opt.bound = value;
    else if (!strcmp(name, "chain"))
        g_k2opt.chain = value;
    else if (!strcmp(name, "budget") && value > 0)
        g_k2opt.budget = value;
    else if (!strcmp(name, "pf"))
        g_k2opt.pf = value;
    else if (!strcmp(name, "split"))
        g_k2opt.split = value;
    else if (!strcmp(name, "list"))
        g_k2opt.list = value;
    else
        return set_err(ABUB_E_INVALID, "abub_k2_set_option: unknown option or bad value");
    return ABUB_OK;
}

template <int NDW>
static int launch_k2_rows(const uint8_t *frames, const uint8_t *sigma6, const abub_job *jobs,
                           int njobs, int W, int H, int R, int nchunks, uint32_t *hist, uint8_t *diff,
                           const CompactArgs &ca, hipStream_t st)
{
    const K2Options opt = k2_options();
    if (opt.bound && (size_t)H * (size_t)(W / 4) < ((size_t)1 << 32) && H < 65536) { // (row, group) codes are 32-bit
        // bound-and-verify (see k2_bound_scan); with `diff` the scan also writes the rows it proves (or remembers)
        const size_t nunits = (size_t)njobs * nchunks;
        // a chunk remembers up to `budget` suspicious groups in LDS, then it hands its remaining rows to the row machine
        const uint32_t budget = (uint32_t)(opt.budget < K2B_PEND ? opt.budget : K2B_PEND);
        const size_t unitCap = nunits * (size_t)((R + K2B_SUB - 1) / K2B_SUB); // handed-over pieces, worst case
        const size_t unitBytes = (unitCap * sizeof(uint2) + 255) & ~(size_t)255;
        // the global suspect list (see sus_tail_list): room for 1024 groups per job on average within 64 K .. 16 M entries
        size_t gcap = (size_t)njobs * 1024;
        gcap = gcap < ((size_t)1 << 16) ? ((size_t)1 << 16) : (gcap > ((size_t)1 << 24) ? ((size_t)1 << 24) : gcap);
        if (!opt.list)
            gcap = 0;
        const size_t bytes = 256 + unitBytes + gcap * sizeof(uint2) + 256;
        std::unique_lock<std::mutex> hold;
        uint8_t *scr = (uint8_t *)k2_scratch(st, bytes, hold);
        if (!scr)
            return set_err(ABUB_E_HIP, "abub_diff_hist_dev: scratch allocation failed");
        uint32_t *counters = (uint32_t *)scr; // [0] = handed-over pieces, [32] = entries of the global suspect list
        uint2 *units = (uint2 *)(scr + 256);
        SusList gl;
        gl.list = gcap ? (uint2 *)(scr + 256 + unitBytes) : nullptr;
        gl.count = counters + 32;
        gl.cap = (uint32_t)gcap;
        const unsigned tgrid = (unsigned)((gcap + 256 * SUSL_UB - 1) / (256 * SUSL_UB) < 2048 ? (gcap + 256 * SUSL_UB - 1) / (256 * SUSL_UB) : 2048);
        HIPCHK(hipMemsetAsync(counters, 0, 256, st));
        const int L = ca.chain_len, S = ca.chain_stride;
        const int chainK = opt.chain < 0 ? (NDW <= 5 ? 3 : 2) : opt.chain;
        if (chainK >= 2 && L > 0 && S > 0 && S <= 8 && njobs % L == 0 && !ca.cthr) {
            const int Kc = chainK >= 3 ? 3 : 2;
            int nslot = 0;
            for (int r = 0; r < S; r++) {
                const int nr = (L - r + S - 1) / S;
                nslot += nr > 0 ? (nr + Kc - 1) / Kc : 0;
            }
            const dim3 grid((unsigned)((size_t)(njobs / L) * nslot * nchunks));
            // the scan's own lane mapping: whole 16 / 8 / 4-byte pieces per lane ("split") where the row decomposes
            // that way (W = 1280, 1680, ...), the row machine's blocked mapping otherwise
            // (measured, A/B on one box: store mode -13 % at W = 1280 and -4 % at 1680 -- the zero rows go out as whole
            // lines --; trigger-only equal at 1280 and 2-7 % slower at 1680, where the three segments cost more
            // neighbour-exchange instructions than the loads gain: split there only when D is stored)
            constexpr bool CAN_SPLIT = NDW >= 5 && NDW <= 7;
            const bool split = CAN_SPLIT && opt.split && (diff != nullptr || NDW == 5 || opt.split > 1);
#define K2C_ARGS frames, sigma6, jobs, L, S, nslot, W, H, R, nchunks, budget, units, counters, hist, diff, gl
#define K2C_LAUNCH_SP(KK, ST, SP) hipLaunchKernelGGL((k2_bound_chain<NDW, KK, ST, SP>), grid, dim3(64), 0, st, K2C_ARGS)
#define K2C_LAUNCH(KK, ST)                                                                                          \
    if (split) {                                                                                                    \
        K2C_LAUNCH_SP(KK, ST, CAN_SPLIT);                                                                           \
    } else {                                                                                                        \
        K2C_LAUNCH_SP(KK, ST, false);                                                                               \
    }
            if (Kc == 3) {
                if (diff) {
                    K2C_LAUNCH(3, true);
                } else {
                    K2C_LAUNCH(3, false);
                }
            } else {
                if (diff) {
                    K2C_LAUNCH(2, true);
                } else {
                    K2C_LAUNCH(2, false);
                }
            }
#undef K2C_LAUNCH_SP
#undef K2C_ARGS
#undef K2C_LAUNCH
        } else {
#define K2S_LAUNCH(ST, CO)                                                                                          \
    hipLaunchKernelGGL((k2_bound_scan<NDW, ST, CO>), dim3((unsigned)nunits), dim3(64), 0, st, frames, sigma6, jobs, \
                       W, H, R, nchunks, budget, units, counters, hist, diff, ca.cthr, ca.pairs, ca.cap, ca.count,  \
                       ca.slot_base, gl)
            if (ca.cthr) {
                if (diff)
                    K2S_LAUNCH(true, true);
                else
                    K2S_LAUNCH(false, true);
            } else {
                if (diff)
                    K2S_LAUNCH(true, false);
                else
                    K2S_LAUNCH(false, false);
            }
#undef K2S_LAUNCH
        }
        // the suspects of all scanning waves, evaluated exactly (after the scan: in store mode it overwrites zero rows)
        if (gl.list) {
#define K2T_LAUNCH(CO, ST)                                                                                          \
    hipLaunchKernelGGL((sus_tail_list<2, CO, ST>), dim3(tgrid), dim3(256), 0, st, frames, (const uint8_t *)nullptr,  \
                       sigma6, jobs, W, H, hist, diff, gl.list, gl.count, gl.cap, ca.cthr, ca.pairs, ca.cap, ca.count, \
                       ca.slot_base)
            if (ca.cthr) {
                if (diff) {
                    K2T_LAUNCH(true, true);
                } else {
                    K2T_LAUNCH(true, false);
                }
            } else {
                if (diff) {
                    K2T_LAUNCH(false, true);
                } else {
                    K2T_LAUNCH(false, false);
                }
            }
#undef K2T_LAUNCH
        }
        // the handed-over row ranges through the row machine's list mode (grid-stride over the pieces); with the fused
        // candidate list (cthr) it emits the candidates, with `diff` it writes its rows
        const unsigned g3 = (unsigned)(unitCap < 4096 ? unitCap : 4096);
#define K2R_LAUNCH(ST, CO)                                                                                          \
    hipLaunchKernelGGL((k2_rows<NDW, ST, 1, CO>), dim3(g3), dim3(64), 0, st, frames, sigma6, jobs, W, H, R, nchunks, \
                       hist, diff, ca.cthr, ca.pairs, ca.cap, ca.count, ca.slot_base, units, counters)
        if (ca.cthr) {
            if (diff)
                K2R_LAUNCH(true, true);
            else
                K2R_LAUNCH(false, true);
        } else {
            if (diff)
                K2R_LAUNCH(true, false);
            else
                K2R_LAUNCH(false, false);
        }
#undef K2R_LAUNCH
        return ABUB_OK;
    }
    // prefetch depth 1 won on MI355X: depth 2/3 rings cost a wave of occupancy and ran 10-17 % slower
    // (measured in round 1, see DESIGN.md "Tuning log")
    if (opt.pf == 2)
        launch_k2_rows_pf<NDW, 2>(frames, sigma6, jobs, njobs, W, H, R, nchunks, hist, diff, ca, st);
    else
        launch_k2_rows_pf<NDW, 1>(frames, sigma6, jobs, njobs, W, H, R, nchunks, hist, diff, ca, st);
    return ABUB_OK;
}

static int diff_hist_impl(const uint8_t *frames, const uint8_t *sigma6, const abub_job *jobs, int njobs,
                          int W, int H, uint32_t *hist, uint8_t *diff, int rows_per_chunk,
                          const CompactArgs &ca, void *stream)
{
    if (!frames || !sigma6 || !jobs || !hist || W <= 0 || H <= 0 || njobs < 0 || rows_per_chunk < 0)
        return set_err(ABUB_E_INVALID, "abub_diff_hist_dev: bad arguments");
    if (njobs == 0)
        return ABUB_OK;
    hipStream_t st = (hipStream_t)stream;
    // hist slots are jb.out-indexed; the caller guarantees out < nslots == njobs for stack batches.
    // We zero and finalise exactly njobs consecutive slots starting at 0 (documented contract).
    HIPCHK(hipMemsetAsync(hist, 0, (size_t)njobs * 256 * sizeof(uint32_t), st));
    int ndw = pick_ndw(W);
    if (ndw) {
        int R = rows_per_chunk;
        if (R == 0) {
            // many jobs: 8 chunks per frame (<= ~3% vertical halo re-reads, chunk id == XCD id);
            // fewer jobs: more, shorter chunks so that the launch still offers >= ~8k waves to the chip
            int nch = 8;
            if ((long long)njobs * nch < 8192)
                nch = (8192 + njobs - 1) / njobs;
            static int k2chunks = -1;
            if (k2chunks < 0) {
                const char *e = getenv("ABUB_K2_CHUNKS"); // tuning knob: chunks per frame (0 = automatic)
                k2chunks = e ? atoi(e) : 0;
            }
            if (k2chunks > 0)
                nch = k2chunks;
            nch = (nch + 7) / 8 * 8; // keep chunk id == XCD id
            R = (H + nch - 1) / nch;
            if (R < 16)
                R = 16;
        }
        int nchunks = (H + R - 1) / R;
        int rc = ABUB_OK;
        switch (ndw) {
        case 1: rc = launch_k2_rows<1>(frames, sigma6, jobs, njobs, W, H, R, nchunks, hist, diff, ca, st); break;
        case 2: rc = launch_k2_rows<2>(frames, sigma6, jobs, njobs, W, H, R, nchunks, hist, diff, ca, st); break;
        case 3: rc = launch_k2_rows<3>(frames, sigma6, jobs, njobs, W, H, R, nchunks, hist, diff, ca, st); break;
        case 4: rc = launch_k2_rows<4>(frames, sigma6, jobs, njobs, W, H, R, nchunks, hist, diff, ca, st); break;
        case 5: rc = launch_k2_rows<5>(frames, sigma6, jobs, njobs, W, H, R, nchunks, hist, diff, ca, st); break;
        case 6: rc = launch_k2_rows<6>(frames, sigma6, jobs, njobs, W, H, R, nchunks, hist, diff, ca, st); break;
        case 7: rc = launch_k2_rows<7>(frames, sigma6, jobs, njobs, W, H, R, nchunks, hist, diff, ca, st); break;
        default: rc = launch_k2_rows<8>(frames, sigma6, jobs, njobs, W, H, R, nchunks, hist, diff, ca, st); break;
        }
        if (rc != ABUB_OK)
            return rc;
    } else {
        if (ca.cthr)
            return set_err(ABUB_E_INVALID, "fused compaction needs the fast path (W % 4 == 0, W <= 2048)");
        abub_job dummy = {0, 0, 0, 0};
        dim3 grid((W + G_TW - 1) / G_TW, (H + G_TH - 1) / G_TH, njobs), block(256);
        if (grid.z > 65535)
            return set_err(ABUB_E_INVALID, "abub_diff_hist_dev: too many jobs for the generic kernel");
        hipLaunchKernelGGL(k2_generic, grid, block, 0, st, frames, sigma6, jobs, dummy, 0, W, H, 0, 0,
                           W, H, hist, diff);
    }
    HIPCHK(hipGetLastError());
    hipLaunchKernelGGL(k_hist_bin0, dim3(njobs), dim3(64), 0, st, hist, (uint32_t)((size_t)W * H));
    HIPCHK(hipGetLastError());
    return ABUB_OK;
}

extern "C" int abub_diff_hist_dev(const uint8_t *frames, const uint8_t *sigma6, const abub_job *jobs,
                                  int njobs, int W, int H, uint32_t *hist, uint8_t *diff,
                                  int rows_per_chunk, void *stream)
{
    CompactArgs ca = {nullptr, nullptr, 0, nullptr, 0};
    return diff_hist_impl(frames, sigma6, jobs, njobs, W, H, hist, diff, rows_per_chunk, ca, stream);
}

extern "C" int abub_diff_hist_chained_dev(const uint8_t *frames, const uint8_t *sigma6, const abub_job *jobs,
                                          int njobs, int W, int H, uint32_t *hist, int chain_len, int chain_stride,
                                          void *stream)
{
    if (chain_len < 0 || chain_stride < 0)
        return set_err(ABUB_E_INVALID, "abub_diff_hist_chained_dev: bad arguments");
    CompactArgs ca = {nullptr, nullptr, 0, nullptr, 0};
    ca.chain_len = chain_len;
    ca.chain_stride = chain_stride;
    return diff_hist_impl(frames, sigma6, jobs, njobs, W, H, hist, nullptr, 0, ca, stream);
}

extern "C" int abub_diff_hist_chained_store_dev(const uint8_t *frames, const uint8_t *sigma6, const abub_job *jobs,
                                                int njobs, int W, int H, uint32_t *hist, uint8_t *diff, int chain_len,
                                                int chain_stride, void *stream)
{
    if (chain_len < 0 || chain_stride < 0 || !diff)
        return set_err(ABUB_E_INVALID, "abub_diff_hist_chained_store_dev: bad arguments");
    CompactArgs ca = {nullptr, nullptr, 0, nullptr, 0};
    ca.chain_len = chain_len;
    ca.chain_stride = chain_stride;
    return diff_hist_impl(frames, sigma6, jobs, njobs, W, H, hist, diff, 0, ca, stream);
}

extern "C" int abub_fast_path(int W) { return pick_ndw(W) != 0; }

extern "C" int abub_diff_hist_compact_dev(const uint8_t *frames, const uint8_t *sigma6, const abub_job *jobs,
                                          int njobs, int W, int H, uint32_t *hist, uint8_t *diff,
                                          const int32_t *cthr, uint32_t *pairs, uint32_t cap,
                                          uint32_t *count, uint32_t slot_base, void *stream)
{
    if (!cthr || !pairs || !count || cap == 0)
        return set_err(ABUB_E_INVALID, "abub_diff_hist_compact_dev: bad arguments");
    CompactArgs ca = {cthr, pairs, cap, count, slot_base};
    return diff_hist_impl(frames, sigma6, jobs, njobs, W, H, hist, diff, 0, ca, stream);
}

extern "C" int abub_diff_roi_dev(const uint8_t *cur, const uint8_t *ref, const uint8_t *sigma6, int W,
                                 int H, int rx, int ry, int rw, int rh, uint8_t *diff, uint32_t *hist,
                                 void *stream)
{
    if (!cur || !ref || !sigma6 || !diff || !hist || W <= 0 || H <= 0 || rx < 0 || ry < 0 ||
        rw < 0 || rh < 0 || rx + rw > W || ry + rh > H)
        return set_err(ABUB_E_INVALID, "abub_diff_roi_dev: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    size_t P = (size_t)W * H;
    HIPCHK(hipMemsetAsync(diff, 0, P, st)); // cv::Mat::zeros, AnalyzerUnit.cpp:349
    HIPCHK(hipMemsetAsync(hist, 0, 256 * sizeof(uint32_t), st));
    if (rw > 0 && rh > 0) {
        // cur/ref/sigma6 are separate allocations: express them as frame offsets from `cur`... the
        // generic kernel indexes frames by element count, so pass the three bases through pointer
        // differences only when they share a slab.  Simplest exact way: one job with index 0 and
        // dedicated base pointers.
        abub_job jb = {0, 0, 0, 0};
        dim3 grid((rw + G_TW - 1) / G_TW, (rh + G_TH - 1) / G_TH, 1), block(256);
        // ref is addressed relative to cur: only valid if both lie in one slab at a multiple of P.
        ptrdiff_t dref = ref - cur;
        if (dref % (ptrdiff_t)P != 0 || dref < 0)
            return set_err(ABUB_E_INVALID, "abub_diff_roi_dev: ref must follow cur in the same slab at a multiple of W*H");
        jb.ref = (uint32_t)(dref / (ptrdiff_t)P);
        hipLaunchKernelGGL(k2_generic, grid, block, 0, st, cur, sigma6, (const abub_job *)nullptr, jb, 1,
                           W, H, rx, ry, rw, rh, hist, diff);
        HIPCHK(hipGetLastError());
    }
    hipLaunchKernelGGL(k_hist_bin0, dim3(1), dim3(64), 0, st, hist, (uint32_t)P);
    HIPCHK(hipGetLastError());
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
// K3 generic: O = max(0,|f-mu| - 6 sigma), 3x3 box (S+4)/9, histogram.  32x8 tile + 1-pixel halo.
// (S+4)/9 == ((S+4)*7282)>>16 for every reachable S (0..2295): checked exhaustively in tests.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k3_generic(const uint8_t *__restrict__ frames,
                                                  const uint8_t *__restrict__ mu,
                                                  const uint8_t *__restrict__ sigma6,
                                                  const abub_job *__restrict__ jobs, int W, int H,
                                                  uint32_t *__restrict__ hist, uint8_t *__restrict__ img)
{
    __shared__ uint16_t tile[(G_TH + 2) * (G_TW + 2)];
    __shared__ uint32_t lh[256];
    const abub_job jb = jobs[blockIdx.z];
    const size_t P = (size_t)W * H;
    const uint8_t *f = frames + (size_t)jb.cur * P;
    const uint8_t *m = mu + (size_t)jb.model * P;
    const uint8_t *sg = sigma6 + (size_t)jb.model * P;
    const int tid = threadIdx.x;
    lh[tid] = 0;
    const int tx0 = blockIdx.x * G_TW, ty0 = blockIdx.y * G_TH;
    for (int i = tid; i < (G_TH + 2) * (G_TW + 2); i += 256) {
        int ly = i / (G_TW + 2), lx = i - ly * (G_TW + 2);
        int x = reflect101(tx0 + lx - 1, W);
        int y = reflect101(ty0 + ly - 1, H);
        size_t o = (size_t)y * W + x;
        int a = (int)f[o] - (int)m[o];
        a = a < 0 ? -a : a;
        a -= (int)sg[o];
        tile[i] = (uint16_t)(a < 0 ? 0 : a);
    }
    __syncthreads();
    const int lx = tid % G_TW, ly = tid / G_TW;
    const int x = tx0 + lx, y = ty0 + ly;
    if (x < W && y < H) {
        uint32_t s = 0;
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 0; j < 3; j++)
                s += tile[(ly + i) * (G_TW + 2) + lx + j];
        uint32_t v = (s + 4) / 9;
        if (img)
            img[(size_t)jb.out * P + (size_t)y * W + x] = (uint8_t)v;
        if (v)
            atomicAdd(&lh[v], 1u);
    }
    __syncthreads();
    uint32_t v = lh[tid];
    if (v && tid)
        atomicAdd(&hist[(size_t)jb.out * 256 + tid], v);
}

// ------------------------------------------------------------------------------------------------
// K3 fast: register-rolling rows, same lane mapping as K2.  O = sat(|f - mu| - sigma6) as
//     O = sat(f - HI) | sat(LO - f),   HI = mu + sigma6,   LO = sat(mu - sigma6)        (L3Localizer.cpp:779-782)
// on u16 pairs (at most one of the two terms is non-zero; HI needs no clamp: f <= 255 < 256 <= HI whenever it would
// saturate).  Two kernels, like K2's bound-and-verify:
//   k3_bound_scan A tracking frame differs from its camera's mean by more than 6 sigma only where a bubble is (and
//                 at the odd hot pixel), so almost every output pixel of the 3x3 box is zero.  One wave scans the
//                 rows of a chunk for KF consecutive jobs that share their model (the tracking frames of a stack):
//                 mu / sigma6 rows are loaded and turned into HI / LO once for all KF frames -- (KF + 2) / KF row
//                 streams per frame instead of 3 -- and proves rows zero from a bound on the box sums (see the
//                 kernel); suspect groups are computed exactly by the same wave, dense chunks are handed over.
//   k3_rows       the row machine -- horizontal 3-tap via one alignbit per pair, vertical 3-tap as two in-place
//                 accumulators, (S+4)/9 as ((S+4)*7282)>>16 (exact for S <= 2295, checked exhaustively in tests),
//                 LDS histogram and fused compaction in the rare non-zero path, optional image store -- on whole
//                 chunks (K3 without the scan: ABUB_K3_SCAN=0, or jobs the scan cannot group) or on listed pieces.
// ------------------------------------------------------------------------------------------------

template <int NDW>
__device__ __forceinline__ void k3_thresholds(const uint32_t (&mraw)[NDW], const uint32_t (&sraw)[NDW], uint32_t (&HI)[2 * NDW],
                                              uint32_t (&LO)[2 * NDW])
{
#pragma unroll
    for (int d = 0; d < NDW; d++) {
        const uint32_t m0 = widen_lo(mraw[d]), m1 = widen_hi(mraw[d]);
        const uint32_t s0 = widen_lo(sraw[d]), s1 = widen_hi(sraw[d]);
        HI[2 * d] = m0 + s0;
        HI[2 * d + 1] = m1 + s1;
        LO[2 * d] = pk_subsat(m0, s0);
        LO[2 * d + 1] = pk_subsat(m1, s1);
    }
}

template <int NDW, bool STORE>
__device__ __forceinline__ void k3_row(const uint32_t (&fr)[NDW], const uint32_t (&HI)[2 * NDW], const uint32_t (&LO)[2 * NDW],
                                       uint32_t (&a0)[2 * NDW], uint32_t (&xp)[2 * NDW], bool emit, bool active,
                                       bool first_lane, bool last_lane, uint32_t *lh, uint32_t *__restrict__ po,
                                       const Compact &cp, uint32_t pix0, int &zrun)
{
    constexpr int NP = 2 * NDW;
    uint32_t X[NP];
#pragma unroll
    for (int d = 0; d < NDW; d++) {
        const uint32_t f0 = widen_lo(fr[d]), f1 = widen_hi(fr[d]);
        X[2 * d] = pk_subsat(f0, HI[2 * d]) | pk_subsat(LO[2 * d], f0);
        X[2 * d + 1] = pk_subsat(f1, HI[2 * d + 1]) | pk_subsat(LO[2 * d + 1], f1);
    }
    // zero-run shortcut (wave-uniform, as in K2): two all-zero rows drain a0 and xp; from then on an all-zero row
    // changes nothing and emits (0 + 4) / 9 = 0
    {
        uint32_t nz = 0;
#pragma unroll
        for (int j = 0; j < NP; j++)
            nz |= X[j];
        const bool rowzero = __builtin_amdgcn_ballot_w64(nz != 0) == 0;
        if (rowzero && zrun >= 2) {
            if (STORE && emit && active) {
#pragma unroll
                for (int d = 0; d < NDW; d++)
                    po[d] = 0;
            }
            return;
        }
        zrun = rowzero ? zrun + 1 : 0;
    }
    // neighbours: only p[-1] (hi half of L) and p[n] (lo half of R) are used; reflect-101 in-lane
    uint32_t L = __builtin_amdgcn_update_dpp(0u, X[NP - 1], DPP_WAVE_SHR1, 0xf, 0xf, false);
    uint32_t R = __builtin_amdgcn_update_dpp(0u, X[0], DPP_WAVE_SHL1, 0xf, 0xf, false);
    L = first_lane ? X[0] : L;       // hi half = p[1]
    R = last_lane ? X[NP - 1] : R;   // lo half = p[n-2]
    uint32_t w[NDW];
    uint32_t any = 0;
    uint32_t am1 = __builtin_amdgcn_alignbit(X[0], L, 16); // (p[-1], p[0])
    uint32_t q[4];
#pragma unroll
    for (int j = 0; j < NP; j++) {
        uint32_t xp1 = j + 1 < NP ? X[j + 1] : R;
        uint32_t ap1 = __builtin_amdgcn_alignbit(xp1, X[j], 16); // (p[2j+1], p[2j+2])
        uint32_t h = am1 + X[j] + ap1;                            // cv::blur row sum (:785)
        am1 = ap1;
        uint32_t v = a0[j] + h + 0x00040004u; // S + 4 in both lanes
        a0[j] = xp[j] + h;
        xp[j] = h;
        q[(j & 1) * 2] = __umul24(v & 0xffffu, 7282u);   // result in byte 2
        q[(j & 1) * 2 + 1] = __umul24(v >> 16, 7282u);
        if (j & 1) {
            uint32_t w01 = __builtin_amdgcn_perm(q[1], q[0], 0x0c0c0602u);
            uint32_t w23 = __builtin_amdgcn_perm(q[3], q[2], 0x06020c0cu);
            w[j >> 1] = w01 | w23;
            any |= w[j >> 1];
        }
    }
    if (emit) {
        const bool mine = any && active;
        if (__builtin_amdgcn_ballot_w64(mine)) {
            uint32_t pos = 0;
            if (cp.pairs) {
                uint32_t c = 0;
                if (mine) {
#pragma unroll
                    for (int d = 0; d < NDW; d++)
#pragma unroll
                        for (int b = 0; b < 4; b++)
                            c += (int)((w[d] >> (8 * b)) & 0xffu) > cp.thr;
                }
                pos = compact_reserve(cp, c);
            }
            if (mine) {
#pragma unroll
                for (int d = 0; d < NDW; d++) {
#pragma unroll
                    for (int b = 0; b < 4; b++) {
                        uint32_t v = (w[d] >> (8 * b)) & 0xffu;
                        if (v) {
                            atomicAdd(&lh[v], 1u);
                            if (cp.pairs)
                                compact_put(cp, pos, v, pix0 + 4 * d + b);
                        }
                    }
                }
            }
        }
        if (STORE && active) {
#pragma unroll
            for (int d = 0; d < NDW; d++)
                po[d] = w[d];
        }
    }
}

// Whole chunks (unit = block: job * nchunks + chunk) or, in list mode, the pieces {job, y0 | y1 << 16} of k3_bound_scan.
template <int NDW, bool STORE>
__global__ __launch_bounds__(64) void k3_rows(const uint8_t *__restrict__ frames, const uint8_t *__restrict__ mu,
                                              const uint8_t *__restrict__ sigma6,
                                              const abub_job *__restrict__ jobs, int W, int H,
                                              int rows_per_chunk, int nchunks, uint32_t *__restrict__ hist,
                                              uint8_t *__restrict__ img, const int32_t *__restrict__ cthr,
                                              uint32_t *pairs, uint32_t pcap, uint32_t *pcount, uint32_t slot_base,
                                              const uint2 *__restrict__ piece_list, const uint32_t *__restrict__ piece_count)
{
    constexpr int NP = 2 * NDW;
    __shared__ uint32_t lh[256];
    const int lane = threadIdx.x;
    const size_t P = (size_t)W * H;
    const int nl = W / (4 * NDW);
    const bool active = lane < nl;
    const bool first_lane = lane == 0, last_lane = lane == nl - 1;
    const int xoff = active ? lane * 4 * NDW : 0;
    const uint32_t nunits_ = piece_list ? *piece_count : gridDim.x;
    for (uint32_t ui = blockIdx.x; ui < nunits_; ui += gridDim.x) {
        int job, y0, y1;
        if (piece_list) {
            job = (int)piece_list[ui].x;
            y0 = (int)(piece_list[ui].y & 0xffffu);
            y1 = (int)(piece_list[ui].y >> 16);
        } else {
            job = (int)ui / nchunks;
            const int chunk = (int)ui - job * nchunks;
            y0 = chunk * rows_per_chunk;
            y1 = y0 + rows_per_chunk;
        }
        if (y1 > H)
            y1 = H;
        const abub_job jb = jobs[job];
        const uint8_t *f = frames + (size_t)jb.cur * P;
        const uint8_t *m = mu + (size_t)jb.model * P;
        const uint8_t *sg = sigma6 + (size_t)jb.model * P;
        const int T = y1 - y0 + 2; // input rows y0-1 .. y1 (reflected)

        lh[lane] = 0;
        lh[lane + 64] = 0;
        lh[lane + 128] = 0;
        lh[lane + 192] = 0;
        __syncthreads();

        uint32_t a0[NP], xp[NP];
#pragma unroll
        for (int j = 0; j < NP; j++)
            a0[j] = xp[j] = 0;
        uint8_t *obase = STORE ? img + (size_t)jb.out * P + xoff : nullptr;
        Compact cp;
        cp.pairs = cthr ? pairs : nullptr;
        cp.count = pcount;
        cp.cap = pcap;
        cp.slot = jb.out + slot_base;
        cp.thr = cthr ? cthr[jb.out] : 255;

        uint32_t raw[2][3][NDW]; // frame, mu, sigma6
#define K3_LOAD(SL, Y)                                                                        \
    {                                                                                         \
        const size_t o_ = (size_t)(Y) * W + xoff;                                             \
        const uint32_t *pf_ = reinterpret_cast<const uint32_t *>(f + o_);                     \
        const uint32_t *pm_ = reinterpret_cast<const uint32_t *>(m + o_);                     \
        const uint32_t *ps_ = reinterpret_cast<const uint32_t *>(sg + o_);                    \
        _Pragma("unroll") for (int d = 0; d < NDW; d++)                                       \
        {                                                                                     \
            raw[SL][0][d] = pf_[d];                                                           \
            raw[SL][1][d] = pm_[d];                                                           \
            raw[SL][2][d] = ps_[d];                                                           \
        }                                                                                     \
    }
        K3_LOAD(0, reflect101(y0 - 1, H));
        int zrun = 2; // the vertical state starts out all zero
        for (int t = 0; t < T; t += 2) {
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const int tt = t + u;
                if (tt < T) {
                    const int tn = tt + 1 < T ? tt + 1 : T - 1;
                    K3_LOAD(u ^ 1, reflect101(y0 - 1 + tn, H));
                    uint32_t HI[NP], LO[NP];
                    k3_thresholds<NDW>(raw[u][1], raw[u][2], HI, LO);
                    const int y = y0 + tt - 2;
                    k3_row<NDW, STORE>(raw[u][0], HI, LO, a0, xp, tt >= 2, active, first_lane, last_lane, lh,
                                       reinterpret_cast<uint32_t *>(obase + (ptrdiff_t)y * W), cp, (uint32_t)(y * W + xoff),
                                       zrun);
                }
            }
        }
#undef K3_LOAD
        __syncthreads();
        uint32_t *gh = hist + (size_t)jb.out * 256;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            uint32_t v = lh[lane + 64 * q];
            if (v && (lane + 64 * q))
                atomicAdd(&gh[lane + 64 * q], v);
        }
        __syncthreads(); // lh is zeroed again by the next unit
    }
}

// The K3 scan ("bound and verify", like K2's): KF consecutive jobs per wave, sharing the model rows.
// O(y,x) != 0 needs a 3x3 sum S >= 5.  With the lane's pixels in 4-column groups taken in pairs (as in k2b_row), the
// box sums that touch a pair's columns in one row are bounded by M = left group + own pair + right group (edge groups
// replicated: the reflected column lies inside them), and S <= M(y-1) + M(y) + M(y+1).  Rows where no pair reaches 5
// are proven zero with one ballot (isolated hot pixels -- sigma = 0 leaves |f - mu| of a few ADU -- stay far below).
// Suspect groups are remembered in LDS and computed exactly by the same wave afterwards (k3s_tail); where there are
// too many of them (a large bubble) the next K2B_SUB rows go to the row machine as a piece and the scan resumes behind.
#define K3S_PEND 512 /* suspects per (job, chunk) kept in LDS: the footprint of a tracked bubble fits */

template <int NDW>
struct K3ScanJob {
    static constexpr int GS = K2B_GS > NDW ? NDW : K2B_GS;
    static constexpr int NG = (NDW + GS - 1) / GS;
    uint32_t Mh[2][NG]; // M of the last two input rows, by row parity (the row loops are unrolled by two)
    uint32_t npend, hot;
    int skipTo; // output rows below this one belong to the row machine (a piece was handed over) or are not this job's
};

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

// the wave's own tail (no global list, or it is full): one lane per remembered group
template <bool COMPACT, bool STORE>
__device__ __forceinline__ void k3s_tail(const uint32_t *pend, uint32_t npend, const abub_job jb, const uint8_t *__restrict__ frames,
                                         const uint8_t *__restrict__ mu, const uint8_t *__restrict__ sigma6, int W, int H,
                                         uint32_t *__restrict__ hist, uint8_t *__restrict__ img, const Compact &cp, int lane)
{
    if (npend == 0)
        return;
    wave_lds_fence(); // orders the scan's LDS writes before the reads below
    const size_t P = (size_t)W * H;
    const uint32_t ngroups = (uint32_t)W / 4;
    const uint8_t *f = frames + (size_t)jb.cur * P;
    const uint8_t *m = mu + (size_t)jb.model * P;
    const uint8_t *sg = sigma6 + (size_t)jb.model * P;
    const uint32_t nloop = COMPACT ? (npend + 63u) & ~63u : npend;
#pragma unroll 1
    for (uint32_t e = lane; e < nloop; e += 64) {
        uint32_t packed = 0, pix0 = 0;
        if (e < npend) {
            const uint32_t code = pend[e];
            const int y = (int)(code / ngroups), x0 = (int)(code % ngroups) * 4;
            packed = k3_exact_group(f, m, sg, y, x0, W, H);
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const uint32_t v = (packed >> (8 * q)) & 0xffu;
                if (v)
                    atomicAdd(&hist[(size_t)jb.out * 256 + v], 1u);
            }
            if (STORE)
                *reinterpret_cast<uint32_t *>(img + (size_t)jb.out * P + (size_t)y * W + x0) = packed;
            pix0 = (uint32_t)(y * W + x0);
        }
        if (COMPACT) {
            uint32_t c = 0;
#pragma unroll
            for (int q = 0; q < 4; q++)
                c += (int)((packed >> (8 * q)) & 0xffu) > cp.thr;
            uint32_t pos = compact_reserve(cp, c);
#pragma unroll
            for (int q = 0; q < 4; q++)
                compact_put(cp, pos, (packed >> (8 * q)) & 0xffu, pix0 + q);
        }
    }
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
                if (STORE) // (the scan wrote this row as zeros; an aligned dword)
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

template <int NDW, int KF, bool STORE, bool COMPACT>
__global__ __launch_bounds__(64) void k3_bound_scan(const uint8_t *__restrict__ frames, const uint8_t *__restrict__ mu,
                                                    const uint8_t *__restrict__ sigma6,
                                                    const abub_job *__restrict__ jobs, int njobs, int W, int H,
                                                    int rows_per_chunk, int nchunks, uint32_t *__restrict__ hist,
                                                    uint8_t *__restrict__ img, uint2 *__restrict__ pieces,
                                                    uint32_t *__restrict__ npieces, const int32_t *__restrict__ cthr,
                                                    uint32_t *pairs, uint32_t pcap, uint32_t *pcount, uint32_t slot_base,
                                                    uint32_t budget, uint2 *__restrict__ glist,
                                                    uint32_t *__restrict__ gcount, uint32_t gcap)
{
    constexpr int NP = 2 * NDW;
    constexpr int NG = K3ScanJob<NDW>::NG, GS = K3ScanJob<NDW>::GS;
    __shared__ uint32_t pend[KF][K3S_PEND];
    const int lane = threadIdx.x;
    const int unit = blockIdx.x;
    const int grp = unit / nchunks;
    const int chunk = unit - grp * nchunks;
    const int j0 = grp * KF;
    const int k = njobs - j0 < KF ? njobs - j0 : KF; // jobs of this wave (>= 1)
    abub_job jb[KF];
#pragma unroll
    for (int t = 0; t < KF; t++)
        jb[t] = jobs[j0 + (t < k ? t : k - 1)];
    const int y0 = chunk * rows_per_chunk;
    int y1 = y0 + rows_per_chunk;
    if (y1 > H)
        y1 = H;
    bool shared = true;
#pragma unroll
    for (int t = 1; t < KF; t++)
        if (t < k && jb[t].model != jb[0].model)
            shared = false;
    if (!shared) { // jobs of different cameras in one group: their chunks go to the row machine whole
#pragma unroll
        for (int t = 0; t < KF; t++)
            if (t < k)
                k2b_hand_over(pieces, npieces, (uint32_t)(j0 + t), y0, y1, lane);
        return;
    }
    const size_t P = (size_t)W * H;
    const int nl = W / (4 * NDW);
    const bool active = lane < nl;
    const bool first_lane = lane == 0, last_lane = lane == nl - 1;
    const int xoff = active ? lane * 4 * NDW : 0;
    const uint8_t *m = mu + (size_t)jb[0].model * P;
    const uint8_t *sg = sigma6 + (size_t)jb[0].model * P;
    const uint8_t *f[KF];
    uint8_t *obase[KF];
    K3ScanJob<NDW> J[KF];
#pragma unroll
    for (int t = 0; t < KF; t++) {
        f[t] = frames + (size_t)jb[t].cur * P;
        obase[t] = STORE ? img + (size_t)jb[t].out * P + xoff : nullptr;
#pragma unroll
        for (int g = 0; g < NG; g++)
            J[t].Mh[0][g] = J[t].Mh[1][g] = 0;
        J[t].npend = J[t].hot = 0;
        J[t].skipTo = t < k ? 0 : 0x7fffffff;
    }
    const int T = y1 - y0 + 2; // input rows r = y0-1 .. y1 (reflected at the image border)
    const uint32_t ngroups = (uint32_t)W / 4;

    uint32_t raw[2][KF + 2][NDW]; // [KF] = mu, [KF + 1] = sigma6
#define K3S_LOAD(SL, Y)                                                                         \
    {                                                                                           \
        const size_t o_ = (size_t)(Y) * W + xoff;                                               \
        _Pragma("unroll") for (int t = 0; t < KF; t++)                                          \
        {                                                                                       \
            const uint32_t *pf_ = reinterpret_cast<const uint32_t *>(f[t] + o_);                \
            _Pragma("unroll") for (int d = 0; d < NDW; d++) raw[SL][t][d] = pf_[d];             \
        }                                                                                       \
        const uint32_t *pm_ = reinterpret_cast<const uint32_t *>(m + o_);                       \
        const uint32_t *ps_ = reinterpret_cast<const uint32_t *>(sg + o_);                      \
        _Pragma("unroll") for (int d = 0; d < NDW; d++)                                         \
        {                                                                                       \
            raw[SL][KF][d] = pm_[d];                                                            \
            raw[SL][KF + 1][d] = ps_[d];                                                        \
        }                                                                                       \
    }
    K3S_LOAD(0, reflect101(y0 - 1, H));
    const int Tpad = (T + 1) & ~1;
    for (int t0 = 0; t0 < Tpad; t0 += 2) {
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int tt = t0 + u; // input row r = y0 - 1 + tt; completes the bound of output row y = y0 + tt - 2
            const int tn = tt + 1 < T ? tt + 1 : T - 1;
            K3S_LOAD(u ^ 1, reflect101(y0 - 1 + tn, H));
            uint32_t HI[NP], LO[NP];
            k3_thresholds<NDW>(raw[u][KF], raw[u][KF + 1], HI, LO);
            const int y = y0 + tt - 2;
            const bool emit = tt >= 2 && tt < T;
#pragma unroll
            for (int t = 0; t < KF; t++) {
                if (t >= k)
                    continue;
                // group masses of O in this input row (u16 halves: <= 2 * 255 each)
                uint32_t mg[NDW];
#pragma unroll
                for (int d = 0; d < NDW; d++) {
                    const uint32_t f0 = widen_lo(raw[u][t][d]), f1 = widen_hi(raw[u][t][d]);
                    mg[d] = (pk_subsat(f0, HI[2 * d]) | pk_subsat(LO[2 * d], f0)) +
                            (pk_subsat(f1, HI[2 * d + 1]) | pk_subsat(LO[2 * d + 1], f1));
                }
                uint32_t mL = __builtin_amdgcn_update_dpp(0u, mg[NDW - 1], DPP_WAVE_SHR1, 0xf, 0xf, false);
                uint32_t mR = __builtin_amdgcn_update_dpp(0u, mg[0], DPP_WAVE_SHL1, 0xf, 0xf, false);
                mL = first_lane ? mg[0] : mL;
                mR = last_lane ? mg[NDW - 1] : mR;
                uint32_t B[NG];
                uint32_t worst = 0;
#pragma unroll
                for (int g = 0; g < NG; g++) {
                    const int g0 = GS * g, g1 = GS * g + GS - 1 < NDW ? GS * g + GS - 1 : NDW - 1;
                    uint32_t own = mg[g0];
#pragma unroll
                    for (int q = g0 + 1; q <= g1; q++)
                        own += mg[q];
                    const uint32_t M = (g0 ? mg[g0 - 1] : mL) + own + (g1 + 1 < NDW ? mg[g1 + 1] : mR);
                    B[g] = M + J[t].Mh[u ^ 1][g] + J[t].Mh[u][g]; // rows r, r-1, r-2 (Mh[u] still holds row r-2)
                    J[t].Mh[u][g] = M;
                    worst |= B[g];
                }
                const bool unsure = active && ((worst & 0xffffu) + (worst >> 16)) >= 5u; // (OR over-estimates: verified below)
                if (emit && y >= J[t].skipTo && __builtin_amdgcn_ballot_w64(unsure)) {
                    // ---- rare: some group of this output row cannot be proven zero ------------------------
                    unsigned long long bm[NG];
                    bool mine[NG];
                    uint32_t total = 0;
#pragma unroll
                    for (int g = 0; g < NG; g++) {
                        mine[g] = active && ((B[g] & 0xffffu) + (B[g] >> 16)) >= 5u;
                        bm[g] = __builtin_amdgcn_ballot_w64(mine[g]);
                        const int nq = GS * g + GS <= NDW ? GS : NDW - GS * g;
                        total += (uint32_t)nq * (uint32_t)__builtin_popcountll(bm[g]);
                    }
                    J[t].hot += total > 32u;
                    if (J[t].hot < 4u && J[t].npend + total > budget && total <= budget) {
                        // the LDS list is full: move it to the launch's global list and go on with an empty one
                        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                        uint32_t gb = 0;
                        if (sus_reserve(J[t].npend, glist, gcount, gcap, gb, lane)) {
                            sus_copy_out(pend[t], J[t].npend, (uint32_t)(j0 + t), glist, gb, lane);
                            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                            __builtin_amdgcn_wave_barrier();
                            J[t].npend = 0;
                        }
                    }
                    if (J[t].hot >= 4u || J[t].npend + total > budget) {
                        // dense rows (or no room anywhere): the next K2B_SUB rows go to the row machine as one
                        // piece; the scan goes on underneath and takes over again after them
                        const int ye = y + K2B_SUB < y1 ? y + K2B_SUB : y1;
                        k2b_hand_over(pieces, npieces, (uint32_t)(j0 + t), y, ye, lane);
                        J[t].skipTo = ye;
                        J[t].hot = 0;
                    } else {
                        const uint32_t code0 = (uint32_t)y * ngroups + (uint32_t)lane * NDW;
                        uint32_t base = J[t].npend;
#pragma unroll
                        for (int g = 0; g < NG; g++) {
                            const unsigned long long b = bm[g];
                            if (b) {
                                const int nq = GS * g + GS <= NDW ? GS : NDW - GS * g;
                                const uint32_t below =
                                    __builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, 0u));
                                if (mine[g]) {
#pragma unroll
                                    for (int q = 0; q < nq; q++)
                                        pend[t][base + (uint32_t)nq * below + q] = code0 + GS * g + q;
                                }
                                base += (uint32_t)nq * (uint32_t)__builtin_popcountll(b);
                            }
                        }
                        J[t].npend = base;
                    }
                }
                if (STORE && emit && y >= J[t].skipTo && active) {
                    // the scan is responsible for this row: zeros now, the tail overwrites its suspect groups
                    uint32_t *po = reinterpret_cast<uint32_t *>(obase[t] + (ptrdiff_t)y * W);
#pragma unroll
                    for (int d = 0; d < NDW; d++)
                        po[d] = 0;
                }
            }
        }
    }
#undef K3S_LOAD
    {
        uint32_t tot = 0;
#pragma unroll
        for (int t = 0; t < KF; t++)
            tot += t < k ? J[t].npend : 0u;
        if (tot == 0)
            return;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        uint32_t gb = 0;
        if (sus_reserve(tot, glist, gcount, gcap, gb, lane)) { // the whole wave's suspects in one reservation
#pragma unroll
            for (int t = 0; t < KF; t++)
                if (t < k) {
                    sus_copy_out(pend[t], J[t].npend, (uint32_t)(j0 + t), glist, gb, lane);
                    gb += J[t].npend;
                }
            return;
        }
    }
#pragma unroll
    for (int t = 0; t < KF; t++) {
        if (t < k) {
            Compact cp;
            cp.pairs = COMPACT ? pairs : nullptr;
            cp.count = pcount;
            cp.cap = pcap;
            cp.slot = jb[t].out + slot_base;
            cp.thr = COMPACT ? cthr[jb[t].out] : 255;
            k3s_tail<COMPACT, STORE>(pend[t], J[t].npend, jb[t], frames, mu, sigma6, W, H, hist, img, cp, lane);
        }
    }
}

static int k3_scan_enabled()
{
    static int on = -1;
    if (on < 0) {
        const char *e = getenv("ABUB_K3_SCAN"); // 0: the row machine on every row (no zero scan)
        on = e ? atoi(e) : 1;
    }
    return on;
}

template <int NDW>
static int launch_k3_rows(const uint8_t *frames, const uint8_t *mu, const uint8_t *sigma6, const abub_job *jobs,
                          int njobs, int W, int H, int R, int nchunks, uint32_t *hist, uint8_t *img,
                          const CompactArgs &ca, hipStream_t st)
{
#define K3R_LAUNCH(ST, GRID, PL, PC)                                                                                 \
    hipLaunchKernelGGL((k3_rows<NDW, ST>), dim3(GRID), dim3(64), 0, st, frames, mu, sigma6, jobs, W, H, R, nchunks,   \
                       hist, img, ca.cthr, ca.pairs, ca.cap, ca.count, ca.slot_base, PL, PC)
    if (k3_scan_enabled() && H < 65536 && (size_t)H * (size_t)(W / 4) < ((size_t)1 << 32)) {
        constexpr int KF = NDW <= 5 ? 5 : (NDW <= 7 ? 4 : 3); // jobs per scanning wave (register budget)
        const size_t nunits = (size_t)njobs * nchunks;
        const size_t cap = nunits * (size_t)((R + K2B_SUB - 1) / K2B_SUB); // handed-over pieces, worst case
        // the global suspect list (see sus_tail_list): room for 2048 groups per frame on average -- the footprint of a
        // tracked bubble is a few hundred to a thousand groups -- within 64 K .. 16 M entries
        static int k3list = -1;
        if (k3list < 0) {
            const char *e = getenv("ABUB_K3_LIST"); // 0: suspects are evaluated by the scanning waves themselves
            k3list = e ? atoi(e) : 1;
        }
        size_t gcap = (size_t)njobs * 2048;
        gcap = gcap < ((size_t)1 << 16) ? ((size_t)1 << 16) : (gcap > ((size_t)1 << 24) ? ((size_t)1 << 24) : gcap);
        if (!k3list)
            gcap = 0;
        const size_t piecesBytes = (cap * sizeof(uint2) + 255) & ~(size_t)255;
        std::unique_lock<std::mutex> hold;
        uint8_t *scr = (uint8_t *)k2_scratch(st, 256 + piecesBytes + gcap * sizeof(uint2) + 256, hold);
        if (!scr)
            return set_err(ABUB_E_HIP, "abub_posttrig_dev: scratch allocation failed");
        uint32_t *counter = (uint32_t *)scr;   // [0] = handed-over pieces, [32] = entries of the global suspect list
        uint32_t *gcount = counter + 32;
        uint2 *pieces = (uint2 *)(scr + 256);
        uint2 *glist = gcap ? (uint2 *)(scr + 256 + piecesBytes) : nullptr;
        HIPCHK(hipMemsetAsync(counter, 0, 256, st));
        const dim3 sgrid((unsigned)((size_t)((njobs + KF - 1) / KF) * nchunks));
        static int k3b = -1;
        if (k3b < 0) {
            const char *e = getenv("ABUB_K3_BUDGET"); // suspects a (job, chunk) may remember before it hands a piece over
            k3b = e ? atoi(e) : K3S_PEND;
            if (k3b < 0 || k3b > K3S_PEND)
                k3b = K3S_PEND;
        }
        const uint32_t k3budget = (uint32_t)k3b;
#define K3S_LAUNCH(ST, CO)                                                                                          \
    hipLaunchKernelGGL((k3_bound_scan<NDW, KF, ST, CO>), sgrid, dim3(64), 0, st, frames, mu, sigma6, jobs, njobs, W, \
                       H, R, nchunks, hist, img, pieces, counter, ca.cthr, ca.pairs, ca.cap, ca.count, ca.slot_base, \
                       k3budget, glist, gcount, (uint32_t)gcap);                                                    \
    if (glist)                                                                                                      \
    hipLaunchKernelGGL((sus_tail_list<3, CO, ST>), dim3(tgrid), dim3(256), 0, st, frames, mu, sigma6, jobs, W, H, hist, \
                       img, glist, gcount, (uint32_t)gcap, ca.cthr, ca.pairs, ca.cap, ca.count, ca.slot_base)
        const unsigned tgrid = (unsigned)((gcap + 256 * SUSL_UB - 1) / (256 * SUSL_UB) < 2048 ? (gcap + 256 * SUSL_UB - 1) / (256 * SUSL_UB) : 2048);
        if (ca.cthr) {
            if (img) {
                K3S_LAUNCH(true, true);
            } else {
                K3S_LAUNCH(false, true);
            }
        } else {
            if (img) {
                K3S_LAUNCH(true, false);
            } else {
                K3S_LAUNCH(false, false);
            }
        }
#undef K3S_LAUNCH
        const unsigned g = (unsigned)(cap < 8192 ? cap : 8192);
        if (img)
            K3R_LAUNCH(true, g, pieces, counter);
        else
            K3R_LAUNCH(false, g, pieces, counter);
        return ABUB_OK;
    }
    const unsigned grid = (unsigned)njobs * nchunks;
    if (img)
        K3R_LAUNCH(true, grid, (const uint2 *)nullptr, (const uint32_t *)nullptr);
    else
        K3R_LAUNCH(false, grid, (const uint2 *)nullptr, (const uint32_t *)nullptr);
#undef K3R_LAUNCH
    return ABUB_OK;
}

static int posttrig_impl(const uint8_t *frames, const uint8_t *mu, const uint8_t *sigma6, const abub_job *jobs,
                         int njobs, int W, int H, uint32_t *hist, uint8_t *img, const CompactArgs &ca, void *stream);

extern "C" int abub_posttrig_dev(const uint8_t *frames, const uint8_t *mu, const uint8_t *sigma6,
                                 const abub_job *jobs, int njobs, int W, int H, uint32_t *hist,
                                 uint8_t *img, void *stream)
{
    CompactArgs ca = {nullptr, nullptr, 0, nullptr, 0};
    return posttrig_impl(frames, mu, sigma6, jobs, njobs, W, H, hist, img, ca, stream);
}

extern "C" int abub_posttrig_compact_dev(const uint8_t *frames, const uint8_t *mu, const uint8_t *sigma6,
                                         const abub_job *jobs, int njobs, int W, int H, uint32_t *hist,
                                         uint8_t *img, const int32_t *cthr, uint32_t *pairs, uint32_t cap,
                                         uint32_t *count, uint32_t slot_base, void *stream)
{
    if (!cthr || !pairs || !count || cap == 0)
        return set_err(ABUB_E_INVALID, "abub_posttrig_compact_dev: bad arguments");
    CompactArgs ca = {cthr, pairs, cap, count, slot_base};
    return posttrig_impl(frames, mu, sigma6, jobs, njobs, W, H, hist, img, ca, stream);
}

static int posttrig_impl(const uint8_t *frames, const uint8_t *mu, const uint8_t *sigma6, const abub_job *jobs,
                         int njobs, int W, int H, uint32_t *hist, uint8_t *img, const CompactArgs &ca, void *stream)
{
    if (!frames || !mu || !sigma6 || !jobs || !hist || W <= 0 || H <= 0 || njobs < 0)
        return set_err(ABUB_E_INVALID, "abub_posttrig_dev: bad arguments");
    if (njobs == 0)
        return ABUB_OK;
    hipStream_t st = (hipStream_t)stream;
    HIPCHK(hipMemsetAsync(hist, 0, (size_t)njobs * 256 * sizeof(uint32_t), st));
    int ndw = pick_ndw(W);
    if (ndw) {
        // waves of the launch = (groups of jobs one wave serves) x chunks: enough of them (>= ~8k, several rounds of the
        // chip's wave slots) that the exact tails of early waves run under the scans of later ones
        const int kf = k3_scan_enabled() ? (ndw <= 5 ? 5 : (ndw <= 7 ? 4 : 3)) : 1;
        const long long ngrp = (njobs + kf - 1) / kf;
        int nch = 8;
        if (ngrp * nch < 8192)
            nch = (int)((8192 + ngrp - 1) / ngrp);
        static int k3chunks = -1;
        if (k3chunks < 0) {
            const char *e = getenv("ABUB_K3_CHUNKS"); // tuning knob: chunks per frame (0 = automatic)
            k3chunks = e ? atoi(e) : 0;
        }
        if (k3chunks > 0)
            nch = k3chunks;
        nch = (nch + 7) / 8 * 8;
        int R = (H + nch - 1) / nch;
        if (R < 16)
            R = 16;
        int nchunks = (H + R - 1) / R;
        int rc3 = ABUB_OK;
        switch (ndw) {
        case 1: rc3 = launch_k3_rows<1>(frames, mu, sigma6, jobs, njobs, W, H, R, nchunks, hist, img, ca, st); break;
        case 2: rc3 = launch_k3_rows<2>(frames, mu, sigma6, jobs, njobs, W, H, R, nchunks, hist, img, ca, st); break;
        case 3: rc3 = launch_k3_rows<3>(frames, mu, sigma6, jobs, njobs, W, H, R, nchunks, hist, img, ca, st); break;
        case 4: rc3 = launch_k3_rows<4>(frames, mu, sigma6, jobs, njobs, W, H, R, nchunks, hist, img, ca, st); break;
        case 5: rc3 = launch_k3_rows<5>(frames, mu, sigma6, jobs, njobs, W, H, R, nchunks, hist, img, ca, st); break;
        case 6: rc3 = launch_k3_rows<6>(frames, mu, sigma6, jobs, njobs, W, H, R, nchunks, hist, img, ca, st); break;
        case 7: rc3 = launch_k3_rows<7>(frames, mu, sigma6, jobs, njobs, W, H, R, nchunks, hist, img, ca, st); break;
        default: rc3 = launch_k3_rows<8>(frames, mu, sigma6, jobs, njobs, W, H, R, nchunks, hist, img, ca, st); break;
        }
        if (rc3 != ABUB_OK)
            return rc3;
    } else {
        if (ca.cthr)
            return set_err(ABUB_E_INVALID, "fused compaction needs the fast path (W % 4 == 0, W <= 2048)");
        if (njobs > 65535)
            return set_err(ABUB_E_INVALID, "abub_posttrig_dev: njobs > 65535");
        dim3 grid((W + G_TW - 1) / G_TW, (H + G_TH - 1) / G_TH, njobs), block(256);
        hipLaunchKernelGGL(k3_generic, grid, block, 0, st, frames, mu, sigma6, jobs, W, H, hist, img);
    }
    HIPCHK(hipGetLastError());
    hipLaunchKernelGGL(k_hist_bin0, dim3(njobs), dim3(64), 0, st, hist, (uint32_t)((size_t)W * H));
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
