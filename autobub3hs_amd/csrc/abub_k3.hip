// abub_k3.hip -- K3: post-trigger image O = blur3x3(sat(|f - mu| - 6 sigma)) + 256-bin histogram (L3Localizer.cpp:779-787)
// for gfx950 and its stateless C-ABI launchers (include/abub_hip.h).
#include "abub_dev.hpp"

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
// O(y,x) != 0 needs a 3x3 sum S >= 5.  Group masses come from v_sad_u8 on packed bytes (two per frame dword, the packed
// thresholds shared by the wave's KF frames; see the loop).  With the lane's pixels in 4-column groups taken in pairs (as in k2b_row), the
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

// the wave's own tail (no global list, or it is full): one lane per remembered group
template <bool COMPACT>
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
            if (img) // (the launcher cleared the image)
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

template <int NDW, int KF, bool COMPACT>
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
    K3ScanJob<NDW> J[KF];
#pragma unroll
    for (int t = 0; t < KF; t++) {
        f[t] = frames + (size_t)jb[t].cur * P;
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
            // thresholds of this model row as packed bytes: HI = min(mu + sigma6, 255), LO = sat(mu - sigma6); then
            //     2 * sum over a 4-pixel group of O = SAD(f, HI) + SAD(f, LO) - SAD(HI, LO)        (f inside [LO, HI]: the
            // first two add up to HI - LO) -- exact, two v_sad_u8 per frame dword, the thresholds shared by the KF frames
            uint32_t HI8[NDW], LO8[NDW], nS[NDW];
#pragma unroll
            for (int d = 0; d < NDW; d++) {
                const uint32_t m0 = widen8_lo(raw[u][KF][d]), m1 = widen8_hi(raw[u][KF][d]);
                const uint32_t s0 = widen8_lo(raw[u][KF + 1][d]), s1 = widen8_hi(raw[u][KF + 1][d]);
                HI8[d] = pack8(pk_addsat(m0, s0), pk_addsat(m1, s1));
                LO8[d] = pack8(pk_subsat(m0, s0), pk_subsat(m1, s1));
                nS[d] = 0u - __builtin_amdgcn_sad_u8(HI8[d], LO8[d], 0u);
            }
            const int y = y0 + tt - 2;
            const bool emit = tt >= 2 && tt < T;
#pragma unroll
            for (int t = 0; t < KF; t++) {
                if (t >= k)
                    continue;
                // twice the group masses of O in this input row (plain u32: <= 8 * 255 each)
                uint32_t mg[NDW];
#pragma unroll
                for (int d = 0; d < NDW; d++)
                    mg[d] = __builtin_amdgcn_sad_u8(raw[u][t][d], LO8[d], __builtin_amdgcn_sad_u8(raw[u][t][d], HI8[d], nS[d]));
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
                    worst = B[g] > worst ? B[g] : worst;
                }
                const bool unsure = active && worst >= 10u; // S >= 5 <=> twice the mass bound >= 10
                if (emit && y >= J[t].skipTo && __builtin_amdgcn_ballot_w64(unsure)) {
                    // ---- rare: some group of this output row cannot be proven zero ------------------------
                    unsigned long long bm[NG];
                    bool mine[NG];
                    uint32_t total = 0;
#pragma unroll
                    for (int g = 0; g < NG; g++) {
                        mine[g] = active && B[g] >= 10u;
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
            k3s_tail<COMPACT>(pend[t], J[t].npend, jb[t], frames, mu, sigma6, W, H, hist, img, cp, lane);
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
    hipLaunchKernelGGL((k3_bound_scan<NDW, KF, CO>), sgrid, dim3(64), 0, st, frames, mu, sigma6, jobs, njobs, W, H, R,  \
                       nchunks, hist, img, pieces, counter, ca.cthr, ca.pairs, ca.cap, ca.count, ca.slot_base,      \
                       k3budget, glist, gcount, (uint32_t)gcap);                                                    \
    if (glist)                                                                                                      \
    hipLaunchKernelGGL((sus_tail_list<3, CO, ST>), dim3(tgrid), dim3(256), 0, st, frames, mu, sigma6, jobs, W, H, hist, \
                       img, glist, gcount, (uint32_t)gcap, ca.cthr, ca.pairs, ca.cap, ca.count, ca.slot_base)
        const unsigned tgrid = (unsigned)((gcap + 256 * SUSL_UB - 1) / (256 * SUSL_UB) < 2048 ? (gcap + 256 * SUSL_UB - 1) / (256 * SUSL_UB) : 2048);
        // store mode: the images are cleared first (a pure write stream at the chip's fill rate); only the exact tails
        // and the row machine write pixels afterwards.  (Contract, as for `hist`: output slots 0 .. njobs - 1.)
        if (img)
            HIPCHK(hipMemsetAsync(img, 0, (size_t)njobs * (size_t)W * (size_t)H, st));
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
