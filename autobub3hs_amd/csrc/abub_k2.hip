// abub_k2.hip -- K2: fused AnalyzerUnit::ProcessFrame + 256-bin histogram (gfx950, wave64, no MFMA) and its stateless
// C-ABI launchers (include/abub_hip.h).  Kernel inventory: DESIGN.md section "Kernels".
//   k2_rows<NDW,STORE,PF,COMPACT>  register-rolling row machine (optional stored image / fused candidate list; list mode
//                                  for handed-over rows)
//   k2_sad_chain / k2_bound_scan   bound-and-verify: proves rows of D zero from a bound, lists the rest
//   k2_generic                     same arithmetic, any size / ROI, LDS tile (also the ROI overload)
#include "abub_dev.hpp"

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
                                              const uint32_t *__restrict__ unit_count,
                                              const uint8_t *__restrict__ want = nullptr)
{
    constexpr int NP = 2 * NDW; // u16-pair registers per plane per lane
    __shared__ uint32_t lh[256];

    const int lane = threadIdx.x;
    // list mode (bound-and-verify hand-over): the grid strides over the listed units; otherwise unit = block
    const uint32_t nunits_ = unit_list ? *unit_count : gridDim.x;
    for (uint32_t ui = blockIdx.x; ui < nunits_; ui += gridDim.x) {
    const int unit = unit_list ? (int)unit_list[ui].x : (int)ui;
    const int job = unit / nchunks;
    if (want && !want[job]) // deferred pieces: only the jobs the trigger search actually reached (workgroup-uniform)
        continue;
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
//   k2_bound_scan / k2_sad_chain
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
    // the lane's piece of segment s as an UNSIGNED 32-bit byte offset: row pointer (wave-uniform, SGPRs) + zext(offset) lets
    // the loads take the scalar-base + VGPR-offset form instead of a 64-bit VALU add per load
    __device__ __forceinline__ uint32_t uoff(int s, int lane) const { return (uint32_t)byteoff(s, lane); }
    // byte offset of the lane's piece of segment s in a row (idle lanes shadow lane 0: valid address, results unused)
    __device__ __forceinline__ int byteoff(int s, int lane) const { return 4 * (gbase[s] + segK(s) * (lane < nl ? lane : 0)); }
    __device__ __forceinline__ void load(uint32_t (&r)[NDW], const uint8_t *__restrict__ row, int lane) const
    {
#pragma unroll
        for (int s = 0; s < NSEG; s++) {
            const uint32_t *p = reinterpret_cast<const uint32_t *>(row + (size_t)uoff(s, lane));
#pragma unroll
            for (int d = 0; d < segK(s); d++)
                r[segD0(s) + d] = p[d];
        }
    }
    // the same pieces through buffer addressing: `rs` describes the slab from a wave-uniform base, `soff` (SGPR) is the
    // frame's and row's byte offset from it, the lane's piece offset is the only VGPR -- no VALU address arithmetic at all
    template <int N>
    static __device__ __forceinline__ void loadn(uint32_t *r, __amdgpu_buffer_rsrc_t rs, uint32_t voff, uint32_t soff)
    {
        if constexpr (N >= 4) {
            const auto v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)voff, (int)soff, 0);
            r[0] = v[0], r[1] = v[1], r[2] = v[2], r[3] = v[3];
            if constexpr (N > 4)
                loadn<N - 4>(r + 4, rs, voff + 16u, soff);
        } else if constexpr (N == 3) {
            const auto v = __builtin_amdgcn_raw_buffer_load_b96(rs, (int)voff, (int)soff, 0);
            r[0] = v[0], r[1] = v[1], r[2] = v[2];
        } else if constexpr (N == 2) {
            const auto v = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)voff, (int)soff, 0);
            r[0] = v[0], r[1] = v[1];
        } else {
            r[0] = __builtin_amdgcn_raw_buffer_load_b32(rs, (int)voff, (int)soff, 0);
        }
    }
    __device__ __forceinline__ void loadb(uint32_t (&r)[NDW], __amdgpu_buffer_rsrc_t rs, uint32_t soff, int lane) const
    {
        if constexpr (!SPLIT) {
            loadn<NDW>(&r[0], rs, uoff(0, lane), soff);
        } else {
            loadn<4>(&r[0], rs, uoff(0, lane), soff);
            loadn<NDW - 4>(&r[4], rs, uoff(NSEG - 1, lane), soff);
        }
    }
};

template <int NDW, bool CASC = (NDW <= 5), bool PLAINM = false>
struct K2BoundJob { // per-job state of the bound recurrence and of its suspect list (all wave-uniform but b*/Mprev)
    // The recurrence runs on PAIRS of 4-pixel groups (8 columns, the last one alone when NDW is odd): the taps of a
    // column still reach at most the neighbouring 4-pixel group on either side, so M = left group + own pair + right
    // group bounds every column of the pair; suspects are listed as their 4-pixel groups.
    static constexpr int GS = K2B_GS > NDW ? NDW : K2B_GS; // 4-pixel groups per recurrence group
    static constexpr int NG = (NDW + GS - 1) / GS;
    // vertical 1-4-6-4-1 of the group masses as four cascaded two-tap sums (binomial = (1 + z^-1)^4): per group four
    // plain 32-bit adds and no shift / multiply; P[k][parity] = output of stage k at the previous row of that parity
    // (the row loops are unrolled by two, so nothing is ever copied).  Wide rows (NDW >= 6) keep the state in four
    // in-place accumulators instead (P[k][0]): 16 instead of 32 registers per job there, which is a wave of occupancy.
    static constexpr bool CASCADE = CASC;
    // mass format: packed halves (mass of columns 0,2 | mass of columns 1,3) of the u16-pair arithmetic, a row is unsure
    // when lo + hi > 21 -- or PLAIN = TWICE the group's mass as one u32 (the v_sad_u8 scan), unsure when > 42
    static constexpr bool PLAIN = PLAINM;
    static __device__ __forceinline__ bool over(uint32_t b) { return PLAINM ? b > 42u : ((b & 0xffffu) + (b >> 16)) > 21u; }
    uint32_t P[4][CASC ? 2 : 1][NG];
    uint32_t Mp[PLAINM ? 2 : 1][NG]; // PLAIN, in-place form: M of the previous row, by row parity (nothing is copied)
    uint32_t npend, hot, jidx;
    uint32_t mark; // npend when the current streak of dense rows began (hot = its length)
    int handover; // < 0: scanning; >= 0: first output row left to the row machine (or "nothing to do")
};

// one input row of one job: group masses m[] -> bound of output row y; suspects go to the job's LDS list.
// Everything that steers control flow is read through SGPRs (ballots, s_bcnt1), so the scan loops compile to scalar
// branches.
template <int NDW, bool SPLIT, typename JOB>
__device__ __forceinline__ void k2b_row(JOB &J, const int par, uint32_t (&m)[NDW], bool emit, int y,
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
    constexpr int NG = JOB::NG;
    constexpr int GS = JOB::GS;
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
        if constexpr (JOB::CASCADE) {
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
            J.P[1][0][g] = JOB::PLAIN ? __umul24(M, 6u) + J.P[2][0][g] : pk_madk<6>(M, J.P[2][0][g]); // (M < 2^24)
            if constexpr (JOB::PLAIN) {
                J.P[2][0][g] = J.Mp[par ^ 1][g] + M4;
                J.Mp[par][g] = M;
            } else {
                J.P[2][0][g] = J.P[3][0][g] + M4;
                J.P[3][0][g] = M;
            }
        }
        worst = JOB::PLAIN ? (B[g] > worst ? B[g] : worst) : (worst | B[g]);
    }
    const bool unsure = act && JOB::over(worst); // 6 * mass < 128 <=> mass <= 21 (packed: lo + hi; plain: twice the mass <= 42)
    if (!(emit && __builtin_amdgcn_ballot_w64(unsure))) {
        J.hot = 0; // (a quiet row ends a streak of dense ones)
        return;
    }
    // ---- rare: some group of this row cannot be proven zero -----------------------------------------------
    unsigned long long bm[NG]; // lanes whose recurrence group g (GS 4-pixel groups) is suspect
    bool mine[NG];
    uint32_t total = 0;
#pragma unroll
    for (int g = 0; g < NG; g++) {
        mine[g] = act && JOB::over(B[g]);
        bm[g] = __builtin_amdgcn_ballot_w64(mine[g]);
        const int nq = GS * g + GS <= NDW ? GS : NDW - GS * g; // 4-pixel groups of this recurrence group
        total += (uint32_t)nq * (uint32_t)__builtin_popcountll(bm[g]);
    }
    // One row of the row machine costs about as much as 30 exact groups.  Four dense rows in a row: the chunk is handed
    // over FROM THE FIRST OF THEM, and what the streak put on the list is dropped again -- a dense frame then costs its
    // scanning wave a few rows and no exact tail at all (round 2 kept the streak's groups: ~450 exact groups per chunk of
    // every dense frame, whether anybody ever looked at that frame or not).
    if (total > 32u) {
        if (J.hot == 0)
            J.mark = J.npend;
        ++J.hot;
    } else
        J.hot = 0;
    if (J.hot >= 4u) {
        J.handover = y - 3;
        J.npend = J.mark;
        return;
    }
    // (A full LDS list means a chunk with a lot of structure: for K2 the row machine is the cheaper way through such
    // rows -- an exact group costs 45 window loads and two 5x5 sums --, so the list is NOT flushed to the global suspect
    // list to make room, as K3 does; measured: flushing made the tail kernel 2.4x longer than the pieces it saved.)
    if (J.npend + total > budget) {
        J.handover = y; // the LDS list is full: the rest of the chunk goes to the row machine
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

// Store mode of the bound-and-verify pass: the launcher clears the output images first (hipMemsetAsync: a pure write
// stream at the chip's fill rate, 6.5 TB/s), the scan then runs exactly as in trigger-only mode, the tails write the
// dwords of their groups and the row machine writes the handed-over rows.  (Round 2 had the scanning waves write the
// proven rows as zeros themselves: 4.6 ms per 8000 jobs at 1280x1024 against 1.6 ms fill + 2.2 ms scan.)
// The wave's own tail (no global list, or it is full): D for the four pixels of every remembered group, one lane per
// group; histogram by global atomics (rare), optional store / candidates.
template <bool COMPACT>
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
            if (diff) // (the launcher cleared the image; x0 is a multiple of 4 and W % 4 == 0: an aligned dword)
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

template <int NDW, bool COMPACT>
__global__ __launch_bounds__(64) void k2_bound_scan(const uint8_t *__restrict__ frames,
                                                    const uint8_t *__restrict__ sigma6,
                                                    const abub_job *__restrict__ jobs, int W, int H,
                                                    int rows_per_chunk, int nchunks, uint32_t budget,
                                                    uint2 *__restrict__ units, uint32_t *__restrict__ nunits,
                                                    uint32_t *__restrict__ hist, uint8_t *__restrict__ diff,
                                                    const int32_t *__restrict__ cthr, uint32_t *pairs, uint32_t pcap,
                                                    uint32_t *pcount, uint32_t slot_base, SusList gl,
                                                    uint8_t *__restrict__ incomplete)
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
            J.P[q][0][g] = J.P[q][K2BoundJob<NDW>::CASCADE ? 1 : 0][g] = 0;
    J.npend = J.hot = 0;
    J.jidx = (uint32_t)job;
    J.handover = -1;

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
            }
        }
    }
    if (J.handover >= 0)
        k2b_hand_over(units, nunits, (uint32_t)unit, J.handover, y1, lane, incomplete, (uint32_t)job);
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
    k2b_tail<COMPACT>(pend, J.npend, jb, frames, sigma6, W, H, hist, diff, cp, lane);
}

#define K2C_MAXW 8 /* waves per workgroup of the chained scan, at most */
#ifndef K2C_DEFAULT_WG
#define K2C_DEFAULT_WG 1
#endif
#ifndef K2C_DEFAULT_SYNC
#define K2C_DEFAULT_SYNC 0
#endif
// ------------------------------------------------------------------------------------------------
// k2_sad_chain: the bound scan for job lists with the trigger search's structure -- blocks of `L` consecutive jobs in
// which job q takes the cur frame of job q - S as its ref (FindTriggerFrame: S = 2, or 1 for small training sets).
// One wave serves up to K jobs of one such chain for one chunk: every frame row is loaded once and serves the job that
// has it as cur and the job that has it as ref, sigma6 once for all -- (K + 2) / K row loads per job instead of 3.
// The chain property is only a hint: the wave checks it on the job records and hands units it cannot chain to the row
// machine, so any job list gives the same histograms as k2_bound_scan / k2_rows.
//
// Group masses come from v_sad_u8, not from u16-pair arithmetic:
//   X = sat(c - r - s) + sat(r - c - s) is symmetric in its two frames, and with HI = min(r + s, 255), LO = sat(r - s)
//   (packed bytes) the mass of a 4-pixel group is
//       2 * sum_b X_b = SAD(c, HI) + SAD(c, LO) - SAD(HI, LO)          (c inside [LO, HI]: the first two add up to HI - LO)
//   -- exact, two v_sad_u8 per dword and job once HI / LO / -SAD(HI, LO) of ONE of the job's frames are in registers.
//   In a chain f0 f1 .. fK (job t = (cur f[t+1], ref f[t])) the odd frames f1, f3, .. are the ones that get HI / LO: job
//   t - 1 meets f[t] as its cur, job t as its ref, so every HI / LO pair serves two jobs and the even frames are never
//   widened at all.  Per dword and job: 7 quarter-rate VALU instructions (2 SAD + half of: 2 widen, 2 add-sat, 2 sub-sat,
//   2 pack, 1 SAD, + sigma6's widening) where the u16-pair form of round 2 needed 7.3 + 5.7 full-rate ones (4.6 instead of
//   5.7 VALU instructions per pixel in all); the bound, the suspect lists, hand-over and tails are k2b_row's, fed with
//   twice the mass as a plain u32 (K2BoundJob<.., PLAIN>).  Four in-place accumulators per recurrence group (12 registers
//   per job at NDW = 5) leave room for K = 4 jobs per wave.
//
// Workgroups of NW > 1 waves (option "wg", off by default): the waves are consecutive segments of ONE chain on the same
// chunk, sharing nothing but progress words in LDS (option "sync": a wave more than `sync` row steps ahead of the slowest
// one naps), so that the frame two neighbouring segments share is asked for by both within microseconds and the second
// request hits the XCD's L2.  Measured: HBM traffic per job falls from 1.24 to 1.12 W*H, the pass
// gets 2 - 20 % SLOWER -- the pass is bound by neither bytes nor VALU issue alone.
// ------------------------------------------------------------------------------------------------
template <int NDW, int K, bool SPLIT, int PF = 1>
__global__ __launch_bounds__(64 * K2C_MAXW) void k2_sad_chain(const uint8_t *__restrict__ frames,
                                                   const uint8_t *__restrict__ sigma6,
                                                   const abub_job *__restrict__ jobs, int L, int S, int nwgs, int W,
                                                   int H, int rows_per_chunk, int nchunks, uint32_t budget,
                                                   uint2 *__restrict__ units, uint32_t *__restrict__ nunits,
                                                   uint32_t *__restrict__ hist, uint8_t *__restrict__ diff, SusList gl,
                                                   int syncD, uint8_t *__restrict__ incomplete)
{
    using Job = K2BoundJob<NDW, false, true>;
    // dynamic LDS: [NW][K][K2B_PEND] suspect lists (one set per wave), then [NW] progress words.  Progress word of a wave =
    // the row step it is about to start; 0x7fffffff once it no longer loads rows.  No barrier anywhere: the slowest wave
    // never waits, a finished wave counts as infinitely far ahead.
    extern __shared__ uint32_t k2c_lds[];
    const int NW = (int)(blockDim.x >> 6);
    const int lane = (int)(threadIdx.x & 63u);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    uint32_t(*pend)[K2B_PEND] = reinterpret_cast<uint32_t(*)[K2B_PEND]>(k2c_lds + (size_t)wave * K * K2B_PEND);
    // (relaxed workgroup-scope LDS atomics, NOT volatile accesses: those go through a generic pointer and compile to FLAT
    // instructions, and one pending FLAT operation turns every vmcnt wait of the row loop into vmcnt(0))
    const int progBase = NW * K * K2B_PEND;
#define K2C_PUBLISH(V)                                                                                         \
    {                                                                                                          \
        if (NW > 1 && lane == 0)                                                                               \
            __hip_atomic_store(&k2c_lds[progBase + wave], (uint32_t)(V), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); \
    }
    const int chunk = blockIdx.x % nchunks;
    const int bs = blockIdx.x / nchunks;
    const int blk = bs / nwgs;
    int wg = bs - blk * nwgs;
    int r = 0, nr = 0, ns = 0;
    for (r = 0; r < S; r++) {
        nr = (L - r + S - 1) / S;
        ns = (nr + K - 1) / K;
        const int nw = (ns + NW - 1) / NW;
        if (wg < nw)
            break;
        wg -= nw;
    }
    const int slot = wg * NW + wave;
    if (r >= S || slot >= ns) {
        K2C_PUBLISH(0x7fffffff);
        return;
    }
    const int k = nr - slot * K < K ? nr - slot * K : K; // jobs of this wave (>= 1)
    Job J[K];
    abub_job jb[K]; // (only alive up to the frame offsets below: the tails read their job records again)
    const uint32_t jidx0 = (uint32_t)(blk * L + r + S * slot * K); // job t of this wave = jidx0 + S * t
#pragma unroll
    for (int t = 0; t < K; t++)
        jb[t] = jobs[jidx0 + (uint32_t)(S * (t < k ? t : k - 1))];
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
                k2b_hand_over(units, nunits, (jidx0 + (uint32_t)(S * t)) * (uint32_t)nchunks + (uint32_t)chunk, y0, y1, lane, incomplete,
                              jidx0 + (uint32_t)(S * t));
        K2C_PUBLISH(0x7fffffff);
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
    const int T = y1 - y0 + 4;
    const uint32_t ngroups = (uint32_t)W / 4;
#pragma unroll
    for (int t = 0; t < K; t++) {
#pragma unroll
        for (int g = 0; g < Job::NG; g++)
#pragma unroll
            for (int q = 0; q < 4; q++)
                J[t].P[q][0][g] = 0;
#pragma unroll
        for (int g = 0; g < Job::NG; g++)
            J[t].Mp[0][g] = J[t].Mp[1][g] = 0;
        J[t].npend = J[t].hot = 0;
        J[t].handover = t < k ? -1 : 0x7fffffff; // (jobs beyond k do nothing and report nothing)
    }
    // rows are fetched PF steps ahead through a ring of PF + 1 register sets ([K + 1] = sigma6); the loop is unrolled by
    // U = lcm(PF + 1, 2) so that ring slots and the parity of the recurrence state are compile-time registers.
    // (Plain global loads: buffer addressing -- scalar base and row offset, no VALU address arithmetic -- was measured
    // 1 - 3 % SLOWER on this pass.)
    constexpr int RING = PF + 1, U = (RING % 2 == 0) ? RING : 2 * RING;
    uint32_t raw[RING][K + 2][NDW];
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
        if (NW > 1) { // publish the row step, nap while too far ahead of the workgroup's slowest wave
            K2C_PUBLISH(t0);
            if (syncD > 0) {
                for (;;) {
                    int mn = 0x7fffffff;
                    for (int w = 0; w < NW; w++) {
                        const int pw_ = (int)__hip_atomic_load(&k2c_lds[progBase + w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        mn = pw_ < mn ? pw_ : mn;
                    }
                    if (t0 <= __builtin_amdgcn_readfirstlane(mn) + syncD)
                        break;
                    __builtin_amdgcn_s_sleep(8);
                }
            }
        }
#pragma unroll
        for (int uu = 0; uu < U; uu++) {
            const int tt = t0 + uu;
            const int tn = tt + PF < T ? tt + PF : T - 1;
            K2C_LOAD((uu + PF) % RING, reflect101(y0 - 2 + tn, H));
            const int u = uu % RING; // ring slot of this step's rows
            const int par = uu & 1;  // parity of the recurrence state
            const bool emit = tt >= 4 && tt < T;
            const int y = y0 + tt - 4;
            uint32_t s8[2 * NDW]; // sigma6 in the high bytes of u16 lanes
#pragma unroll
            for (int d = 0; d < NDW; d++) {
                s8[2 * d] = widen8_lo(raw[u][K + 1][d]);
                s8[2 * d + 1] = widen8_hi(raw[u][K + 1][d]);
            }
#pragma unroll
            for (int h = 1; h <= K; h += 2) { // frame h gets HI / LO: it is the cur of job h - 1 and the ref of job h
                const bool ja = J[h - 1].handover < 0, jc = h < K && J[h < K ? h : K - 1].handover < 0;
                if (!(ja || jc))
                    continue;
                uint32_t HI[NDW], LO[NDW], nS[NDW];
#pragma unroll
                for (int d = 0; d < NDW; d++) {
                    const uint32_t r0 = widen8_lo(raw[u][h][d]), r1 = widen8_hi(raw[u][h][d]);
                    HI[d] = pack8(pk_addsat(r0, s8[2 * d]), pk_addsat(r1, s8[2 * d + 1]));
                    LO[d] = pack8(pk_subsat(r0, s8[2 * d]), pk_subsat(r1, s8[2 * d + 1]));
                    nS[d] = 0u - __builtin_amdgcn_sad_u8(HI[d], LO[d], 0u);
                }
#pragma unroll
                for (int q = 0; q < 2; q++) { // q = 0: job h - 1 against frame h - 1; q = 1: job h against frame h + 1
                    const int t = h - 1 + q, fc = q ? h + 1 : h - 1;
                    if (t >= K)
                        continue;
                    if (J[t].handover < 0) {
                        uint32_t m[NDW];
#pragma unroll
                        for (int d = 0; d < NDW; d++)
                            m[d] = __builtin_amdgcn_sad_u8(raw[u][fc][d], LO[d], __builtin_amdgcn_sad_u8(raw[u][fc][d], HI[d], nS[d]));
                        k2b_row<NDW, SPLIT>(J[t], par, m, emit, y, map, lane, ngroups, budget, pend[t], gl);
                    }
                }
            }
        }
    }
#undef K2C_LOAD
    K2C_PUBLISH(0x7fffffff); // the scan is over: nobody waits for this wave's tail
#undef K2C_PUBLISH
    uint32_t tot = 0;
#pragma unroll
    for (int t = 0; t < K; t++) {
        if (t < k) {
            if (J[t].handover >= 0)
                k2b_hand_over(units, nunits, (jidx0 + (uint32_t)(S * t)) * (uint32_t)nchunks + (uint32_t)chunk, J[t].handover, y1, lane,
                              incomplete, jidx0 + (uint32_t)(S * t));
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
                sus_copy_out(pend[t], J[t].npend, jidx0 + (uint32_t)(S * t), gl.list, gb, lane);
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
            k2b_tail<false>(pend[t], J[t].npend, jobs[jidx0 + (uint32_t)(S * t)], frames, sigma6, W, H, hist, diff, cp, lane);
}

// ------------------------------------------------------------------------------------------------
// K2 generic: any W,H and the ROI overload.  256-thread workgroup, 32x8 output tile, LDS tile of
// packed (pos | neg<<16) with a 2-pixel halo; borders reflect inside the ROI.
// ------------------------------------------------------------------------------------------------
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
    int chain = -1;    // jobs per wave in the chained scan: 2 or 4 (>= 3 means 4); 0 = never chain; -1 = automatic (4)
    int budget = 512;  // suspects a chunk may remember (LDS) before it hands its rows over (<= K2B_PEND)
    int pf = 1;        // software-prefetch depth of the row machine in rows (1 or 2)
    int scanpf = -1;   // chained scan: rows fetched ahead, 1 or 2 (-1 = automatic: 2 where that is instantiated and measured faster)
    int split = 1;     // chained scan: "split" lane mapping where the row width allows it (0: always blocked)
    int list = 0;      // 1: suspects go to a global list that a second kernel evaluates (sus_tail_list), 0: the scanning
                       // waves evaluate their own.  K2's exact groups are expensive (45 window loads, two 5x5 sums) and few
                       // per wave: measured on the bench's trigger pass, in-wave 2.36 ms vs 2.47 ms through the list (K3,
                       // where every frame has its bubble and a group costs a 3x3 box, is the other way round: list always)
    int wg = -1;       // chained scan: waves per workgroup (1 .. K2C_MAXW) = consecutive segments of one chain (k2_sad_chain);
                       // -1 = automatic (K2C_DEFAULT_WG)
    int sync = -1;     // chained scan: row steps a wave may run ahead of its workgroup's slowest wave (0 = never waits;
                       // -1 = automatic, K2C_DEFAULT_SYNC)
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
        if (const char *e = getenv("ABUB_K2_WG"))
            g_k2opt.wg = atoi(e);
        if (const char *e = getenv("ABUB_K2_SYNC"))
            g_k2opt.sync = atoi(e);
        if (const char *e = getenv("ABUB_K2_SCANPF"))
            g_k2opt.scanpf = atoi(e);
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
    else if (!strcmp(name, "wg"))
        g_k2opt.wg = value;
    else if (!strcmp(name, "sync"))
        g_k2opt.sync = value;
    else if (!strcmp(name, "scanpf"))
        g_k2opt.scanpf = value;
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
        // bound-and-verify (see k2_bound_scan).  With `diff` (store mode) the images are cleared first -- a pure write
        // stream at the chip's fill rate -- and only the exact tails and the row machine write pixels afterwards.
        // (Contract, as for `hist`: the njobs output slots are 0 .. njobs - 1.)
        if (diff)
            HIPCHK(hipMemsetAsync(diff, 0, (size_t)njobs * (size_t)W * (size_t)H, st));
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
        uint32_t *npieces = counters;
        const bool deferred = ca.df_pieces != nullptr;
        if (deferred) { // the pieces go to the caller's list; the row machine is not launched here
            if (diff || ca.cthr || !ca.df_count || !ca.df_incomplete || (size_t)ca.df_cap < unitCap)
                return set_err(ABUB_E_INVALID, "deferred pieces: trigger-only launches with a list of abub_k2_pieces_cap() entries");
            units = ca.df_pieces;
            npieces = ca.df_count;
            HIPCHK(hipMemsetAsync(npieces, 0, sizeof(uint32_t), st));
            HIPCHK(hipMemsetAsync(ca.df_incomplete, 0, (size_t)njobs, st));
        }
        SusList gl;
        gl.list = gcap ? (uint2 *)(scr + 256 + unitBytes) : nullptr;
        gl.count = counters + 32;
        gl.cap = (uint32_t)gcap;
        const unsigned tgrid = (unsigned)((gcap + 256 * SUSL_UB - 1) / (256 * SUSL_UB) < 2048 ? (gcap + 256 * SUSL_UB - 1) / (256 * SUSL_UB) : 2048);
        HIPCHK(hipMemsetAsync(counters, 0, 256, st));
        const int L = ca.chain_len, S = ca.chain_stride;
        // jobs per wave: 4 (automatic) or 2 (option "chain" = 2)
        const int chainK = opt.chain < 0 ? 4 : opt.chain;
        if (chainK >= 2 && L > 0 && S > 0 && S <= 8 && njobs % L == 0 && !ca.cthr) {
            const int Kc = chainK >= 3 ? 4 : 2; // (the grid below must agree with the kernel's K)
            const int NW = opt.wg < 0 ? K2C_DEFAULT_WG : (opt.wg < 1 ? 1 : (opt.wg > K2C_MAXW ? K2C_MAXW : opt.wg));
            int nwgs = 0; // workgroups per block of L jobs: NW consecutive segments of Kc jobs each, per residue chain
            for (int r = 0; r < S; r++) {
                const int nr = (L - r + S - 1) / S;
                const int ns = nr > 0 ? (nr + Kc - 1) / Kc : 0;
                nwgs += (ns + NW - 1) / NW;
            }
            const dim3 grid((unsigned)((size_t)(njobs / L) * nwgs * nchunks));
            const size_t ldsBytes = ((size_t)NW * Kc * K2B_PEND + 64) * sizeof(uint32_t);
            const int syncD = opt.sync < 0 ? K2C_DEFAULT_SYNC : opt.sync;
            // the scan's own lane mapping: whole 16 / 8 / 4-byte pieces per lane ("split") where the row decomposes that
            // way (W = 1280, 1680, ...), the row machine's blocked mapping otherwise.  Measured with the v_sad_u8 scan,
            // A/B on one box: trigger-only -12 % at W = 1280 and -6 % at 1680, store mode -13 % / -4 % (round 2)
            constexpr bool CAN_SPLIT = NDW >= 5 && NDW <= 7;
            const bool split = CAN_SPLIT && opt.split;
            // rows fetched two steps ahead (183 instead of 153 registers: 2 waves per SIMD, measured -2 % on the bench's
            // trigger pass): NDW = 5, K = 4 with the split mapping only
            const bool pf2 = NDW == 5 && split && Kc == 4 && (opt.scanpf < 0 || opt.scanpf == 2);
#define K2C_ARGS frames, sigma6, jobs, L, S, nwgs, W, H, R, nchunks, budget, units, npieces, hist, diff, gl, syncD, ca.df_incomplete
#define K2C_LAUNCH_SP(KK, SP, PFD) \
    hipLaunchKernelGGL((k2_sad_chain<NDW, KK, SP, PFD>), grid, dim3(64 * NW), ldsBytes, st, K2C_ARGS)
#define K2C_LAUNCH(KK)                                                                                              \
    if (split) {                                                                                                    \
        K2C_LAUNCH_SP(KK, CAN_SPLIT, 1);                                                                            \
    } else {                                                                                                        \
        K2C_LAUNCH_SP(KK, false, 1);                                                                                \
    }
            if (pf2) {
                if constexpr (NDW == 5)
                    K2C_LAUNCH_SP(4, true, 2);
            } else if (Kc == 4) {
                K2C_LAUNCH(4);
            } else {
                K2C_LAUNCH(2);
            }
#undef K2C_LAUNCH_SP
#undef K2C_ARGS
#undef K2C_LAUNCH
        } else {
#define K2S_LAUNCH(CO)                                                                                              \
    hipLaunchKernelGGL((k2_bound_scan<NDW, CO>), dim3((unsigned)nunits), dim3(64), 0, st, frames, sigma6, jobs, W, H, \
                       R, nchunks, budget, units, npieces, hist, diff, ca.cthr, ca.pairs, ca.cap, ca.count,         \
                       ca.slot_base, gl, ca.df_incomplete)
            if (ca.cthr) {
                K2S_LAUNCH(true);
            } else {
                K2S_LAUNCH(false);
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
        if (deferred)
            return ABUB_OK;
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

// automatic rows per chunk of a launch of `njobs` jobs
static int k2_auto_rows(int njobs, int H)
{
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
    int R = (H + nch - 1) / nch;
    if (R < 16)
        R = 16;
    return R;
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
        const int R = rows_per_chunk ? rows_per_chunk : k2_auto_rows(njobs, H);
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

// ---- deferred pieces ------------------------------------------------------------------------------------------------
// The trigger search of a stack stops at its trigger frame (AnalyzerUnit.cpp:191, break at :307), but a batched launch
// evaluates whole blocks of frames before the host knows where that is.  The bound scan is cheap on dense frames (a chunk
// that keeps exceeding its suspect budget hands its remaining rows over and stops); the row machine on those rows is not.
// So the scan can leave the handed-over row ranges in a caller-owned list and only FLAG the jobs that have some; the host
// runs the row machine later, and only for the jobs its state machines actually reach.
extern "C" size_t abub_k2_pieces_cap(int njobs, int W, int H)
{
    if (njobs <= 0 || H <= 0)
        return 0;
    const int R = k2_auto_rows(njobs, H), nchunks = (H + R - 1) / R;
    return (size_t)njobs * nchunks * (size_t)((R + K2B_SUB - 1) / K2B_SUB);
}

extern "C" int abub_diff_hist_chained_deferred_dev(const uint8_t *frames, const uint8_t *sigma6, const abub_job *jobs, int njobs,
                                                   int W, int H, uint32_t *hist, int chain_len, int chain_stride,
                                                   void *pieces, uint32_t pieces_cap, uint32_t *npieces,
                                                   uint8_t *incomplete, void *stream)
{
    if (chain_len < 0 || chain_stride < 0 || !pieces || !npieces || !incomplete || !pick_ndw(W) || !k2_options().bound)
        return set_err(ABUB_E_INVALID, "abub_diff_hist_chained_deferred_dev: bad arguments (or no bound-and-verify pass for this width)");
    CompactArgs ca = {nullptr, nullptr, 0, nullptr, 0};
    ca.chain_len = chain_len;
    ca.chain_stride = chain_stride;
    ca.df_pieces = (uint2 *)pieces;
    ca.df_cap = pieces_cap;
    ca.df_count = npieces;
    ca.df_incomplete = incomplete;
    return diff_hist_impl(frames, sigma6, jobs, njobs, W, H, hist, nullptr, 0, ca, stream);
}

template <int NDW>
static void launch_pieces(const uint8_t *frames, const uint8_t *sigma6, const abub_job *jobs, int W, int H, int R, int nchunks,
                          uint32_t *hist, const uint2 *pieces, const uint32_t *npieces, const uint8_t *want, unsigned grid,
                          hipStream_t st)
{
    hipLaunchKernelGGL((k2_rows<NDW, false, 1, false>), dim3(grid), dim3(64), 0, st, frames, sigma6, jobs, W, H, R, nchunks, hist,
                       (uint8_t *)nullptr, (const int32_t *)nullptr, (uint32_t *)nullptr, 0u, (uint32_t *)nullptr, 0u, pieces,
                       npieces, want);
}

extern "C" int abub_diff_hist_pieces_dev(const uint8_t *frames, const uint8_t *sigma6, const abub_job *jobs, int njobs, int W,
                                         int H, uint32_t *hist, const void *pieces_, const uint32_t *npieces,
                                         const uint8_t *want, void *stream)
{
    const uint2 *pieces = (const uint2 *)pieces_;
    if (!frames || !sigma6 || !jobs || !hist || !pieces || !npieces || !want || njobs <= 0 || W <= 0 || H <= 0)
        return set_err(ABUB_E_INVALID, "abub_diff_hist_pieces_dev: bad arguments");
    const int ndw = pick_ndw(W);
    if (!ndw)
        return set_err(ABUB_E_INVALID, "abub_diff_hist_pieces_dev: no fast path for this width");
    hipStream_t st = (hipStream_t)stream;
    const int R = k2_auto_rows(njobs, H), nchunks = (H + R - 1) / R; // the geometry of the deferred launch over these njobs jobs
    const size_t cap = abub_k2_pieces_cap(njobs, W, H);
    const unsigned grid = (unsigned)(cap < 4096 ? cap : 4096);
    switch (ndw) {
    case 1: launch_pieces<1>(frames, sigma6, jobs, W, H, R, nchunks, hist, pieces, npieces, want, grid, st); break;
    case 2: launch_pieces<2>(frames, sigma6, jobs, W, H, R, nchunks, hist, pieces, npieces, want, grid, st); break;
    case 3: launch_pieces<3>(frames, sigma6, jobs, W, H, R, nchunks, hist, pieces, npieces, want, grid, st); break;
    case 4: launch_pieces<4>(frames, sigma6, jobs, W, H, R, nchunks, hist, pieces, npieces, want, grid, st); break;
    case 5: launch_pieces<5>(frames, sigma6, jobs, W, H, R, nchunks, hist, pieces, npieces, want, grid, st); break;
    case 6: launch_pieces<6>(frames, sigma6, jobs, W, H, R, nchunks, hist, pieces, npieces, want, grid, st); break;
    case 7: launch_pieces<7>(frames, sigma6, jobs, W, H, R, nchunks, hist, pieces, npieces, want, grid, st); break;
    default: launch_pieces<8>(frames, sigma6, jobs, W, H, R, nchunks, hist, pieces, npieces, want, grid, st); break;
    }
    HIPCHK(hipGetLastError());
    // bin 0 of the jobs whose rows are complete now (the deferred launch computed it from partial counts)
    hipLaunchKernelGGL(k_hist_bin0, dim3(njobs), dim3(64), 0, st, hist, (uint32_t)((size_t)W * H), want);
    HIPCHK(hipGetLastError());
    return ABUB_OK;
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
