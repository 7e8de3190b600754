// abub_png.hip -- the frames of a run decoded ON the GPU: PNG (8-bit grey or 8-bit palette, non-interlaced) -> u8 frames
// in the slab the trigger search reads.  Replaces, for the batched ingestion path, the per-frame host decode behind
// Parser::GetImage (ZipParser.cpp:186-239 / RawParser.cpp:30-47 -> cv::imread / cv::imdecode, i.e. libpng + zlib):
// a GPU box of this pool grants 16 host cores, and zlib's inflate is a serial byte recurrence (~2.3 ms per 1280x1024
// frame and core), so the host cannot decode more than ~7 k frames/s however the rest is arranged.
//
// Three kernels per batch of encoded frames (the file bytes are uploaded as they are on disk):
//   k_png_gather    one workgroup per frame: the IDAT chunks of the file (host-parsed offsets) -> one contiguous zlib stream
//   k_png_inflate   ONE WAVE per frame (inflate is sequential in the stream, so the parallelism is frames x lanes):
//                   every lane decodes one token SPECULATIVELY at bit offset ip + lane of the input (root-table look-up in
//                   LDS, canonical compare for codes longer than the root); the real token starts are the chain
//                   s, s + n(s), ... walked with v_readlane; an exclusive DPP scan of the output lengths places the
//                   tokens; literals and short matches with old sources are written by their own lanes, dependent /
//                   overlapping / long matches one after the other by the whole wave.  The 32 KiB history lives in LDS
//                   (a ring), flushed to HBM in 16-byte pieces with the Adler-32 of the stream accumulated on the way
//                   and checked against the trailer, as zlib does.
//   k_png_unfilter  one wave per frame, lanes along x, rows in order (filter types None / Sub / Up in a few ops per
//                   byte; Average and Paeth rows are a serial recurrence along x and take a lane-by-lane pass),
//                   palette -> grey through the frame's 256-entry table, rows written coalesced into the frame slab.
// Every access is bounds-checked against the sizes the caller states: the input is file content.  A frame the kernels
// refuse (status != 0) is left to the caller, which decodes it on the host (the result is the same image or the same
// failure: both follow RFC 1950/1951 and the PNG specification).
#include "abub_dev.hpp"
#include <stddef.h>

namespace {

constexpr int PNG_WIN = 32768;   // deflate history
constexpr int PNG_CAP = 512;     // output bytes one pass over a record's tokens may produce (ring = history + this)
constexpr int PNG_RING = PNG_WIN + PNG_CAP;
constexpr int PNG_INDW = 128;    // input ring in dwords (512 B, refilled 256 B at a time, one refill prefetched in registers)
constexpr int PNG_NSLOT = 2;     // token records in flight between the parsing and the writing wave
#ifndef PNG_FLUSH_BYTES
#define PNG_FLUSH_BYTES 2048
#endif
#ifndef PNG_SLEEP
#define PNG_SLEEP 1
#endif
constexpr int PNG_FLUSH = PNG_FLUSH_BYTES;  // unflushed output that triggers a flush (a tuning macro, like PNG_SLEEP below)
constexpr int LIT_ROOT = 10, DIST_ROOT = 8;
constexpr int PNG_SHORT = 8;     // matches up to this length are copied by their own lane


__device__ __forceinline__ uint32_t rfl(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ uint32_t rdl(uint32_t v, uint32_t lane) { return __builtin_amdgcn_readlane(v, lane); }
__device__ __forceinline__ uint64_t ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
__device__ __forceinline__ uint32_t below(uint64_t m) // set bits of m below this lane
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}
// cross-lane traffic through LDS inside one wave: the hardware keeps a wave's LDS operations in order, this keeps the
// compiler from moving them across the hand-over
__device__ __forceinline__ void wave_sync() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); }

// inclusive wave scan (add) in 6 DPP steps (the sequence of LLVM's atomic optimizer for gfx9: row_shr 1,2,4,8, then
// row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3)
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v)
{
    v += __builtin_amdgcn_update_dpp(0u, v, 0x111, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0u, v, 0x112, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0u, v, 0x114, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0u, v, 0x118, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0u, v, 0x142, 0xa, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0u, v, 0x143, 0xc, 0xf, false);
    return v;
}
__device__ __forceinline__ uint32_t wave_sum(uint32_t v) { return rdl(wave_incl_scan(v), 63); }

// canonical Huffman code of one alphabet (RFC 1951 3.2.2): uniform values, constant indices only (registers)
struct Canon {
    uint32_t first[16], count[16], offset[16];
    uint32_t longer; // codes longer than the root table's index
};

// counts per length, Kraft check (zlib's inftrees.c rules: over-subscribed is an error; incomplete only for a lone
// 1-bit distance code or an alphabet without any code), first codes and offsets.  `len` = this lane's symbols' lengths,
// one per chunk of 64 symbols.  Returns false for a code zlib rejects.
template <int CHUNKS>
__device__ __forceinline__ bool canon_counts(const uint32_t (&len)[CHUNKS], Canon &C, int root, bool allow_single)
{
#pragma unroll
    for (int l = 0; l < 16; ++l)
        C.count[l] = 0;
#pragma unroll
    for (int c = 0; c < CHUNKS; ++c)
#pragma unroll
        for (int l = 1; l < 16; ++l)
            C.count[l] += (uint32_t)__popcll(ballot(len[c] == (uint32_t)l));
    int left = 1;
    uint32_t mx = 0;
    bool over = false;
#pragma unroll
    for (int l = 1; l < 16; ++l) {
        left = 2 * left - (int)C.count[l];
        over |= left < 0;
        if (C.count[l])
            mx = l;
    }
    if (over)
        return false;
    if (left > 0 && mx != 0 && !(allow_single && mx == 1))
        return false;
    uint32_t code = 0, off = 0;
    C.first[0] = C.offset[0] = 0;
    C.longer = 0;
#pragma unroll
    for (int l = 1; l < 16; ++l) {
        code = (code + C.count[l - 1]) << 1;
        C.first[l] = code;
        C.offset[l] = off;
        off += C.count[l];
        if (l > root)
            C.longer += C.count[l];
    }
    return true;
}

// symbols sorted by (length, symbol): sorted[offset[l] + rank among the symbols of length l] = symbol
template <int CHUNKS>
__device__ __forceinline__ void canon_sort(const uint32_t (&len)[CHUNKS], const Canon &C, uint16_t *sorted, int lane)
{
    uint32_t run[16];
#pragma unroll
    for (int l = 0; l < 16; ++l)
        run[l] = C.offset[l];
#pragma unroll
    for (int c = 0; c < CHUNKS; ++c) {
        uint32_t pos = 0;
#pragma unroll
        for (int l = 1; l < 16; ++l) {
            const uint64_t m = ballot(len[c] == (uint32_t)l);
            if (len[c] == (uint32_t)l)
                pos = run[l] + below(m);
            run[l] += (uint32_t)__popcll(m);
        }
        if (len[c])
            sorted[pos] = (uint16_t)(c * 64 + lane);
    }
}

// canonical decode of the first bits of `rev` (the code bits MSB first in its low MAXL bits) for the lengths LO..HI:
// position in `sorted` and length, or length 0
template <int LO, int HI, int MAXL>
__device__ __forceinline__ uint32_t canon_find(uint32_t rev, const Canon &C, uint32_t &pos)
{
    uint32_t found = 0;
    pos = 0;
#pragma unroll
    for (int l = LO; l <= HI; ++l) {
        const uint32_t idx = (rev >> (MAXL - l)) - C.first[l];
        if (!found && idx < C.count[l]) {
            found = l;
            pos = C.offset[l] + idx;
        }
    }
    return found;
}

// table entry (u32): code length (4 bits; 0 = no code of at most root bits starts here) | kind << 4 | extra bits << 6 |
// base << 10 (literal: the byte; length / distance: the base value) | bit 31: a code longer than the root may match
enum { T_LIT = 0, T_LEN = 1, T_EOB = 2, T_BAD = 3 };
constexpr uint32_t T_LONG = 0x80000000u | (T_BAD << 4);
__device__ __forceinline__ uint32_t lit_entry(uint32_t l, uint32_t sym)
{
    if (sym < 256)
        return l | (T_LIT << 4) | (sym << 10);
    if (sym == 256)
        return l | (T_EOB << 4);
    if (sym >= 286)
        return T_BAD << 4;
    const uint32_t i = sym - 257;
    const uint32_t eb = (i < 8 || i == 28) ? 0u : (i - 4) >> 2;
    const uint32_t base = i < 8 ? 3 + i : i == 28 ? 258u : 3 + ((4 + (i & 3)) << eb);
    return l | (T_LEN << 4) | (eb << 6) | (base << 10);
}
__device__ __forceinline__ uint32_t dist_entry(uint32_t l, uint32_t sym)
{
    if (sym >= 30)
        return T_BAD << 4;
    const uint32_t eb = sym < 4 ? 0u : (sym - 2) >> 1;
    const uint32_t base = sym < 4 ? 1 + sym : 1 + ((2 + (sym & 1)) << eb);
    return l | (T_LIT << 4) | (eb << 6) | (base << 10);
}

// root table, one lane per entry: the entry's index bits are the next ROOT bits of the stream
template <int ROOT, bool DIST>
__device__ __forceinline__ void canon_table(const Canon &C, const uint16_t *sorted, uint32_t *tab, int lane)
{
    for (int e = lane; e < (1 << ROOT); e += 64) {
        const uint32_t rev = __brev((uint32_t)e) >> (32 - ROOT);
        uint32_t pos;
        const uint32_t l = canon_find<1, ROOT, ROOT>(rev, C, pos);
        uint32_t ent = C.longer ? T_LONG : (uint32_t)(T_BAD << 4);
        if (l) {
            const uint32_t sym = sorted[pos];
            ent = DIST ? dist_entry(l, sym) : lit_entry(l, sym);
        }
        tab[e] = ent;
    }
}
// the part of a canonical code the rare path needs (codes longer than the root), kept in LDS between the blocks' headers
struct CanonLds {
    uint16_t first[16], count[16], offset[16];
};
__device__ __forceinline__ void canon_store(const Canon &C, CanonLds &D, int lane)
{
    // (uniform values: every lane writes the same)
#pragma unroll
    for (int l = 0; l < 16; ++l) {
        D.first[l] = (uint16_t)C.first[l];
        D.count[l] = (uint16_t)C.count[l];
        D.offset[l] = (uint16_t)C.offset[l];
    }
}
// codes longer than the root: canonical compare per length on the next 15 bits
template <int ROOT, bool DIST>
__device__ __forceinline__ uint32_t canon_slow(uint32_t bits, const CanonLds &C, const uint16_t *sorted)
{
    const uint32_t rev = __brev(bits) >> 17;
    uint32_t found = 0, pos = 0;
#pragma unroll
    for (int l = ROOT + 1; l <= 15; ++l) {
        const uint32_t idx = (rev >> (15 - l)) - C.first[l];
        if (!found && idx < C.count[l]) {
            found = l;
            pos = C.offset[l] + idx;
        }
    }
    if (!found)
        return T_BAD << 4;
    const uint32_t sym = sorted[pos];
    return DIST ? dist_entry(found, sym) : lit_entry(found, sym);
}

__device__ const uint8_t png_cl_order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

// what the parsing wave hands to the writing wave
enum { REC_TOKENS = 0, REC_STORED = 1, REC_END = 2, REC_ERROR = 3 };
struct TokSlot {
    uint32_t tok[2][64]; // two windows of 64 bit offsets: literal = byte << 1; match = 1 | length << 1 | distance << 10
    uint32_t v[4];       // which of them are tokens of the stream (window 0: v[0] | v[1] << 32, window 1: v[2], v[3])
    uint32_t kind;       // REC_* (REC_STORED: | length << 2)
    uint32_t aux;        // REC_STORED: byte offset of the data in the stream; REC_END: the stream's Adler-32; REC_ERROR: status
    uint32_t pad[2];
};
struct InflateLds {
    alignas(16) uint8_t win[PNG_RING + 16]; // the ring, then a copy of its first bytes (a short read never has to wrap)
    alignas(16) uint32_t in[PNG_INDW];
    uint32_t lit[1 << LIT_ROOT];   // while a block's header is read, its code lengths live here (hdr_lens() ...)
    uint32_t dist[1 << DIST_ROOT];
    uint16_t slit[288];
    uint16_t sdist[32];
    CanonLds cl, cd;
    TokSlot q[PNG_NSLOT];
    uint32_t q_head, q_tail, q_abort, q_pad;
    uint8_t sink[16]; // where the byte stores of lanes that have nothing to store go (straight-line code, no exec masks)
};
// scratch of the header parse inside the (then dead) literal table: lens[320] | cl[32] | scl[32] (u16)
__device__ __forceinline__ uint8_t *hdr_lens(InflateLds &L) { return (uint8_t *)L.lit; }
__device__ __forceinline__ uint8_t *hdr_cl(InflateLds &L) { return (uint8_t *)L.lit + 320; }
__device__ __forceinline__ uint16_t *hdr_scl(InflateLds &L) { return (uint16_t *)((uint8_t *)L.lit + 352); }
static_assert(sizeof(InflateLds) <= 40960, "four inflate streams per CU");
static_assert(offsetof(InflateLds, win) == 0, "png_apply indexes the LDS block from the ring's first byte");

// ---- gather: IDAT chunks -> one zlib stream per frame ------------------------------------------------------------
__global__ __launch_bounds__(256) void k_png_gather(const uint8_t *__restrict__ files, uint64_t files_bytes,
                                                    const abub_png_frame *__restrict__ frames,
                                                    const abub_png_seg *__restrict__ segs, uint32_t nsegs,
                                                    uint8_t *__restrict__ zbuf, uint64_t zbuf_bytes,
                                                    int32_t *__restrict__ status)
{
    const int f = blockIdx.x, tid = threadIdx.x;
    const abub_png_frame fr = frames[f];
    const uint64_t zcap = (((uint64_t)fr.zlen + 15) & ~15ull) + 16;
    bool bad = (fr.zoff & 15) || (uint64_t)fr.zoff + zcap > zbuf_bytes || (uint64_t)fr.seg_begin + fr.seg_count > nsegs;
    uint64_t sum = 0;
    if (!bad)
        for (uint32_t s = 0; s < fr.seg_count; ++s) {
            const abub_png_seg sg = segs[fr.seg_begin + s];
            bad |= (uint64_t)sg.off + sg.len > files_bytes;
            sum += sg.len;
        }
    bad |= sum != fr.zlen;
    if (bad) { // (uniform: every thread read the same descriptor)
        if (tid == 0)
            status[f] = ABUB_PNG_E_DESC;
        return;
    }
    uint8_t *z = zbuf + fr.zoff;
    uint32_t run = 0;
    for (uint32_t s = 0; s < fr.seg_count; ++s) {
        const abub_png_seg sg = segs[fr.seg_begin + s];
        const uint8_t *src = files + sg.off;
        uint8_t *dst = z + run;
        // head bytes up to a dword boundary of the destination, then dwords assembled from two aligned source dwords
        const uint32_t head = min((uint32_t)((4 - ((uintptr_t)dst & 3)) & 3), sg.len);
        if ((uint32_t)tid < head)
            dst[tid] = src[tid];
        const uint32_t nd = (sg.len - head) >> 2;
        const uintptr_t sa0 = (uintptr_t)(src + head);
        const uint32_t mis = (uint32_t)(sa0 & 3);
        const uint32_t *sw = (const uint32_t *)(sa0 & ~(uintptr_t)3);
        // (the second dword of the last pair may lie past the segment: inside `files` unless the segment ends it)
        const bool tailSafe = (uint64_t)sg.off + sg.len + 8 <= files_bytes;
        const uint32_t ndSafe = tailSafe ? nd : (nd > 2 ? nd - 2 : 0);
        uint32_t *dw = (uint32_t *)(dst + head);
        for (uint32_t k = tid; k < ndSafe; k += 256)
            dw[k] = mis ? __builtin_amdgcn_alignbyte(sw[k + 1], sw[k], mis) : sw[k];
        const uint32_t done = head + 4 * ndSafe;
        for (uint32_t k = done + tid; k < sg.len; k += 256)
            dst[k] = src[k];
        run += sg.len;
    }
    for (uint32_t k = fr.zlen + tid; k < (uint32_t)zcap; k += 256)
        z[k] = 0;
    if (tid == 0)
        status[f] = 0;
}

// ---- inflate: one stream per workgroup; one wave does everything, or a parsing and a writing wave share the work ----
struct Parser {
    const uint8_t *z;  // the frame's zlib stream
    uint32_t zlen;
    uint32_t zpad;     // its length rounded up to 16 (readable, zero-filled behind the stream)
    uint32_t nbits;    // its length in bits
    uint32_t in_hi;    // the input ring holds the stream's bytes [in_hi - 512, in_hi)
    uint32_t pf;       // bytes [in_hi + 4 * lane, + 4), requested ahead
};
struct Writer {
    const uint8_t *z;  // (stored blocks are copied from the stream)
    uint8_t *raw;      // output of this frame
    uint32_t rawLen;   // exactly this many bytes are expected
    uint32_t op, op_r; // output position, and the same modulo the ring
    uint32_t fp;       // flushed up to here (multiple of 16)
    uint32_t a1, a2;   // Adler-32 of the flushed bytes
    bool far;          // this lane saw a distance that reaches before the output's first byte (checked at every flush)
};

__device__ __forceinline__ uint32_t png_load4(const Parser &P, uint32_t off)
{
    uint32_t v = 0;
    if (off + 4 <= P.zpad)
        v = *(const uint32_t *)(P.z + off);
    return v;
}
__device__ __forceinline__ void png_refill(Parser &P, InflateLds &L, int lane)
{
    L.in[((P.in_hi & (PNG_INDW * 4 - 1)) >> 2) + lane] = P.pf;
    P.in_hi += 256;
    P.pf = png_load4(P, P.in_hi + 4 * lane);
    wave_sync();
}
// (re)start the input ring at bit position ip
__device__ __forceinline__ void png_in_start(Parser &P, InflateLds &L, uint32_t ip, int lane)
{
    P.in_hi = (ip >> 3) & ~255u;
    P.pf = png_load4(P, P.in_hi + 4 * lane);
    png_refill(P, L, lane);
}
// the ring covers every dword the window reads at bit positions ip .. ip + 127 touch (and 32-bit peeks at ip)
__device__ __forceinline__ void png_ensure(Parser &P, InflateLds &L, uint32_t ip, int lane)
{
    const uint32_t need = ((ip + 191) >> 5) * 4 + 12;
    while (need > P.in_hi)
        png_refill(P, L, lane);
}
// 32 bits of the stream at bit position pos (any lane-specific pos inside the ensured range)
__device__ __forceinline__ uint32_t png_bits32(const InflateLds &L, uint32_t pos)
{
    const uint32_t w = pos >> 5;
    return __builtin_amdgcn_alignbit(L.in[(w + 1) & (PNG_INDW - 1)], L.in[w & (PNG_INDW - 1)], pos & 31);
}
__device__ __forceinline__ uint32_t ring_wrap(uint32_t i) { return i >= (uint32_t)PNG_RING ? i - PNG_RING : i; }

// ring -> HBM for the bytes [fp, upto), Adler-32 on the way; upto is a multiple of 16 unless `last`
__device__ __forceinline__ void png_flush(Writer &S, InflateLds &L, uint32_t upto, bool last, int lane)
{
    const uint32_t n = upto - S.fp;
    if (!n)
        return;
    uint32_t s1 = 0, s2 = 0; // this lane's sum of bytes and sum of (n - k) * byte[k], k = index inside the piece
    const uint32_t fr = S.fp % PNG_RING;
    for (uint32_t k0 = 16 * lane; k0 + 16 <= n; k0 += 1024) {
        const uint4 v = *(const uint4 *)&L.win[ring_wrap(fr + k0)];
        *(uint4 *)(S.raw + S.fp + k0) = v;
        const uint32_t d[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const uint32_t byte = (d[q] >> (8 * b)) & 255;
                s1 += byte;
                s2 += (n - (k0 + 4 * q + b)) * byte;
            }
    }
    if (last) {
        const uint32_t t0 = n & ~15u;
        if (t0 + lane < n) {
            const uint32_t byte = L.win[ring_wrap(fr + t0 + lane)];
            S.raw[S.fp + t0 + lane] = (uint8_t)byte;
            s1 += byte;
            s2 += (n - (t0 + lane)) * byte;
        }
    }
    // (a lane sees at most ceil(n / 1024) * 16 + 1 bytes of a piece of < 8 KiB: both sums stay far below 2^32)
    const uint32_t S1 = wave_sum(s1 % 65521u), S2 = wave_sum(s2 % 65521u);
    S.a2 = (S.a2 + ((n % 65521u) * S.a1) % 65521u + S2) % 65521u;
    S.a1 = (S.a1 + S1) % 65521u;
    S.fp = upto;
}
__device__ __forceinline__ void png_advance(Writer &S, InflateLds &L, uint32_t total, int lane)
{
    const uint32_t o0 = S.op_r;
    S.op += total;
    S.op_r = ring_wrap(o0 + total);
    if (o0 < 16 || S.op_r < o0) { // the ring's first bytes were written: renew their copy behind its end
        wave_sync();
        if (lane < 16)
            L.win[PNG_RING + lane] = L.win[lane];
    }
    if (S.op - S.fp >= (uint32_t)PNG_FLUSH) {
        wave_sync();
        png_flush(S, L, S.op & ~15u, false, lane);
    }
}

// ---- the writing side: tokens -> the history ring ----
// One record's tokens: two windows (valid[r] = which bit offsets of window r hold a token; tok as in TokSlot).  The two
// windows go through every step side by side: their dependent instruction chains interleave.  Returns a status.
__device__ __forceinline__ int png_apply(Writer &S, InflateLds &L, const uint64_t (&valid)[2], const uint32_t (&tok)[2], int lane, int dbg = 0)
{
    // the ring is the first member of the LDS block: byte stores go through one pointer, the sink (where lanes without a
    // byte to store write) is an index like any other
    uint8_t *const win = reinterpret_cast<uint8_t *>(&L);
    constexpr uint32_t SINK = (uint32_t)offsetof(InflateLds, sink);
    bool isM[2];
    uint32_t olen[2], dist[2], lit[2], incl[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const bool mine = (valid[r] >> lane) & 1;
        isM[r] = tok[r] & 1;
        olen[r] = mine ? (isM[r] ? (tok[r] >> 1) & 511u : 1u) : 0u;
        dist[r] = tok[r] >> 10;
        lit[r] = (tok[r] >> 1) & 255u;
    }
    incl[0] = wave_incl_scan(olen[0]);
    incl[1] = wave_incl_scan(olen[1]);
    incl[1] += rdl(incl[0], 63);
    uint64_t todo[2] = {valid[0], valid[1]};
    uint32_t consumed = 0;
    do {
        // the tokens of this pass: the whole record unless it carries more than PNG_CAP bytes (runs of long matches)
        bool sel[2];
        uint64_t selm[2];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            sel[r] = ((todo[r] >> lane) & 1) && incl[r] - consumed <= (uint32_t)PNG_CAP;
            selm[r] = ballot(sel[r]);
        }
        const uint32_t total = (selm[1] ? rdl(incl[1], 63 - (uint32_t)__builtin_clzll(selm[1]))
                                        : rdl(incl[0], 63 - (uint32_t)__builtin_clzll(selm[0]))) - consumed;
        if (S.op + total > S.rawLen)
            return ABUB_PNG_E_TOOMUCH;
        const bool nowrap = S.op_r + (uint32_t)PNG_CAP + 264 <= (uint32_t)PNG_RING; // no destination of this pass wraps
        if (nowrap) {
            // ---- the common pass: straight-line, three branches ----
            // A short match whose source lies wholly before this pass's first byte is copied by its own lane: 5 bytes read
            // and stored at once (8 when some match is longer; bytes behind a match's end go to the sink), no wrap to think
            // of (a source that runs over the ring's end reads the copy of the ring's first bytes kept there).  The others
            // -- sources inside this pass, overlapping or long ones -- follow one by one in stream order.
            uint32_t dst[2], sa[2];
            bool own[2], slow[2];
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const uint32_t off = incl[r] - olen[r] - consumed;
                dst[r] = S.op_r + off;
                const bool isMatch = sel[r] && isM[r];
                S.far |= isMatch && dist[r] > S.op + off; // (looked at when the ring is flushed: the bytes are not out before that)
                own[r] = isMatch && dist[r] >= off + olen[r] && olen[r] <= (uint32_t)PNG_SHORT;
                slow[r] = isMatch && !own[r];
                int si = (int)dst[r] - (int)dist[r]; // > -RING
                if (si < 0)
                    si += PNG_RING;
                sa[r] = own[r] ? (uint32_t)si : 0u;
                if (dbg != 3)
                    win[(sel[r] && !isM[r]) ? dst[r] : SINK] = (uint8_t)lit[r];
            }
            if (dbg < 2) {
                const bool any6 = ballot(own[0] && olen[0] > 5) | ballot(own[1] && olen[1] > 5);
                uint32_t b[2][PNG_SHORT];
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int k = 0; k < 5; ++k)
                        b[r][k] = win[sa[r] + k];
                if (any6) {
#pragma unroll
                    for (int r = 0; r < 2; ++r)
#pragma unroll
                        for (int k = 5; k < PNG_SHORT; ++k)
                            b[r][k] = win[sa[r] + k];
                }
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    const uint32_t da = own[r] ? dst[r] : SINK;
#pragma unroll
                    for (int k = 0; k < 3; ++k)
                        win[da + k] = (uint8_t)b[r][k];
                    win[(own[r] && olen[r] > 3) ? dst[r] + 3 : SINK] = (uint8_t)b[r][3];
                    win[(own[r] && olen[r] > 4) ? dst[r] + 4 : SINK] = (uint8_t)b[r][4];
                }
                if (any6) {
#pragma unroll
                    for (int r = 0; r < 2; ++r)
#pragma unroll
                        for (int k = 5; k < PNG_SHORT; ++k)
                            win[(own[r] && olen[r] > (uint32_t)k) ? dst[r] + k : SINK] = (uint8_t)b[r][k];
                }
                const uint64_t rest0 = ballot(slow[0]), rest1 = ballot(slow[1]);
                if (rest0 | rest1) {
#pragma unroll
                    for (int r = 0; r < 2; ++r) {
                        uint64_t rest = r ? rest1 : rest0;
                        while (rest) {
                            const uint32_t j = (uint32_t)__builtin_ctzll(rest);
                            rest &= rest - 1;
                            const uint32_t len = rdl(olen[r], j), d = rdl(dist[r], j), dst0 = rdl(dst[r], j);
                            wave_sync();
                            int s0 = (int)dst0 - (int)d;
                            if (s0 < 0)
                                s0 += PNG_RING;
                            const float rcp = 1.0f / (float)d;
                            for (uint32_t k0 = 0; k0 < len; k0 += 64) {
                                const uint32_t k = k0 + lane;
                                uint32_t q = k;
                                if (d < len) { // overlapping: byte k repeats the pattern of d bytes
                                    const uint32_t quo = (uint32_t)((float)k * rcp);
                                    int rr = (int)k - (int)(quo * d);
                                    if (rr < 0)
                                        rr += (int)d;
                                    if (rr >= (int)d)
                                        rr -= (int)d;
                                    q = (uint32_t)rr;
                                }
                                if (k < len) {
                                    const uint8_t byte = win[ring_wrap((uint32_t)s0 + q)];
                                    win[dst0 + k] = byte;
                                }
                            }
                        }
                    }
                }
            }
        } else {
            // ---- a pass near the ring's end (one in ~60): every index wrapped, every match in stream order ----
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const uint32_t off = incl[r] - olen[r] - consumed;
                const bool isMatch = sel[r] && isM[r];
                S.far |= isMatch && dist[r] > S.op + off;
                if (sel[r] && !isM[r] && dbg != 3)
                    win[ring_wrap(S.op_r + off)] = (uint8_t)lit[r];
                uint64_t rest = dbg < 2 ? ballot(isMatch) : 0;
                while (rest) {
                    const uint32_t j = (uint32_t)__builtin_ctzll(rest);
                    rest &= rest - 1;
                    const uint32_t o = rdl(off, j), len = rdl(olen[r], j), d = rdl(dist[r], j);
                    wave_sync();
                    const uint32_t dst0 = ring_wrap(S.op_r + o);
                    const uint32_t src0 = ring_wrap(ring_wrap(S.op_r + o + PNG_RING - min(d, (uint32_t)PNG_RING)));
                    const float rcp = 1.0f / (float)d;
                    for (uint32_t k0 = 0; k0 < len; k0 += 64) {
                        const uint32_t k = k0 + lane;
                        uint32_t q = k;
                        if (d < len) {
                            const uint32_t quo = (uint32_t)((float)k * rcp);
                            int rr = (int)k - (int)(quo * d);
                            if (rr < 0)
                                rr += (int)d;
                            if (rr >= (int)d)
                                rr -= (int)d;
                            q = (uint32_t)rr;
                        }
                        if (k < len) {
                            const uint8_t byte = win[ring_wrap(src0 + q)];
                            win[ring_wrap(dst0 + k)] = byte;
                        }
                    }
                }
            }
        }
        png_advance(S, L, total, lane);
        todo[0] &= ~selm[0];
        todo[1] &= ~selm[1];
        consumed += total;
    } while (todo[0] | todo[1]);
    return 0;
}
// a stored block's bytes, straight from the stream
__device__ __forceinline__ int png_apply_stored(Writer &S, InflateLds &L, uint32_t bp, uint32_t len, int lane)
{
    if (S.op + len > S.rawLen)
        return ABUB_PNG_E_TOOMUCH;
    for (uint32_t rem = len; rem;) {
        const uint32_t chunk = min(rem, (uint32_t)PNG_CAP);
        for (uint32_t k = lane; k < chunk; k += 64)
            L.win[ring_wrap(S.op_r + k)] = S.z[bp + k];
        bp += chunk;
        rem -= chunk;
        png_advance(S, L, chunk, lane);
    }
    return 0;
}
// the end of the stream: everything out, size and checksum as zlib checks them
__device__ __forceinline__ int png_apply_end(Writer &S, InflateLds &L, uint32_t adler, int lane)
{
    wave_sync();
    png_flush(S, L, S.op, true, lane);
    if (ballot(S.far))
        return ABUB_PNG_E_DISTANCE;
    if (S.op != S.rawLen)
        return ABUB_PNG_E_TOOLITTLE;
    return adler == ((S.a2 << 16) | S.a1) ? 0 : ABUB_PNG_E_ADLER;
}

// ---- the hand-over between the two waves ----
__device__ __forceinline__ uint32_t q_ld(uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void q_st(uint32_t *p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
constexpr uint32_t PNG_SPINS = 1u << 19; // x >= 64 cycles: a wave that waits this long has lost its partner

// The receiving end of the parser's records: TWO = false applies them on the spot (one wave does everything), TWO = true
// puts them into the LDS queue for the writing wave.
template <bool TWO>
struct Sink {
    InflateLds &L;
    Writer *W;      // !TWO
    uint32_t head;  // TWO
    uint32_t tail_pf; // TWO: the consumer's counter as last seen
    int lane;
    int dbg;        // measurement only (ABUB_PNG_DEBUG): 1 = the records are dropped (what the parsing alone costs), 2 = no match
                    // copies, 3 = no stores at all, 4 = no chain walk, 5 = two waves, the status carries the queue's hand-over latency
    // returns a status for !TWO; for TWO: 0, or -1 when the writing wave has gone (the parser then just stops)
    __device__ __forceinline__ int put(uint32_t kind, uint32_t aux, const uint64_t (&valid)[2], const uint32_t (&tok)[2])
    {
        if (!TWO) {
            const uint32_t k = kind & 3;
            if (dbg == 1 || dbg == 4)
                return 0;
            if (k == REC_TOKENS)
                return png_apply(*W, L, valid, tok, lane, dbg);
            if (k == REC_STORED)
                return png_apply_stored(*W, L, aux, kind >> 2, lane);
            if (k == REC_END)
                return dbg ? 0 : png_apply_end(*W, L, aux, lane);
            return (int)aux;
        }
        // (tail_pf was requested when the last record went out: it is here by now, and usually says "room")
        uint32_t spins = 0;
        while (head - tail_pf >= (uint32_t)PNG_NSLOT) {
            tail_pf = q_ld(&L.q_tail);
            if (head - tail_pf < (uint32_t)PNG_NSLOT)
                break;
            if (q_ld(&L.q_abort) || ++spins > PNG_SPINS)
                return -1;
            __builtin_amdgcn_s_sleep(PNG_SLEEP);
        }
        TokSlot &sl = L.q[head % PNG_NSLOT];
        sl.tok[0][lane] = tok[0];
        sl.tok[1][lane] = tok[1];
        if (lane == 0) {
            sl.v[0] = (uint32_t)valid[0];
            sl.v[1] = (uint32_t)(valid[0] >> 32);
            sl.v[2] = (uint32_t)valid[1];
            sl.v[3] = (uint32_t)(valid[1] >> 32);
            sl.kind = kind;
            sl.aux = (dbg == 5 && (kind & 3) == REC_TOKENS) ? (uint32_t)__builtin_readcyclecounter() : aux;
        }
        // (a wave's LDS operations execute in order: the counter's store follows the record's without waiting for them)
        wave_sync();
        ++head;
        if (lane == 0)
            q_st(&L.q_head, head);
        tail_pf = q_ld(&L.q_tail);
        return 0;
    }
    __device__ __forceinline__ int put(uint32_t kind, uint32_t aux)
    {
        const uint64_t v[2] = {0, 0};
        const uint32_t t[2] = {0, 0};
        return put(kind, aux, v, t);
    }
};

// ---- the parsing side ----
// tables of one block (btype 1: the fixed code, 2: the code lengths in the stream at ip); returns a status
__device__ __forceinline__ int png_block_tables(Parser &P, InflateLds &L, uint32_t &ip, uint32_t btype, int lane)
{
    uint8_t *const lens = hdr_lens(L); // lens[0 .. 288) literal/length, lens[288 .. 320) distance
    uint32_t hlit = 288, hdist = 32;
    if (btype == 1) {
        for (int q = lane; q < 320; q += 64)
            lens[q] = (uint8_t)(q < 144 ? 8 : q < 256 ? 9 : q < 280 ? 7 : q < 288 ? 8 : 5);
    } else {
        uint8_t *const cl = hdr_cl(L);
        uint16_t *const scl = hdr_scl(L);
        if (ip + 14 > P.nbits)
            return ABUB_PNG_E_TRUNCATED;
        const uint32_t hv = rfl(png_bits32(L, ip));
        hlit = (hv & 31) + 257;
        hdist = ((hv >> 5) & 31) + 1;
        const uint32_t hclen = ((hv >> 10) & 15) + 4;
        ip += 14;
        if (hlit > 286 || hdist > 30)
            return ABUB_PNG_E_SYMBOLS;
        png_ensure(P, L, ip, lane);
        if (lane < 32)
            cl[lane] = 0;
        wave_sync();
        if ((uint32_t)lane < hclen)
            cl[png_cl_order[lane]] = (uint8_t)(png_bits32(L, ip + 3 * lane) & 7);
        wave_sync();
        ip += 3 * hclen;
        uint32_t cll[1] = {lane < 19 ? (uint32_t)cl[lane] : 0u};
        Canon CC;
        if (!canon_counts<1>(cll, CC, 7, false) || CC.count[1] + CC.count[2] + CC.count[3] + CC.count[4] + CC.count[5] + CC.count[6] + CC.count[7] == 0)
            return ABUB_PNG_E_CODES;
        canon_sort<1>(cll, CC, scl, lane);
        wave_sync();
        // the code-length code's 128-entry table lives in two registers per lane (entry e: lane e & 63, register e >> 6)
        uint32_t clt[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const uint32_t rev = __brev((uint32_t)(lane + 64 * q)) >> 25;
            uint32_t pos;
            const uint32_t l = canon_find<1, 7, 7>(rev, CC, pos);
            clt[q] = l ? (l | ((uint32_t)scl[pos] << 4)) : 0u;
        }
        const uint32_t total = hlit + hdist;
        uint32_t n = 0, prev = 0;
        while (n < total) {
            png_ensure(P, L, ip, lane);
            uint32_t v = rfl(png_bits32(L, ip));
            const uint32_t idx = v & 127;
            const uint32_t ent = idx < 64 ? rdl(clt[0], idx) : rdl(clt[1], idx - 64);
            const uint32_t l = ent & 15, sym = ent >> 4;
            if (!l)
                return ABUB_PNG_E_CODES;
            v >>= l;
            ip += l;
            uint32_t rep = 1, val = sym;
            if (sym == 16) {
                if (!n)
                    return ABUB_PNG_E_CODES;
                rep = 3 + (v & 3);
                val = prev;
                ip += 2;
            } else if (sym == 17) {
                rep = 3 + (v & 7);
                val = 0;
                ip += 3;
            } else if (sym == 18) {
                rep = 11 + (v & 127);
                val = 0;
                ip += 7;
            }
            if (n + rep > total)
                return ABUB_PNG_E_CODES;
            for (uint32_t k = lane; k < rep; k += 64) {
                const uint32_t q = n + k;
                lens[q < hlit ? q : 288 + (q - hlit)] = (uint8_t)val;
            }
            prev = val;
            n += rep;
        }
        if (ip > P.nbits)
            return ABUB_PNG_E_TRUNCATED;
    }
    wave_sync();
    Canon CL, CD;
    uint32_t ll[5];
#pragma unroll
    for (int c = 0; c < 5; ++c) {
        const uint32_t q = c * 64 + lane;
        ll[c] = q < hlit ? (uint32_t)lens[q] : 0u;
    }
    uint32_t dl[1] = {(uint32_t)lane < hdist ? (uint32_t)lens[288 + lane] : 0u};
    wave_sync(); // (the lengths are in registers: the literal table may now overwrite them)
    const uint32_t eobLen = rdl(ll[4], 0); // symbol 256
    if (!canon_counts<5>(ll, CL, LIT_ROOT, true) || !canon_counts<1>(dl, CD, DIST_ROOT, true))
        return ABUB_PNG_E_CODES;
    if (!eobLen)
        return ABUB_PNG_E_NOEOB;
    canon_sort<5>(ll, CL, L.slit, lane);
    canon_sort<1>(dl, CD, L.sdist, lane);
    canon_store(CL, L.cl, lane);
    canon_store(CD, L.cd, lane);
    wave_sync();
    canon_table<LIT_ROOT, false>(CL, L.slit, L.lit, lane);
    canon_table<DIST_ROOT, true>(CD, L.sdist, L.dist, lane);
    wave_sync();
    return 0;
}

// the whole stream: headers, tables, tokens -> records for `sink`.  Returns a status (0 after REC_END went out; -1 when the
// sink has gone).
template <bool TWO>
__device__ __forceinline__ int png_parse(Parser &P, InflateLds &L, Sink<TWO> &sink, int lane)
{
    uint32_t ip = 16;
    png_in_start(P, L, 0, lane);
    png_ensure(P, L, 0, lane);
    if (P.zlen < 6 || P.zlen >= (1u << 28))
        return ABUB_PNG_E_TRUNCATED;
    {
        const uint32_t h = rfl(png_bits32(L, 0));
        const uint32_t cmf = h & 255, flg = (h >> 8) & 255;
        if ((cmf & 15) != 8 || (cmf >> 4) > 7 || ((cmf << 8) | flg) % 31 || (flg & 0x20))
            return ABUB_PNG_E_HEADER;
    }
    bool final = false;
    while (!final) {
        png_ensure(P, L, ip, lane);
        if (ip + 3 > P.nbits)
            return ABUB_PNG_E_TRUNCATED;
        uint32_t hv = rfl(png_bits32(L, ip));
        final = hv & 1;
        const uint32_t btype = (hv >> 1) & 3;
        ip += 3;
        if (btype == 3)
            return ABUB_PNG_E_BLOCKTYPE;
        if (btype == 0) { // stored: the writing side copies it out of the stream
            ip = (ip + 7) & ~7u;
            png_ensure(P, L, ip, lane);
            if (ip + 32 > P.nbits)
                return ABUB_PNG_E_TRUNCATED;
            hv = rfl(png_bits32(L, ip));
            const uint32_t len = hv & 0xffff;
            if ((len ^ (hv >> 16)) != 0xffff)
                return ABUB_PNG_E_STORED;
            ip += 32;
            const uint32_t bp = ip >> 3;
            if ((uint64_t)bp + len > P.zlen)
                return ABUB_PNG_E_TRUNCATED;
            if (len) {
                const int rc = sink.put(REC_STORED | (len << 2), bp);
                if (rc)
                    return rc;
            }
            ip = (bp + len) * 8;
            png_in_start(P, L, ip, lane);
            continue;
        }
        {
            const int rc = png_block_tables(P, L, ip, btype, lane);
            if (rc)
                return rc;
        }
        // ---- tokens ----
        uint32_t s = 0; // first real token start inside the two windows
        for (;;) {
            png_ensure(P, L, ip, lane);
            if (ip > P.nbits)
                return ABUB_PNG_E_TRUNCATED;
            // every lane: one token at bit offset ip + lane (window 0) and one at ip + 64 + lane (window 1; it only counts
            // when window 0 does not end the block).  The two go through the look-ups side by side.
            const uint32_t pos = ip + lane, w = pos >> 5, sh = pos & 31;
            uint32_t d[5];
#pragma unroll
            for (int k = 0; k < 5; ++k)
                d[k] = L.in[(w + k) & (PNG_INDW - 1)];
            uint32_t lo[2], hi[2], e[2], de[2], w2[2], nb[2], kind[2], tok[2];
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                lo[r] = __builtin_amdgcn_alignbit(d[2 * r + 1], d[2 * r], sh);
                hi[r] = __builtin_amdgcn_alignbit(d[2 * r + 2], d[2 * r + 1], sh);
                e[r] = L.lit[lo[r] & ((1u << LIT_ROOT) - 1)];
            }
            // the distance code behind a length code (looked up for every lane: the ones without a length code drop it)
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const uint32_t nb2 = (e[r] & 15) + ((e[r] >> 6) & 15);   // <= 20
                w2[r] = __builtin_amdgcn_alignbit(hi[r], lo[r], nb2);     // the 32 bits behind the length code
                de[r] = L.dist[w2[r] & ((1u << DIST_ROOT) - 1)];
            }
            // A lane whose bits start a code longer than the tables' index is left for later (it advances by one bit like an
            // invalid code): nearly all of them are lanes the chain never visits -- half of the records have such a lane
            // somewhere among their 128, a few per cent have one ON the chain.
            bool lng[2];
            auto finish = [&](int r) {
                const uint32_t lnb = e[r] & 15, eb = (e[r] >> 6) & 15, base = (e[r] >> 10) & 0x1fffffu;
                kind[r] = (e[r] >> 4) & 3;
                const uint32_t len = base + __builtin_amdgcn_ubfe(lo[r], lnb, eb);
                const uint32_t dnb = de[r] & 15, deb = (de[r] >> 6) & 15; // dnb + deb <= 28
                const uint32_t dd = ((de[r] >> 10) & 0x1fffffu) + __builtin_amdgcn_ubfe(w2[r], dnb, deb);
                const bool isLen = kind[r] == T_LEN;
                lng[r] = (e[r] >> 31) || (isLen && (de[r] >> 31));
                tok[r] = isLen ? (1u | (len << 1) | (dd << 10)) : ((base & 255u) << 1);
                nb[r] = isLen ? lnb + eb + dnb + deb : lnb;
                if (isLen && ((de[r] >> 4) & 3) == T_BAD)
                    kind[r] = T_BAD;
            };
            finish(0);
            finish(1);
            // ---- the chain of real tokens: s, s + n(s), ...  (an invalid code advances by one bit: looked at afterwards) ----
            uint64_t valid[2];
            uint32_t p;
            auto walk = [&]() {
                const uint32_t step0 = (kind[0] == T_BAD) ? 1u : nb[0], step1 = (kind[1] == T_BAD) ? 1u : nb[1];
                valid[0] = valid[1] = 0;
                p = s;
                do {
                    valid[0] |= 1ull << p;
                    p += rdl(step0, p);
                } while (p < 64);
                p -= 64;
                do {
                    valid[1] |= 1ull << p;
                    p += rdl(step1, p);
                } while (p < 64);
            };
            if (sink.dbg == 4) { // (measurement only: what everything but the walk costs)
                valid[0] = valid[1] = 0x0101010101010101ull;
                p = 64 + (rdl(nb[0], 0) & 1) + (rdl(nb[1], 0) & 1);
            } else {
                walk();
                if (ballot(((valid[0] >> lane) & 1) && lng[0]) | ballot(((valid[1] >> lane) & 1) && lng[1])) {
                    // a long code on the chain: the canonical compare for every lane that has one, then the walk again
#pragma unroll
                    for (int r = 0; r < 2; ++r) {
                        if (e[r] >> 31) {
                            e[r] = canon_slow<LIT_ROOT, false>(lo[r], L.cl, L.slit);
                            const uint32_t nb2 = (e[r] & 15) + ((e[r] >> 6) & 15);
                            w2[r] = __builtin_amdgcn_alignbit(hi[r], lo[r], nb2);
                            de[r] = L.dist[w2[r] & ((1u << DIST_ROOT) - 1)];
                        }
                        if (((e[r] >> 4) & 3) == T_LEN && (de[r] >> 31))
                            de[r] = canon_slow<DIST_ROOT, true>(w2[r], L.cd, L.sdist);
                        finish(r);
                    }
                    walk();
                }
            }
            bool eob = false;
            {
                // end of block or an invalid code on the chain: what follows is not data
                const uint64_t stop0 = ballot(((valid[0] >> lane) & 1) && kind[0] >= T_EOB);
                const uint64_t stop1 = ballot(((valid[1] >> lane) & 1) && kind[1] >= T_EOB);
                if ((stop0 | stop1) && sink.dbg != 4) {
                    if (stop0) {
                        const uint32_t j = (uint32_t)__builtin_ctzll(stop0);
                        if (rdl(kind[0], j) == T_BAD)
                            return ABUB_PNG_E_CODE;
                        valid[0] &= (1ull << j) - 1; // (the end-of-block token itself carries no bytes)
                        valid[1] = 0;
                        p = j + rdl(nb[0], j);
                    } else {
                        const uint32_t j = (uint32_t)__builtin_ctzll(stop1);
                        if (rdl(kind[1], j) == T_BAD)
                            return ABUB_PNG_E_CODE;
                        valid[1] &= (1ull << j) - 1;
                        p = 64 + j + rdl(nb[1], j);
                    }
                    eob = true;
                }
            }
            if (eob)
                ip += p;
            else {
                ip += 128;
                s = p - 64;
            }
            if (ip > P.nbits)
                return ABUB_PNG_E_TRUNCATED;
            if (valid[0] | valid[1]) {
                const int rc = sink.put(REC_TOKENS, 0, valid, tok);
                if (rc)
                    return rc;
            }
            if (eob)
                break;
        }
    }
    // the trailer: Adler-32 of the output, big-endian, at the next byte boundary
    ip = (ip + 7) & ~7u;
    if (ip + 32 > P.nbits)
        return ABUB_PNG_E_TRUNCATED;
    png_ensure(P, L, ip, lane);
    const uint32_t adler = __builtin_bswap32(rfl(png_bits32(L, ip)));
    return sink.put(REC_END, adler);
}

template <bool TWO>
__global__ __launch_bounds__(TWO ? 128 : 64) void k_png_inflate(const uint8_t *__restrict__ zbuf, const abub_png_frame *__restrict__ frames,
                                                                  uint32_t rawLen, uint64_t rawStride, uint8_t *__restrict__ rawbuf,
                                                                  int32_t *__restrict__ status, int dbg)
{
    __shared__ InflateLds L;
    const int f = blockIdx.x, lane = threadIdx.x & 63;
    const uint32_t wave = rfl(threadIdx.x >> 6);
    if (TWO) {
        if (threadIdx.x == 0) {
            L.q_head = 0;
            L.q_tail = 0;
            L.q_abort = 0;
        }
        __syncthreads();
    }
    if (status[f] != 0) // (refused by the gather kernel; the same for the whole workgroup)
        return;
    const abub_png_frame fr = frames[f];
    Writer W;
    W.z = zbuf + fr.zoff;
    W.raw = rawbuf + (uint64_t)f * rawStride;
    W.rawLen = rawLen;
    W.op = W.op_r = W.fp = 0;
    W.a1 = 1;
    W.a2 = 0;
    W.far = false;
    if (!TWO || wave == 0) {
        Parser P;
        P.z = zbuf + fr.zoff;
        P.zlen = fr.zlen;
        P.zpad = (fr.zlen + 15) & ~15u;
        P.nbits = fr.zlen * 8;
        Sink<TWO> sink = {L, &W, 0, 0, lane, dbg};
        int rc = png_parse<TWO>(P, L, sink, lane);
        if (!TWO) {
            if (lane == 0)
                status[f] = rc;
        } else if (rc > 0)
            (void)sink.put(REC_ERROR, (uint32_t)rc); // (rc == -1: the writing wave has set the status and gone)
        return;
    }
    // the writing wave
    uint32_t tail = 0;
    int rc = 0;
    uint64_t latSum = 0; // dbg == 5: cycles between a record's publication and its pick-up, over the pick-ups that had waited
    uint32_t latN = 0;
    for (;;) {
        uint32_t spins = 0;
        while (q_ld(&L.q_head) == tail) {
            if (++spins > PNG_SPINS) {
                rc = ABUB_PNG_E_INTERNAL;
                break;
            }
            __builtin_amdgcn_s_sleep(PNG_SLEEP);
        }
        if (rc)
            break;
        wave_sync();
        const TokSlot &sl = L.q[tail % PNG_NSLOT];
        const uint32_t kind = rfl(sl.kind), aux = rfl(sl.aux);
        const uint64_t valid[2] = {((uint64_t)rfl(sl.v[1]) << 32) | rfl(sl.v[0]), ((uint64_t)rfl(sl.v[3]) << 32) | rfl(sl.v[2])};
        const uint32_t tok[2] = {sl.tok[0][lane], sl.tok[1][lane]};
        if (dbg == 5 && spins > 0 && (kind & 3) == REC_TOKENS) {
            latSum += (uint32_t)((uint32_t)__builtin_readcyclecounter() - aux);
            ++latN;
        }
        // (the record is in registers: the slot may be refilled while its tokens are applied)
        wave_sync(); // (in-order LDS: the slot's loads execute before the counter's store)
        ++tail;
        if (lane == 0)
            q_st(&L.q_tail, tail);
        const uint32_t k = kind & 3;
        bool done = true;
        if (k == REC_TOKENS) {
            rc = png_apply(W, L, valid, tok, lane);
            done = rc != 0;
        } else if (k == REC_STORED) {
            rc = png_apply_stored(W, L, aux, kind >> 2, lane);
            done = rc != 0;
        } else if (k == REC_END)
            rc = png_apply_end(W, L, aux, lane);
        else
            rc = (int)aux;
        if (done)
            break;
    }
    if (lane == 0) {
        status[f] = (dbg == 5 && rc == 0) ? -(int)(latN ? latSum / latN : 0) - 1000000 * (int)min(latN / 1000u, 2000u) : rc;
        q_st(&L.q_abort, 1u);
    }
}

// ---- unfilter: one wave per frame, lanes along x, rows in order ----------------------------------------------------
__device__ __forceinline__ uint32_t add4(uint32_t a, uint32_t b) // four byte-wise sums modulo 256
{
    return ((a & 0x7f7f7f7fu) + (b & 0x7f7f7f7fu)) ^ ((a ^ b) & 0x80808080u);
}
__device__ __forceinline__ int paeth(int a, int b, int c)
{
    const int pa = abs(b - c), pb = abs(a - c), pc = abs(a + b - 2 * c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

template <int NDW>
__global__ __launch_bounds__(64) void k_png_unfilter(const uint8_t *__restrict__ rawbuf, uint64_t rawStride,
                                                     const abub_png_frame *__restrict__ frames, const uint8_t *__restrict__ luts,
                                                     uint32_t nluts, int W, int H, uint8_t *__restrict__ out, uint64_t out_bytes,
                                                     int32_t *__restrict__ status)
{
    __shared__ uint8_t lut[256];
    const int f = blockIdx.x, lane = threadIdx.x;
    if (status[f] != 0)
        return;
    const abub_png_frame fr = frames[f];
    const bool pal = fr.lut != 0xffffffffu;
    if (fr.dst + (uint64_t)W * H > out_bytes || (fr.dst & 3) || (pal && fr.lut >= nluts)) {
        if (lane == 0)
            status[f] = ABUB_PNG_E_DESC;
        return;
    }
    if (pal) {
        for (int i = lane; i < 256; i += 64)
            lut[i] = luts[(uint64_t)fr.lut * 256 + i];
        wave_sync();
    }
    const uint8_t *raw = rawbuf + (uint64_t)f * rawStride; // 16-byte aligned; row y at y * (W + 1), its filter type first
    uint8_t *dst = out + fr.dst;
    const int x0 = lane * 4 * NDW;            // this lane's bytes of a row: [x0, x0 + 4 * NDW)
    const int nmine = min(max(W - x0, 0), 4 * NDW); // (a multiple of 4: W is)
    uint32_t prev[NDW];
#pragma unroll
    for (int q = 0; q < NDW; ++q)
        prev[q] = 0;
    // row loads run two rows ahead of the arithmetic
    auto load_row = [&](int y, uint32_t (&d)[NDW + 1], uint32_t &ft) {
        const uint64_t rb = (uint64_t)y * (W + 1);
        ft = raw[rb];
        const uint64_t a = rb + 1 + x0;
        const uint32_t *p = (const uint32_t *)(raw + (a & ~3ull));
#pragma unroll
        for (int q = 0; q <= NDW; ++q) // (rawStride leaves 8 readable bytes behind the last row)
            d[q] = (nmine > 0 && 4 * q < nmine + 4) ? p[q] : 0u;
    };
    auto row_words = [&](int y, const uint32_t (&d)[NDW + 1], uint32_t (&r)[NDW]) {
        const uint32_t mis = (uint32_t)(((uint64_t)y * (W + 1) + 1 + x0) & 3);
#pragma unroll
        for (int q = 0; q < NDW; ++q)
            r[q] = 4 * q < nmine ? (mis ? __builtin_amdgcn_alignbyte(d[q + 1], d[q], mis) : d[q]) : 0u;
    };
    uint32_t dA[NDW + 1], dB[NDW + 1], ftA = 0, ftB = 0;
    if (H > 0)
        load_row(0, dA, ftA);
    if (H > 1)
        load_row(1, dB, ftB);
    int bad = 0;
    for (int y = 0; y < H; ++y) {
        uint32_t r[NDW];
        row_words(y, dA, r);
        const uint32_t ft = rfl(ftA);
#pragma unroll
        for (int q = 0; q <= NDW; ++q)
            dA[q] = dB[q];
        ftA = ftB;
        if (y + 2 < H)
            load_row(y + 2, dB, ftB);
        uint32_t o[NDW];
        if (ft == 0) {
#pragma unroll
            for (int q = 0; q < NDW; ++q)
                o[q] = r[q];
        } else if (ft == 2) {
#pragma unroll
            for (int q = 0; q < NDW; ++q)
                o[q] = add4(r[q], prev[q]);
        } else if (ft == 1) {
            // prefix sums modulo 256 along the row: inside the lane, then the lanes' totals
            uint32_t acc = 0;
#pragma unroll
            for (int q = 0; q < NDW; ++q) {
                uint32_t w = 0;
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    acc = (acc + ((r[q] >> (8 * b)) & 255)) & 255;
                    w |= acc << (8 * b);
                }
                o[q] = w;
            }
            const uint32_t carry = (wave_incl_scan(acc) - acc) & 255;
            const uint32_t c4 = carry * 0x01010101u;
#pragma unroll
            for (int q = 0; q < NDW; ++q)
                o[q] = add4(o[q], c4);
        } else if (ft == 3 || ft == 4) {
            // serial along x: every lane works its bytes from the left neighbour's last output; after round i the lanes
            // 0 .. i hold their final bytes (lane i's carry-in was final in round i)
            uint32_t left = 0; // the last output byte of the lane to the left
            for (int round = 0; round < 64; ++round) {
                int a = (int)left;
                // up-left of my first byte (the DPP move runs with every lane active: a disabled source lane would not be read;
                // lane 0 has no source and keeps the 0)
                int c = (int)((uint32_t)__builtin_amdgcn_update_dpp(0u, prev[NDW - 1], 0x138, 0xf, 0xf, false) >> 24); // (the builtin returns int)
#pragma unroll
                for (int q = 0; q < NDW; ++q) {
                    uint32_t w = 0;
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        const int up = (int)((prev[q] >> (8 * b)) & 255);
                        const int x = (int)((r[q] >> (8 * b)) & 255);
                        const int pred = ft == 3 ? ((a + up) >> 1) : paeth(a, up, c);
                        a = (x + pred) & 255;
                        c = up;
                        w |= (uint32_t)a << (8 * b);
                    }
                    o[q] = w;
                }
                left = __builtin_amdgcn_update_dpp(0u, (uint32_t)a, 0x138, 0xf, 0xf, false); // wave_shr:1 (lane 0 keeps 0)
                if ((round + 1) * 4 * NDW >= W) // the lanes that hold pixels are final
                    break;
            }
        } else {
            bad = 1;
            break;
        }
#pragma unroll
        for (int q = 0; q < NDW; ++q)
            prev[q] = o[q];
        uint32_t *drow = (uint32_t *)(dst + (uint64_t)y * W + x0);
#pragma unroll
        for (int q = 0; q < NDW; ++q) {
            uint32_t v = o[q];
            if (pal) {
                v = (uint32_t)lut[v & 255] | ((uint32_t)lut[(v >> 8) & 255] << 8) | ((uint32_t)lut[(v >> 16) & 255] << 16) |
                    ((uint32_t)lut[v >> 24] << 24);
            }
            if (4 * q < nmine)
                drow[q] = v;
        }
    }
    if (bad && lane == 0)
        status[f] = ABUB_PNG_E_FILTER;
}

} // namespace

extern "C" size_t abub_png_raw_stride(int W, int H)
{
    if (W <= 0 || H <= 0)
        return 0;
    return (((size_t)H * ((size_t)W + 1) + 15) & ~(size_t)15) + 16;
}

extern "C" int abub_png_decode_dev(const uint8_t *files, size_t files_bytes, const abub_png_frame *frames, int nframes,
                                   const abub_png_seg *segs, int nsegs, const uint8_t *luts, int nluts, int W, int H,
                                   uint8_t *zbuf, size_t zbuf_bytes, uint8_t *rawbuf, size_t rawbuf_bytes, uint8_t *out,
                                   size_t out_bytes, int32_t *status, void *stream)
{
    if (nframes == 0)
        return ABUB_OK;
    if (!files || !frames || !zbuf || !rawbuf || !out || !status || nframes < 0 || nsegs < 0 || nluts < 0 || (nsegs && !segs) ||
        (nluts && !luts))
        return set_err(ABUB_E_INVALID, "abub_png_decode_dev: null pointer or negative count");
    if (W < 4 || (W & 3) || W > 2048 || H < 1 || (size_t)H * ((size_t)W + 1) >= ((size_t)1 << 31))
        return set_err(ABUB_E_INVALID, "abub_png_decode_dev: width must be a multiple of 4 in [4, 2048]");
    if (((uintptr_t)files & 3) || ((uintptr_t)zbuf & 15) || ((uintptr_t)rawbuf & 15) || ((uintptr_t)out & 3))
        return set_err(ABUB_E_INVALID, "abub_png_decode_dev: files / out 4-byte, zbuf / rawbuf 16-byte aligned");
    const size_t stride = abub_png_raw_stride(W, H);
    if ((size_t)nframes * stride > rawbuf_bytes)
        return set_err(ABUB_E_INVALID, "abub_png_decode_dev: rawbuf smaller than nframes * abub_png_raw_stride(W, H)");
    hipStream_t st = (hipStream_t)stream;
    k_png_gather<<<nframes, 256, 0, st>>>(files, (uint64_t)files_bytes, frames, segs, (uint32_t)nsegs, zbuf, (uint64_t)zbuf_bytes, status);
    // ABUB_PNG_WAVES=1: one wave parses and writes a stream; default: a parsing and a writing wave per stream
    static const bool oneWave = [] { const char *e = getenv("ABUB_PNG_WAVES"); return e && atoi(e) == 1; }();
    static const int dbg = [] { const char *e = getenv("ABUB_PNG_DEBUG"); return e ? atoi(e) : 0; }(); // (measurement only)
    if ((oneWave || dbg) && dbg != 5)
        k_png_inflate<false><<<nframes, 64, 0, st>>>(zbuf, frames, (uint32_t)((size_t)H * ((size_t)W + 1)), (uint64_t)stride, rawbuf, status, dbg);
    else
        k_png_inflate<true><<<nframes, 128, 0, st>>>(zbuf, frames, (uint32_t)((size_t)H * ((size_t)W + 1)), (uint64_t)stride, rawbuf, status, dbg);
    const int ndw = (W + 255) / 256;
#define UNF(N)                                                                                                             \
    k_png_unfilter<N><<<nframes, 64, 0, st>>>(rawbuf, (uint64_t)stride, frames, luts, (uint32_t)nluts, W, H, out,         \
                                              (uint64_t)out_bytes, status)
    switch (ndw) {
    case 1: UNF(1); break;
    case 2: UNF(2); break;
    case 3: UNF(3); break;
    case 4: UNF(4); break;
    case 5: UNF(5); break;
    case 6: UNF(6); break;
    case 7: UNF(7); break;
    default: UNF(8); break;
    }
#undef UNF
    HIPCHK(hipGetLastError());
    return ABUB_OK;
}
