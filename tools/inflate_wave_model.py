#!/usr/bin/env python3
"""Lane-level model (numpy, 64 "lanes") of the wave-per-stream inflate that autobub3hs_amd/csrc/abub_png.hip runs on the GPU:
the same phases and the same tables, so the algorithm can be checked against zlib where there is no GPU.

  per iteration: every lane decodes ONE token (literal / length+distance / end of block) at bit offset ip + lane of the
  input (speculatively: only the lanes on the chain s, s + n_s, ... are real token starts) -> the chain is walked with
  readlane -> exclusive scan of the output lengths -> literals are written, short matches whose source lies before this
  iteration's first match are copied by their own lanes, the others (dependent / overlapping / long) one after the other
  by the whole wave.

The kernel differs in engineering, not in the algorithm: two windows per iteration, table entries that carry base and
extra-bit count, long codes resolved only when the chain visits their lane, a parsing and a writing wave per stream.

Usage: python tools/inflate_wave_model.py file.png [...]   (compares with zlib.decompress, prints token statistics)"""
import struct
import sys
import zlib

import numpy as np

NW = 64
LIT_ROOT = 10
DIST_ROOT = 9
K_LIT, K_LEN, K_EOB, K_LONG, K_BAD = 0, 1, 2, 3, 4
SHORT = 8  # matches up to this length are copied by their own lane


class Corrupt(Exception):
    pass


def bitrev(v, n):
    r = 0
    for _ in range(n):
        r = (r << 1) | (v & 1)
        v >>= 1
    return r


class Canon:
    """canonical code of one alphabet: first[l], count[l], offset[l], sorted symbols, root table"""

    def __init__(self, lens, root, kind_of, allow_incomplete_single):
        lens = list(lens)
        self.count = [0] * 16
        for l in lens:
            self.count[l] += 1
        self.count[0] = 0
        left = 1
        for l in range(1, 16):
            left = (left << 1) - self.count[l]
            if left < 0:
                raise Corrupt("over-subscribed code")
        mx = max([l for l in range(16) if self.count[l]] or [0])
        self.empty = mx == 0
        if left > 0 and not self.empty and not (allow_incomplete_single and mx == 1):
            raise Corrupt("incomplete code")
        self.first = [0] * 16
        self.offset = [0] * 16
        code = 0
        off = 0
        for l in range(1, 16):
            code = (code + self.count[l - 1]) << 1
            self.first[l] = code
            self.offset[l] = off
            off += self.count[l]
        self.sorted = [s for l in range(1, 16) for s in range(len(lens)) if lens[s] == l]
        self.root = root
        # root table, filled per ENTRY (as the kernel does: one lane per entry, canonical decode of the entry's bits)
        self.tab = np.zeros(1 << root, dtype=np.uint32)
        for e in range(1 << root):
            rev = bitrev(e, root)
            ent = K_BAD << 4
            for l in range(1, root + 1):
                c = rev >> (root - l)
                idx = c - self.first[l]
                if 0 <= idx < self.count[l]:
                    sym = self.sorted[self.offset[l] + idx]
                    ent = l | (kind_of(sym) << 4) | (sym << 8)
                    break
            else:
                # no code of <= root bits is a prefix of these bits: a longer one may be
                if any(self.count[l] for l in range(root + 1, 16)):
                    ent = K_LONG << 4
            self.tab[e] = ent

    def slow(self, w, kind_of):
        """codes longer than the root: canonical compare per length (w = next bits, LSB first)"""
        rev15 = bitrev(int(w) & 0x7FFF, 15)
        for l in range(self.root + 1, 16):
            c = rev15 >> (15 - l)
            idx = c - self.first[l]
            if 0 <= idx < self.count[l]:
                sym = self.sorted[self.offset[l] + idx]
                return l | (kind_of(sym) << 4) | (sym << 8)
        return K_BAD << 4


def lit_kind(sym):
    return K_LIT if sym < 256 else K_EOB if sym == 256 else K_LEN if sym < 286 else K_BAD


def dist_kind(sym):
    return K_LIT if sym < 30 else K_BAD


def len_base_extra(i):  # i = symbol - 257
    if i < 8:
        return 3 + i, 0
    if i == 28:
        return 258, 0
    e = (i - 4) >> 2
    return 3 + ((4 + (i & 3)) << e), e


def dist_base_extra(d):
    if d < 4:
        return 1 + d, 0
    e = (d - 2) >> 1
    return 1 + ((2 + (d & 1)) << e), e


class Stats:
    def __init__(self):
        self.iters = self.tokens = self.lits = self.matches = self.dep = self.longm = self.slow_iters = self.blocks = 0


def inflate_wave(z, expect, stats=None):
    """zlib stream z -> bytes, with the kernel's phases.  `expect` = output size the caller has room for."""
    st = stats or Stats()
    if len(z) < 6:
        raise Corrupt("short stream")
    if (z[0] & 15) != 8 or (z[0] >> 4) > 7 or ((z[0] << 8) | z[1]) % 31 or (z[1] & 0x20):
        raise Corrupt("zlib header")
    zb = np.frombuffer(bytes(z) + b"\0" * 32, dtype=np.uint8)
    nbits_total = len(z) * 8
    out = np.zeros(expect + 512, dtype=np.uint8)
    op = 0
    ip = 16

    def peek(pos, n):
        v = 0
        for k in range(n):
            b = pos + k
            v |= ((int(zb[b >> 3]) >> (b & 7)) & 1) << k
        return v

    def window64(pos):  # per-lane 64-bit window (array of python ints is too slow: use u64 from 9 bytes)
        byte = pos >> 3
        sh = (pos & 7).astype(np.uint64)
        w = np.zeros(pos.shape, dtype=np.uint64)
        for k in range(8):
            w |= zb[np.minimum(byte + k, len(zb) - 1)].astype(np.uint64) << np.uint64(8 * k)
        hi = zb[np.minimum(byte + 8, len(zb) - 1)].astype(np.uint64)
        return (w >> sh) | np.where(sh > 0, hi << (np.uint64(64) - np.maximum(sh, np.uint64(1))), np.uint64(0))

    lanes = np.arange(NW)
    final = False
    while not final:
        if ip + 3 > nbits_total:
            raise Corrupt("truncated")
        final = bool(peek(ip, 1))
        btype = peek(ip + 1, 2)
        ip += 3
        st.blocks += 1
        if btype == 3:
            raise Corrupt("block type 3")
        if btype == 0:
            ip = (ip + 7) & ~7
            if ip + 32 > nbits_total:
                raise Corrupt("truncated")
            ln, nln = peek(ip, 16), peek(ip + 16, 16)
            if ln ^ nln != 0xFFFF:
                raise Corrupt("stored length")
            ip += 32
            if ip + 8 * ln > nbits_total:
                raise Corrupt("truncated")
            if op + ln > expect:
                raise Corrupt("too much output")
            out[op:op + ln] = zb[ip >> 3:(ip >> 3) + ln]
            op += ln
            ip += 8 * ln
            continue
        if btype == 1:
            ll = [8] * 144 + [9] * 112 + [7] * 24 + [8] * 8
            dl = [5] * 32
        else:
            if ip + 14 > nbits_total:
                raise Corrupt("truncated")
            hlit, hdist, hclen = peek(ip, 5) + 257, peek(ip + 5, 5) + 1, peek(ip + 10, 4) + 4
            ip += 14
            if hlit > 286 or hdist > 30:
                raise Corrupt("too many symbols")
            order = [16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15]
            cl = [0] * 19
            for i in range(hclen):
                cl[order[i]] = peek(ip, 3)
                ip += 3
            ct = Canon(cl, 7, lambda s: K_LIT, False)
            if ct.empty:
                raise Corrupt("no code length codes")
            L = []
            while len(L) < hlit + hdist:
                if ip > nbits_total:
                    raise Corrupt("truncated")
                e = int(ct.tab[peek(ip, 7)])
                if (e >> 4) & 7 != K_LIT:
                    raise Corrupt("bad code length code")
                ip += e & 15
                s = e >> 8
                if s < 16:
                    L.append(s)
                elif s == 16:
                    if not L:
                        raise Corrupt("repeat without a length")
                    L += [L[-1]] * (3 + peek(ip, 2))
                    ip += 2
                elif s == 17:
                    L += [0] * (3 + peek(ip, 3))
                    ip += 3
                else:
                    L += [0] * (11 + peek(ip, 7))
                    ip += 7
            if len(L) > hlit + hdist:
                raise Corrupt("repeat overflows")
            ll, dl = L[:hlit], L[hlit:]
            if ll[256] == 0:
                raise Corrupt("no end-of-block code")
        lt = Canon(ll, LIT_ROOT, lit_kind, False if btype == 2 else False)
        dt = Canon(dl, DIST_ROOT, dist_kind, True)
        # ---- the token loop --------------------------------------------------------------------------------
        s = 0  # chain start inside the window
        while True:
            st.iters += 1
            if ip > nbits_total:
                raise Corrupt("truncated")
            pos = ip + lanes
            w = window64(pos)
            e = lt.tab[(w & np.uint64((1 << LIT_ROOT) - 1)).astype(np.int64)]
            kind = (e >> 4) & 7
            if (kind == K_LONG).any():
                st.slow_iters += 1
                for j in np.nonzero(kind == K_LONG)[0]:
                    e[j] = lt.slow(w[j], lit_kind)
                kind = (e >> 4) & 7
            nb = (e & 15).astype(np.int64)
            val = (e >> 8).astype(np.int64)
            olen = np.where(kind == K_LIT, 1, 0).astype(np.int64)
            dist = np.zeros(NW, dtype=np.int64)
            is_len = kind == K_LEN
            if is_len.any():
                i = np.where(is_len, val - 257, 0)
                eb = np.where(i < 8, 0, np.where(i == 28, 0, (i - 4) >> 2))
                base = np.where(i < 8, 3 + i, np.where(i == 28, 258, 3 + ((4 + (i & 3)) << eb)))
                ln = base + ((w >> nb.astype(np.uint64)).astype(np.int64) & ((1 << eb) - 1))
                nb2 = nb + eb
                w2 = w >> nb2.astype(np.uint64)
                de = dt.tab[(w2 & np.uint64((1 << DIST_ROOT) - 1)).astype(np.int64)]
                dk = (de >> 4) & 7
                for j in np.nonzero(is_len & (dk == K_LONG))[0]:
                    de[j] = dt.slow(w2[j], dist_kind)
                dk = (de >> 4) & 7
                dsym = (de >> 8).astype(np.int64)
                dnb = (de & 15).astype(np.int64)
                deb = np.where(dsym < 4, 0, (dsym - 2) >> 1)
                dbase = np.where(dsym < 4, 1 + dsym, 1 + ((2 + (dsym & 1)) << deb))
                nb3 = nb2 + dnb
                dd = dbase + ((w >> nb3.astype(np.uint64)).astype(np.int64) & ((1 << deb) - 1))
                nb4 = nb3 + deb
                bad_d = is_len & (dk != K_LIT)
                olen = np.where(is_len, ln, olen)
                dist = np.where(is_len, dd, 0)
                nb = np.where(is_len, nb4, nb)
                kind = np.where(bad_d, K_BAD, kind)
            # ---- chain walk (readlane per hop) ----
            valid = np.zeros(NW, dtype=bool)
            p = s
            eob = False
            while p < NW:
                valid[p] = True
                if kind[p] == K_BAD:
                    raise Corrupt("invalid code")
                if kind[p] == K_EOB:
                    eob = True
                    p += int(nb[p])
                    break
                p += int(nb[p])
            if eob:
                ip += p
            else:
                ip += NW
                s = p - NW
            if ip > nbits_total:
                raise Corrupt("truncated")
            ntok = int(valid.sum())
            st.tokens += ntok
            # ---- exclusive scan of output lengths over valid lanes ----
            ol = np.where(valid, olen, 0)
            off = np.cumsum(ol) - ol
            total = int(ol.sum())
            if op + total > expect:
                raise Corrupt("too much output")
            ismatch = valid & (kind == K_LEN)
            islit = valid & (kind == K_LIT)
            st.lits += int(islit.sum())
            st.matches += int(ismatch.sum())
            if (ismatch & (dist > op + off)).any():
                raise Corrupt("distance too far back")
            # literals
            out[op + off[islit]] = val[islit].astype(np.uint8)
            if ismatch.any():
                m0 = int(off[ismatch][0])  # output offset of this iteration's first match
                src_end_rel = off - dist + olen  # relative to op
                first = np.zeros(NW, dtype=bool)
                first[np.nonzero(ismatch)[0][0]] = True
                dep = ismatch & (np.where(first, dist < olen, src_end_rel > m0))
                longm = ismatch & ~dep & (olen > SHORT)
                indep = ismatch & ~dep & ~longm
                # independent short matches: each lane reads its bytes, then writes them (all reads before all writes)
                idx = np.nonzero(indep)[0]
                got = [out[op + off[j] - dist[j]: op + off[j] - dist[j] + olen[j]].copy() for j in idx]
                for j, g in zip(idx, got):
                    out[op + off[j]: op + off[j] + olen[j]] = g
                # the rest in token order, by the whole wave: byte k <- src[k mod dist]
                for j in np.nonzero(dep | longm)[0]:
                    d, L, o = int(dist[j]), int(olen[j]), op + int(off[j])
                    k = np.arange(L)
                    out[o + k] = out[o - d + (k % d)] if d < L else out[o - d + k]
                st.dep += int(dep.sum())
                st.longm += int(longm.sum())
            op += total
            if eob:
                break
    # trailer: Adler-32 of the output, big-endian, at the next byte boundary
    ip = (ip + 7) & ~7
    if ip + 32 > nbits_total:
        raise Corrupt("truncated (no checksum)")
    want = struct.unpack(">I", bytes(zb[ip >> 3:(ip >> 3) + 4]))[0]
    res = bytes(out[:op])
    if zlib.adler32(res) != want:
        raise Corrupt("adler32")
    return res


def idat_of(png):
    o, out = 8, b""
    while o + 12 <= len(png):
        ln, = struct.unpack(">I", png[o:o + 4])
        if png[o + 4:o + 8] == b"IDAT":
            out += png[o + 8:o + 8 + ln]
        o += 12 + ln
    return out


def main():
    for path in sys.argv[1:]:
        data = open(path, "rb").read()
        z = idat_of(data) if data[:4] == b"\x89PNG" else data
        ref = zlib.decompress(z)
        limit = int(sys.argv[0] and 0) or len(ref)
        st = Stats()
        got = inflate_wave(z, limit, st)
        print(path, "OK" if got == ref else "MISMATCH", len(ref), "bytes;", st.blocks, "blocks,", st.iters, "iterations,",
              f"{st.tokens / max(st.iters, 1):.2f} tokens/iter, {len(ref) / max(st.iters, 1):.1f} bytes/iter; literals {st.lits}, matches {st.matches} "
              f"(dependent {st.dep}, long {st.longm}), iterations with a long code {st.slow_iters}")


if __name__ == "__main__":
    main()
