"""Thin Python mirror of the stateless device-pointer launchers (include/abub_hip.h layer A).

torch is plumbing only: it owns the HBM allocations and the stream the kernels are launched on.
"""
import torch

from . import _lib


def _ptr(t):
    return None if t is None else t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _lib.AbubError("device-pointer launcher called with a CPU tensor (no CPU fallback)")
        if t is not None and not t.is_contiguous():
            raise _lib.AbubError("tensor must be contiguous")


def sigma6(sigma):
    _need_cuda(sigma)
    out = torch.empty_like(sigma)
    _lib.check(_lib.lib().abub_sigma6_dev(_ptr(sigma), _ptr(out), sigma.numel(), _stream()), "abub_sigma6_dev")
    return out


def stack_jobs(nstacks, F, first, count, ref_offset, nmodels, device):
    jobs = torch.empty((nstacks * count, 4), dtype=torch.int32, device=device)
    _lib.check(_lib.lib().abub_fill_stack_jobs_dev(_ptr(jobs), nstacks, F, first, count, ref_offset,
                                                   nmodels, _stream()), "abub_fill_stack_jobs_dev")
    return jobs


def make_jobs(rows, device):
    """rows: iterable of (cur, ref, model, out)."""
    return torch.tensor(list(rows), dtype=torch.int64).to(torch.int32).reshape(-1, 4).to(device)


def diff_hist(frames, sigma6_, jobs, W, H, store=False, rows_per_chunk=0, hist=None, diff=None, chain=None):
    """K2. frames: u8 [..,H,W]; sigma6_: u8 [nmodels,H,W]; jobs: int32 [n,4] -> (hist [n,256] i32, diff|None).
    chain=(chain_len, chain_stride): trigger-only form with the stack-structure hint (abub_diff_hist_chained_dev)."""
    _need_cuda(frames, sigma6_, jobs)
    n = jobs.shape[0]
    if hist is None:
        hist = torch.empty((n, 256), dtype=torch.int32, device=frames.device)
    if store and diff is None:
        diff = torch.empty((n, H, W), dtype=torch.uint8, device=frames.device)
    if chain is not None:
        assert rows_per_chunk == 0
        if store:
            _lib.check(_lib.lib().abub_diff_hist_chained_store_dev(_ptr(frames), _ptr(sigma6_), _ptr(jobs), n, W, H,
                                                                   _ptr(hist), _ptr(diff), int(chain[0]), int(chain[1]),
                                                                   _stream()), "abub_diff_hist_chained_store_dev")
            return hist, diff
        _lib.check(_lib.lib().abub_diff_hist_chained_dev(_ptr(frames), _ptr(sigma6_), _ptr(jobs), n, W, H, _ptr(hist),
                                                         int(chain[0]), int(chain[1]), _stream()),
                   "abub_diff_hist_chained_dev")
        return hist, None
    _lib.check(_lib.lib().abub_diff_hist_dev(_ptr(frames), _ptr(sigma6_), _ptr(jobs), n, W, H, _ptr(hist),
                                             _ptr(diff) if store else None, rows_per_chunk, _stream()),
               "abub_diff_hist_dev")
    return hist, (diff if store else None)


def diff_hist_deferred(frames, sigma6_, jobs, W, H, chain):
    """Trigger-only chained K2 with the row machine left out (abub_diff_hist_chained_deferred_dev): -> (hist [n,256] i32,
    state) where state = (pieces, npieces, incomplete u8 [n]); histograms of jobs with incomplete != 0 are not final
    until diff_hist_pieces() has completed them."""
    _need_cuda(frames, sigma6_, jobs)
    n = jobs.shape[0]
    hist = torch.empty((n, 256), dtype=torch.int32, device=frames.device)
    cap = int(_lib.lib().abub_k2_pieces_cap(n, W, H))
    pieces = torch.empty((max(cap, 1), 2), dtype=torch.int32, device=frames.device)
    npieces = torch.zeros((1,), dtype=torch.int32, device=frames.device)
    incomplete = torch.empty((n,), dtype=torch.uint8, device=frames.device)
    _lib.check(_lib.lib().abub_diff_hist_chained_deferred_dev(_ptr(frames), _ptr(sigma6_), _ptr(jobs), n, W, H, _ptr(hist),
                                                              int(chain[0]), int(chain[1]), _ptr(pieces), cap, _ptr(npieces),
                                                              _ptr(incomplete), _stream()),
               "abub_diff_hist_chained_deferred_dev")
    return hist, (pieces, npieces, incomplete)


def diff_hist_pieces(frames, sigma6_, jobs, W, H, hist, state, want):
    """Completes the jobs with want[job] != 0 (u8 [n]) of a diff_hist_deferred() launch: row machine on their handed-over
    rows, histograms finalised in place."""
    _need_cuda(frames, sigma6_, jobs, want)
    pieces, npieces, _ = state
    _lib.check(_lib.lib().abub_diff_hist_pieces_dev(_ptr(frames), _ptr(sigma6_), _ptr(jobs), jobs.shape[0], W, H, _ptr(hist),
                                                    _ptr(pieces), _ptr(npieces), _ptr(want), _stream()),
               "abub_diff_hist_pieces_dev")
    return hist


def k2_set_option(name, value):
    """Run-time K2 launcher knob ("bound", "chain", "budget", "pf"); results never depend on them."""
    _lib.check(_lib.lib().abub_k2_set_option(name.encode(), int(value)), "abub_k2_set_option")


def bound_counts(stream=None):
    """(row pieces handed over to the row machine, global suspect-list entries) of the last bound-and-verify launch on
    `stream` (default: torch's current stream).  Waits for the stream."""
    import ctypes as C

    st = torch.cuda.current_stream().cuda_stream if stream is None else stream
    out = (C.c_uint32 * 2)()
    _lib.check(_lib.lib().abub_bound_counts_dev(st, out), "abub_bound_counts_dev")
    return int(out[0]), int(out[1])


def diff_roi(slab, cur, ref, sigma6_, W, H, roi):
    """ProcessFrame ROI overload on frames `cur`,`ref` (indices, ref >= cur... see header) of one slab."""
    _need_cuda(slab, sigma6_)
    rx, ry, rw, rh = roi
    diff = torch.empty((H, W), dtype=torch.uint8, device=slab.device)
    hist = torch.empty((256,), dtype=torch.int32, device=slab.device)
    P = W * H
    base = slab.data_ptr()
    _lib.check(_lib.lib().abub_diff_roi_dev(base + cur * P, base + ref * P, _ptr(sigma6_), W, H, rx, ry, rw, rh,
                                            _ptr(diff), _ptr(hist), _stream()), "abub_diff_roi_dev")
    return diff, hist


def train(frames, W, H, idx=None):
    _need_cuda(frames, idx)
    N = frames.shape[0] if idx is None else idx.numel()
    mu = torch.empty((H, W), dtype=torch.uint8, device=frames.device)
    sg = torch.empty((H, W), dtype=torch.uint8, device=frames.device)
    _lib.check(_lib.lib().abub_train_dev(_ptr(frames), _ptr(idx), N, W, H, _ptr(mu), _ptr(sg), _stream()),
               "abub_train_dev")
    return mu, sg


def pair_hist(frames, pairs, W, H):
    _need_cuda(frames, pairs)
    n = pairs.shape[0]
    hist = torch.empty((n, 256), dtype=torch.int32, device=frames.device)
    _lib.check(_lib.lib().abub_pair_hist_dev(_ptr(frames), _ptr(pairs), n, W, H, _ptr(hist), _stream()),
               "abub_pair_hist_dev")
    return hist


def posttrig(frames, mu, sigma6_, jobs, W, H, store=True):
    _need_cuda(frames, mu, sigma6_, jobs)
    n = jobs.shape[0]
    hist = torch.empty((n, 256), dtype=torch.int32, device=frames.device)
    img = torch.empty((n, H, W), dtype=torch.uint8, device=frames.device) if store else None
    _lib.check(_lib.lib().abub_posttrig_dev(_ptr(frames), _ptr(mu), _ptr(sigma6_), _ptr(jobs), n, W, H,
                                            _ptr(hist), _ptr(img), _stream()), "abub_posttrig_dev")
    return hist, img


def fg_compact(img, thr, cap):
    """img: u8 [n,H,W]; thr: int32 [n] -> (idx int32 [n,cap], count int32 [n])."""
    _need_cuda(img, thr)
    n, H, W = img.shape
    idx = torch.empty((n, cap), dtype=torch.int32, device=img.device)
    cnt = torch.empty((n,), dtype=torch.int32, device=img.device)
    _lib.check(_lib.lib().abub_fg_compact_dev(_ptr(img), n, W, H, _ptr(thr), _ptr(idx), cap, _ptr(cnt),
                                              _stream()), "abub_fg_compact_dev")
    return idx, cnt


# ---- PNG frames decoded on the GPU (abub_png_decode_dev) --------------------------------------------------------
def png_parse(data, W, H):
    """What the host does per file before the upload: walk the chunks of a PNG, -> (idat segments [(offset, length)],
    palette->grey table (bytes, 256) or None) for an 8-bit grey / 8-bit palette image of W x H without interlace, or None
    for anything the GPU path does not take (the caller decodes such a file on the host)."""
    import struct
    if len(data) < 33 or data[:8] != b"\x89PNG\r\n\x1a\n":
        return None
    o, segs, pal, hdr, end = 8, [], None, None, False
    while not end and o + 12 <= len(data):
        ln, = struct.unpack(">I", data[o:o + 4])
        typ = data[o + 4:o + 8]
        if o + 12 + ln > len(data):
            return None
        if typ == b"IHDR" and ln >= 13:
            hdr = struct.unpack(">IIBBBBB", data[o + 8:o + 21])
        elif typ == b"PLTE":
            pal = data[o + 8:o + 8 + ln]
        elif typ == b"IDAT":
            segs.append((o + 8, ln))
        elif typ == b"IEND":
            end = True
        o += 12 + ln
    if hdr is None or not segs or hdr[0] != W or hdr[1] != H or hdr[2] != 8 or hdr[3] not in (0, 3) or hdr[6] != 0:
        return None
    lut = None
    if hdr[3] == 3:
        n = min(len(pal or b"") // 3, 256)
        lut = bytes(((pal[3 * i] * 9797 + pal[3 * i + 1] * 19234 + pal[3 * i + 2] * 3737 + 16384) >> 15) & 255 if i < n else 0
                    for i in range(256))
    return segs, lut


def png_decode(files, W, H, device="cuda:0"):
    """files: list of bytes objects (PNG files) -> (frames u8 [n, H, W] on the device, status i32 [n] on the host).  A file
    png_parse() refuses gets status -1 and a frame left untouched (zeros)."""
    import numpy as np
    n = len(files)
    frames_np = np.zeros((max(n, 1), 8), dtype=np.uint32)
    segs, luts, blob, zoff = [], [], bytearray(), 0
    status_pre = np.zeros(n, dtype=np.int32)
    P = W * H
    for i, data in enumerate(files):
        parsed = png_parse(data, W, H)
        base = len(blob)
        if parsed is None:
            status_pre[i] = -1
            frames_np[i] = (len(segs), 0, zoff, 0, 0xFFFFFFFF, 0, (i * P) & 0xFFFFFFFF, (i * P) >> 32)
            zoff += 16
            continue
        sg, lut = parsed
        zlen = sum(l for _, l in sg)
        li = 0xFFFFFFFF
        if lut is not None:
            li = len(luts)
            luts.append(lut)
        frames_np[i] = (len(segs), len(sg), zoff, zlen, li, 0, (i * P) & 0xFFFFFFFF, (i * P) >> 32)
        segs += [(base + o, l) for o, l in sg]
        blob += data
        blob += b"\0" * ((-len(blob)) % 4)
        zoff += ((zlen + 15) & ~15) + 16
    blob += b"\0" * 8
    dev = torch.device(device)
    d_files = torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(dev)
    d_frames = torch.from_numpy(frames_np.view(np.int32).copy()).to(dev)
    d_segs = torch.tensor(segs if segs else [(0, 0)], dtype=torch.int64).to(torch.int32).to(dev) if True else None
    d_luts = torch.frombuffer(bytearray(b"".join(luts) if luts else bytes(256)), dtype=torch.uint8).to(dev)
    stride = int(_lib.lib().abub_png_raw_stride(W, H))
    d_z = torch.empty((max(zoff, 16),), dtype=torch.uint8, device=dev)
    d_raw = torch.empty((max(n, 1) * stride,), dtype=torch.uint8, device=dev)
    out = torch.zeros((max(n, 1), H, W), dtype=torch.uint8, device=dev)
    status = torch.full((max(n, 1),), 99, dtype=torch.int32, device=dev)
    _lib.check(_lib.lib().abub_png_decode_dev(_ptr(d_files), d_files.numel(), _ptr(d_frames), n, _ptr(d_segs), len(segs),
                                              _ptr(d_luts), len(luts), W, H, _ptr(d_z), d_z.numel(), _ptr(d_raw), d_raw.numel(),
                                              _ptr(out), out.numel(), _ptr(status), _stream()), "abub_png_decode_dev")
    st = status.cpu().numpy()[:n].copy()
    st[status_pre != 0] = -1
    return out[:n], st
