"""ctypes front-end of libabub_host.so: the C++ mirror of the reference's AnalyzerUnit / L3Localizer /
Trainer API (include/abub3hs/) driven the way AutoBubStart3.cpp drives it.  Fails loudly when the
native library is missing; nothing here computes on the CPU what the GPU path should compute."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "libabub_host.so")

_u8p = C.POINTER(C.c_uint8)
_u32p = C.POINTER(C.c_uint32)
_ip = C.POINTER(C.c_int)
_dp = C.POINTER(C.c_double)
_lib = None


def build(force=False):
    from . import _lib as hiplib

    hiplib.build(force)
    subprocess.check_call(["make", "-C", os.path.join(HERE, "host"), "-s"])
    return SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(SO):
            raise RuntimeError(f"{SO} is missing: run __graft_entry__.build()")
        L = C.CDLL(SO)
        L.abh_run_new.restype = C.c_void_p
        L.abh_run_free.argtypes = [C.c_void_p]
        L.abh_run_add_event.argtypes = [C.c_void_p, C.c_char_p, C.c_int, _u8p, C.c_int, C.c_int, C.c_int, _u8p]
        L.abh_train.argtypes = [C.c_void_p, C.c_int, _ip, _ip, _u8p, _u8p]
        L.abh_set_model.argtypes = [C.c_void_p, C.c_int, _u8p, _u8p, C.c_int, C.c_int, C.c_int]
        L.abh_analyze.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_char_p]
        for f in ("trig", "status", "loc_thres", "ok", "nbubbles"):
            getattr(L, "abh_last_" + f).argtypes = [C.c_void_p]
        L.abh_last_error.argtypes = [C.c_void_p]
        L.abh_last_error.restype = C.c_char_p
        L.abh_last_ndesc.argtypes = [C.c_void_p, C.c_int]
        L.abh_last_desc.argtypes = [C.c_void_p, C.c_int, C.c_int, _dp]
        L.abh_last_ndz.argtypes = [C.c_void_p, C.c_int]
        L.abh_last_dz.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.abh_last_dz.restype = C.c_float
        L.abh_last_dzdt.argtypes = [C.c_void_p, C.c_int]
        L.abh_last_dzdt.restype = C.c_float
        L.abh_last_drdt.argtypes = [C.c_void_p, C.c_int]
        L.abh_last_drdt.restype = C.c_float
        L.abh_write_header.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int]
        L.abh_event_to_file.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_int, C.c_char_p, C.c_char_p, C.c_char_p, C.c_int]
        L.abh_contours.argtypes = [_u32p, C.c_int, C.c_int, C.c_int, _ip, _ip, C.c_int, C.c_int]
        L.abh_binarize_threshold.argtypes = [_u32p, C.c_int, C.c_int]
        L.abh_entropy.argtypes = [_u32p, C.c_int, C.c_int]
        L.abh_entropy.restype = C.c_float
        L.abh_blob_stats.argtypes = [_ip, C.c_int, _dp]
        L.abh_sig_new.restype = C.c_void_p
        L.abh_sig_free.argtypes = [C.c_void_p]
        L.abh_sig_eval.argtypes = [C.c_void_p, _u32p, C.c_int, C.c_int, C.c_int, _ip]
        L.abh_sig_eval.restype = C.c_double
        _lib = L
    return _lib


DESC_KEYS = ("x", "y", "w", "h", "area", "radius", "m00", "m10", "m01", "cx", "cy")


class Run:
    """A run: in memory (events added per camera), a directory tree (kind="raw") or a zip archive (kind="zip");
    trained per camera, then analysed per (event, camera)."""

    def __init__(self, kind=None, run_folder="", image_folder="Images", image_format="cam%d_image%u.png"):
        L = lib()
        L.abh_run_open.restype = C.c_void_p
        L.abh_run_open.argtypes = [C.c_int, C.c_char_p, C.c_char_p, C.c_char_p]
        L.abh_run_events.restype = C.c_char_p
        L.abh_run_events.argtypes = [C.c_void_p]
        L.abh_run_frames.restype = C.c_char_p
        L.abh_run_frames.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
        L.abh_run_image.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, _u8p, C.c_int, _ip, _ip]
        if kind is None:
            self._h = L.abh_run_new()
        else:
            self._h = L.abh_run_open(0 if kind == "raw" else 1, run_folder.encode(), image_folder.encode(),
                                     image_format.encode())
            if not self._h:
                raise RuntimeError(f"cannot open run source {run_folder!r} (status -10 upstream)")

    def events(self):
        return [e for e in lib().abh_run_events(self._h).decode().split("\n") if e]

    def frames(self, event, cam):
        return [f for f in lib().abh_run_frames(self._h, str(event).encode(), cam).decode().split("\n") if f]

    def image(self, event, frame, cap=1 << 22):
        buf = np.empty(cap, np.uint8)
        w, h = C.c_int(), C.c_int()
        rc = lib().abh_run_image(self._h, str(event).encode(), frame.encode(), buf.ctypes.data_as(_u8p), cap,
                                 C.byref(w), C.byref(h))
        img = buf[: w.value * h.value].reshape(h.value, w.value).copy() if w.value * h.value else None
        return rc, img

    def close(self):
        if self._h:
            lib().abh_run_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def add_event(self, event, cam, frames, ok=None):
        frames = np.ascontiguousarray(frames, dtype=np.uint8)
        F, H, W = frames.shape
        okp = None
        if ok is not None:
            ok = np.ascontiguousarray(ok, dtype=np.uint8)
            okp = ok.ctypes.data_as(_u8p)
        lib().abh_run_add_event(self._h, str(event).encode(), cam, frames.ctypes.data_as(_u8p), F, W, H, okp)
        self._shape = (H, W)

    def train(self, cam, shape=None):
        H, W = shape if shape is not None else self._shape
        mu = np.zeros((H, W), np.uint8)
        sg = np.zeros((H, W), np.uint8)
        st, tss = C.c_int(), C.c_int()
        rc = lib().abh_train(self._h, cam, C.byref(st), C.byref(tss), mu.ctypes.data_as(_u8p), sg.ctypes.data_as(_u8p))
        if rc != 0:
            raise RuntimeError(lib().abh_last_error(self._h).decode())
        return st.value, tss.value, mu, sg

    def set_model(self, cam, mu, sigma, tss):
        mu = np.ascontiguousarray(mu, dtype=np.uint8)
        sigma = np.ascontiguousarray(sigma, dtype=np.uint8)
        H, W = mu.shape
        lib().abh_set_model(self._h, cam, mu.ctypes.data_as(_u8p), sigma.ctypes.data_as(_u8p), W, H, int(tss))

    def probe_frame_stats(self, event, cam, imgs):
        """Test hook (abub::AnalyzerProbe): per image the 128-bin entropy, its z-score over the images so far and the
        256-bin significance, through the private AnalyzerUnit members that the reference compiles but never calls
        (AnalyzerUnit.cpp:386-433) -> float64 [n,3]."""
        L = lib()
        L.abh_probe_frame_stats.argtypes = [C.c_void_p, C.c_char_p, C.c_int, _u8p, C.c_int, C.c_int, C.c_int, _dp]
        imgs = np.ascontiguousarray(imgs, dtype=np.uint8)
        n, H, W = imgs.shape
        out = np.zeros((n, 3), np.float64)
        rc = L.abh_probe_frame_stats(self._h, str(event).encode(), cam, imgs.ctypes.data_as(_u8p), n, W, H,
                                     out.ctypes.data_as(_dp))
        if rc != 0:
            raise RuntimeError(f"abh_probe_frame_stats rc={rc}: " + L.abh_last_error(self._h).decode())
        return out

    def run_batched(self, ncams, outdir, run_number, frame_offset, maskdir="", ngpus=1, nthreads=16, decode_threads=16,
                    batch_mb=0, shard=(0, 1)):
        """Every event of this run through the batched GPU pipeline (host/pipeline.cpp RunBatched): frames decoded (PNG files: on the GPU, abub_png_decode_dev; ABUB_GPU_DECODE=0: by host threads into pinned
        batches), detect, blocks appended to <outdir>abub3hs_<run>.txt in event order.  -> stats dict."""
        L = lib()
        L.abh_run_batched.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_char_p, C.c_char_p] + [C.c_int] * 7 + [_dp]
        st = (C.c_double * 13)()
        rc = L.abh_run_batched(self._h, ncams, maskdir.encode(), outdir.encode(), run_number.encode(), frame_offset, ngpus,
                               nthreads, decode_threads, batch_mb, shard[0], shard[1], st)
        if rc != 0:
            raise RuntimeError(f"abh_run_batched rc={rc}: " + L.abh_last_error(self._h).decode())
        keys = ("total_s", "list_s", "decode_s", "gpu_s", "write_s", "frames", "frames_failed", "batches",
                "events_per_batch", "gpus", "frames_gpu_decoded", "frames_host_decoded", "gpudecode_s")
        return dict(zip(keys, list(st)))

    def analyze(self, event, cam, maskdir=""):
        L = lib()
        staged = L.abh_analyze(self._h, str(event).encode(), cam, maskdir.encode())
        if staged == -100:
            raise RuntimeError(L.abh_last_error(self._h).decode())
        state = {"trig": L.abh_last_trig(self._h), "status": L.abh_last_status(self._h),
                 "ok": bool(L.abh_last_ok(self._h)), "loc_thres": L.abh_last_loc_thres(self._h)}
        bubbles = []
        buf = (C.c_double * 11)()
        for b in range(L.abh_last_nbubbles(self._h)):
            descs = []
            for d in range(L.abh_last_ndesc(self._h, b)):
                L.abh_last_desc(self._h, b, d, buf)
                dd = dict(zip(DESC_KEYS, list(buf)))
                for k in ("x", "y", "w", "h"):
                    dd[k] = int(dd[k])
                descs.append(dd)
            dz = [L.abh_last_dz(self._h, b, i) for i in range(L.abh_last_ndz(self._h, b))]
            bubbles.append({"desc": descs, "dz": dz, "dzdt": L.abh_last_dzdt(self._h, b),
                            "drdt": L.abh_last_drdt(self._h, b)})
        return staged, state, bubbles, L.abh_last_error(self._h).decode()


def imdecode(data, cap=1 << 22):
    L = lib()
    L.abh_imdecode.argtypes = [_u8p, C.c_int, _u8p, C.c_int, _ip, _ip]
    src = np.frombuffer(data, np.uint8)
    out = np.empty(cap, np.uint8)
    w, h = C.c_int(), C.c_int()
    rc = L.abh_imdecode(src.ctypes.data_as(_u8p), len(src), out.ctypes.data_as(_u8p), cap, C.byref(w), C.byref(h))
    if rc != 0:
        return None
    return out[: w.value * h.value].reshape(h.value, w.value).copy()


def png_walk(data, W, H, cap=4096):
    """host/pngwalk.hpp::pngWalk: None for a file the GPU decoder does not take, else (IDAT segments [(offset, length)],
    palette -> grey table (bytes) or None)"""
    L = lib()
    L.abh_png_walk.argtypes = [_u8p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_uint32), C.c_int, _ip, _ip, _u8p]
    src = np.frombuffer(data, np.uint8)
    segs = (C.c_uint32 * (2 * cap))()
    n, pal = C.c_int(), C.c_int()
    lut = np.zeros(256, np.uint8)
    if not L.abh_png_walk(src.ctypes.data_as(_u8p), len(src), W, H, segs, cap, C.byref(n), C.byref(pal), lut.ctypes.data_as(_u8p)):
        return None
    return [(segs[2 * i], segs[2 * i + 1]) for i in range(min(n.value, cap))], (lut.tobytes() if pal.value else None)


def imwrite(path, img):
    """cvlite's cv::imwrite (debug image write-out): 8-bit grey PNG, or BMP when the name ends in .bmp."""
    L = lib()
    L.abh_imwrite.argtypes = [C.c_char_p, _u8p, C.c_int, C.c_int]
    img = np.ascontiguousarray(img, dtype=np.uint8)
    return L.abh_imwrite(path.encode(), img.ctypes.data_as(_u8p), img.shape[1], img.shape[0]) == 0


def write_header(outdir, run_number, frame_offset, ncams):
    """OutputWriter::writeHeader -> <outdir>abub3hs_<run>.txt (outdir is used as a prefix, like upstream)."""
    lib().abh_write_header(outdir.encode(), run_number.encode(), frame_offset, ncams)


def event_to_file(run, event, actual_event_number, ncams, outdir, run_number, frame_offset, maskdir=""):
    """One iteration of the reference's event loop: analyse every camera, append the block to the file."""
    rc = lib().abh_event_to_file(run._h, str(event).encode(), int(actual_event_number), ncams, maskdir.encode(),
                                 outdir.encode(), run_number.encode(), frame_offset)
    if rc != 0:
        raise RuntimeError("abh_event_to_file failed")


def writer_probe(outdir, run_number, frame_offset, event, cams):
    """cams: list of (status, frame0, bubbles) with bubbles = list of descriptor-row lists (11 numbers each)."""
    L = lib()
    L.abh_writer_probe.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int, _ip, _ip, _ip, _ip, _dp]
    n = len(cams)
    status = (C.c_int * n)(*[c[0] for c in cams])
    frame0 = (C.c_int * n)(*[c[1] for c in cams])
    nbub = (C.c_int * n)(*[len(c[2]) for c in cams])
    nd = [len(b) for c in cams for b in c[2]]
    ndesc = (C.c_int * max(1, len(nd)))(*nd)
    rows = [float(v) for c in cams for b in c[2] for d in b for v in d]
    desc = (C.c_double * max(1, len(rows)))(*rows)
    L.abh_writer_probe(outdir.encode(), run_number.encode(), frame_offset, n, event, status, frame0, nbub, ndesc, desc)


# ---- host-logic probes (CPU only) ---------------------------------------------------------------
def contours_from_indices(idx, W, H):
    idx = np.ascontiguousarray(idx, dtype=np.uint32)
    npts = np.zeros(4096, np.int32)
    xy = np.zeros((1 << 18, 2), np.int32)
    n = lib().abh_contours(idx.ctypes.data_as(_u32p), len(idx), W, H, npts.ctypes.data_as(_ip),
                           xy.ctypes.data_as(_ip), len(npts), len(xy))
    if n < 0:
        raise RuntimeError("contour probe capacity exceeded")
    out, o = [], 0
    for k in range(n):
        out.append(xy[o:o + npts[k]].copy())
        o += npts[k]
    return out


def binarize_threshold(hist, P, tozero):
    hist = np.ascontiguousarray(hist, dtype=np.uint32)
    return int(lib().abh_binarize_threshold(hist.ctypes.data_as(_u32p), int(P), int(tozero)))


def entropy(hist, nbins, P):
    hist = np.ascontiguousarray(hist, dtype=np.uint32)
    return float(lib().abh_entropy(hist.ctypes.data_as(_u32p), nbins, int(P)))


def blob_stats(xy):
    xy = np.ascontiguousarray(xy, dtype=np.int32)
    out = (C.c_double * 8)()
    lib().abh_blob_stats(xy.ctypes.data_as(_ip), len(xy), out)
    return dict(zip(("x", "y", "w", "h", "area", "m00", "m10", "m01"), list(out)))


def best_match(num, wsum2, tmpl):
    L = lib()
    u64p = C.POINTER(C.c_uint64)
    L.abh_best_match.argtypes = [u64p, u64p, C.c_int, C.c_int, _u8p, C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    num = np.ascontiguousarray(num, dtype=np.uint64)
    wsum2 = np.ascontiguousarray(wsum2, dtype=np.uint64)
    tmpl = np.ascontiguousarray(tmpl, dtype=np.uint8)
    rh, rw = num.shape
    bx, by = C.c_float(), C.c_float()
    L.abh_best_match(num.ctypes.data_as(u64p), wsum2.ctypes.data_as(u64p), rw, rh, tmpl.ctypes.data_as(_u8p),
                     tmpl.shape[1], tmpl.shape[0], C.byref(bx), C.byref(by))
    return bx.value, by.value


class Significance:
    def __init__(self, tss):
        self._h = lib().abh_sig_new()
        self.tss = tss
        self.loc_thres = C.c_int(3)

    def __call__(self, hist, P, store):
        hist = np.ascontiguousarray(hist, dtype=np.uint32)
        return float(lib().abh_sig_eval(self._h, hist.ctypes.data_as(_u32p), int(P), 1 if store else 0, self.tss,
                                        C.byref(self.loc_thres)))

    def __del__(self):
        try:
            lib().abh_sig_free(self._h)
        except Exception:
            pass


class Pipeline:
    """Run-level batched detect over an HBM-resident slab [E][C][F][H][W] (host/pipeline.cpp)."""

    def __init__(self, device, W, H, F, E, ncams, tss, nthreads=16, maskdir=""):
        L = lib()
        L.abh_pipe_new.restype = C.c_void_p
        L.abh_pipe_new.argtypes = [C.c_int] * 6 + [_ip, C.c_int, C.c_char_p]
        L.abh_pipe_free.argtypes = [C.c_void_p]
        L.abh_pipe_run.argtypes = [C.c_void_p] * 5
        L.abh_pipe_error.restype = C.c_char_p
        L.abh_pipe_result.argtypes = [C.c_void_p, C.c_int, _ip]
        L.abh_pipe_ndesc.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.abh_pipe_desc.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, _dp]
        L.abh_pipe_dzdt.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.abh_pipe_dzdt.restype = C.c_float
        L.abh_pipe_drdt.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.abh_pipe_drdt.restype = C.c_float
        L.abh_pipe_stack_error.argtypes = [C.c_void_p, C.c_int]
        L.abh_pipe_stack_error.restype = C.c_char_p
        L.abh_pipe_timing.argtypes = [C.c_void_p, _dp]
        t = (C.c_int * len(tss))(*tss)
        self.S = E * ncams
        self._h = L.abh_pipe_new(device, W, H, F, E, ncams, t, nthreads, maskdir.encode())
        if not self._h:
            raise RuntimeError("abh_pipe_new failed (see stderr)")

    def close(self):
        if self._h:
            lib().abh_pipe_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_sigma(self, sigma):
        """sigma image(s) (not 6 sigma) on the device: needed only by stacks that fall back to the drop-in path."""
        lib().abh_pipe_set_sigma.argtypes = [C.c_void_p, C.c_void_p]
        lib().abh_pipe_set_sigma(self._h, sigma.data_ptr() if hasattr(sigma, "data_ptr") else int(sigma))

    def run(self, frames, mu, sigma6, stream=0, sigma=None):
        """frames/mu/sigma6 (and optionally sigma, needed only if a stack falls back to the drop-in path for the
        bellows veto): device pointers (ints) or torch tensors."""
        p = [x.data_ptr() if hasattr(x, "data_ptr") else int(x) for x in (frames, mu, sigma6)]
        if sigma is not None:
            lib().abh_pipe_set_sigma.argtypes = [C.c_void_p, C.c_void_p]
            lib().abh_pipe_set_sigma(self._h, sigma.data_ptr() if hasattr(sigma, "data_ptr") else int(sigma))
        rc = lib().abh_pipe_run(self._h, p[0], p[1], p[2], stream)
        if rc != 0:
            raise RuntimeError("pipeline: " + lib().abh_pipe_error().decode())

    def run_host(self, frames_host, mu, sigma6):
        """Streamed mode: `frames_host` is a HOST tensor / pointer ([E][C][F][H][W], pinned for full PCIe rate);
        stack groups are uploaded and processed in a pipeline."""
        L = lib()
        L.abh_pipe_run_host.argtypes = [C.c_void_p] * 4
        p = [x.data_ptr() if hasattr(x, "data_ptr") else int(x) for x in (frames_host, mu, sigma6)]
        if L.abh_pipe_run_host(self._h, p[0], p[1], p[2]) != 0:
            raise RuntimeError("pipeline: " + L.abh_pipe_error().decode())

    def timing(self):
        out = (C.c_double * 12)()
        rounds = lib().abh_pipe_timing(self._h, out)
        return dict(zip(("stage1_ms", "stage2_ms", "stage3_ms", "stage4_ms", "total_ms", "s3_gpu_ms", "s3_list_ms",
                         "s3_bucket_ms", "pairs", "trigger_jobs", "dropin_stacks", "jobs_completed_on_demand"), list(out)),
                    rounds=rounds)

    def result(self, s):
        L = lib()
        o = (C.c_int * 6)()
        L.abh_pipe_result(self._h, s, o)
        state = {"trig": o[1], "status": o[2], "ok": bool(o[4]), "loc_thres": o[3]}
        bubbles = []
        buf = (C.c_double * 11)()
        for b in range(o[5]):
            descs = []
            for d in range(L.abh_pipe_ndesc(self._h, s, b)):
                L.abh_pipe_desc(self._h, s, b, d, buf)
                dd = dict(zip(DESC_KEYS, list(buf)))
                for k in ("x", "y", "w", "h"):
                    dd[k] = int(dd[k])
                descs.append(dd)
            bubbles.append({"desc": descs, "dzdt": L.abh_pipe_dzdt(self._h, s, b), "drdt": L.abh_pipe_drdt(self._h, s, b)})
        return o[0], state, bubbles, L.abh_pipe_stack_error(self._h, s).decode()

    def summary(self):
        """(staged, trig, nbubbles) for every stack -- cheap fingerprint of a run."""
        L = lib()
        o = (C.c_int * 6)()
        out = []
        for s in range(self.S):
            L.abh_pipe_result(self._h, s, o)
            out.append((o[0], o[1], o[5]))
        return out


class PipelineRing:
    """N pipeline objects, each driven by its own host thread: batch k runs on pipeline k % N, so the host stages of
    one batch (trigger state machines, contour tracing, tracking) run while the GPU is busy with the kernels of the
    next.  Every batch still goes through the complete path; only their stages interleave.  N = 1 is a plain loop."""

    def __init__(self, n, device, W, H, F, E, ncams, tss, nthreads=16, maskdir=""):
        self.device = device
        self.pipes = [Pipeline(device, W, H, F, E, ncams, tss, nthreads=nthreads, maskdir=maskdir) for _ in range(max(1, n))]

    def close(self):
        for p in self.pipes:
            p.close()

    def run_batches(self, batches, mu, sigma6, stream=0, on_done=None, host=False):
        """batches: sequence of frame slabs (device tensors / pointers, each [E][C][F][H][W]); with host=True they are
        HOST slabs (pinned for full PCIe rate) and every pipeline streams its batch in (Pipeline.run_host), so the upload
        of one run overlaps the detect stages of the previous one (BASELINE configs[4]: many runs streamed host -> HBM).
        Returns the per-batch timing dicts in completion order; `on_done(k, pipeline)` is called on the driving thread
        right after batch k finished, while its results are still the pipeline's current ones."""
        import threading

        n = len(self.pipes)
        tms, errs = [], []
        lock = threading.Lock()

        def drive(i):
            try:
                try:
                    import torch

                    torch.cuda.set_device(self.device)
                except ImportError:
                    pass
                for k in range(i, len(batches), n):
                    if host:
                        self.pipes[i].run_host(batches[k], mu, sigma6)
                    else:
                        self.pipes[i].run(batches[k], mu, sigma6, stream)
                    if on_done:
                        on_done(k, self.pipes[i])
                    with lock:
                        tms.append(self.pipes[i].timing())
            except BaseException as e:  # noqa: BLE001 -- re-raised on the caller's thread
                errs.append(e)

        if n == 1:
            drive(0)
        else:
            th = [threading.Thread(target=drive, args=(i,)) for i in range(n)]
            for t in th:
                t.start()
            for t in th:
                t.join()
        if errs:
            raise errs[0]
        return tms
