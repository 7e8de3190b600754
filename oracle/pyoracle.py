"""ctypes front-end of the CPU ORACLE (oracle/abub_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never by the product package (autobub3hs_amd/).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libabub_oracle.so")
CFLAGS_NOTE = "gcc -O3 -march=x86-64-v3 -ffp-contract=off"  # what oracle/Makefile builds with (quoted by bench.py)


def build(force=False):
    src = [os.path.join(_HERE, f) for f in ("abub_oracle.c", "abub_oracle.h", "Makefile")]
    if force or not os.path.exists(_SO) or any(
        os.path.getmtime(s) > os.path.getmtime(_SO) for s in src if os.path.exists(s)
    ):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


class Blob(C.Structure):
    _fields_ = [
        ("x", C.c_int), ("y", C.c_int), ("w", C.c_int), ("h", C.c_int),
        ("area", C.c_double), ("radius", C.c_double),
        ("m00", C.c_double), ("m10", C.c_double), ("m01", C.c_double),
        ("cx", C.c_float), ("cy", C.c_float),
    ]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


_u8p = C.POINTER(C.c_uint8)
_u32p = C.POINTER(C.c_uint32)
_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(_SO)
    L.orc_welford.argtypes = [_u8p, C.c_int, C.c_int, C.c_int, _u8p, _u8p]
    L.orc_pair_entropy16.argtypes = [_u8p, _u8p, C.c_int, C.c_int]
    L.orc_pair_entropy16.restype = C.c_float
    L.orc_entropy16.argtypes = [_u8p, C.c_int, C.c_int]
    L.orc_entropy16.restype = C.c_float
    L.orc_entropy128.argtypes = [_u8p, C.c_int, C.c_int]
    L.orc_entropy128.restype = C.c_float
    L.orc_process_frame.argtypes = [_u8p, _u8p, _u8p, C.c_int, C.c_int, _u8p]
    L.orc_process_frame_roi.argtypes = [_u8p, _u8p, _u8p] + [C.c_int] * 6 + [_u8p]
    L.orc_hist256.argtypes = [_u8p, C.c_size_t, _u32p]
    L.orc_posttrig_frame.argtypes = [_u8p, _u8p, _u8p, C.c_int, C.c_int, _u8p]
    L.orc_otsu.argtypes = [_u32p, C.c_size_t]
    L.orc_otsu.restype = C.c_int
    L.orc_binarize.argtypes = [_u8p, C.c_int, C.c_int, C.c_int, _u8p]
    L.orc_binarize.restype = C.c_int
    L.orc_find_contours.argtypes = [_u8p, C.c_int, C.c_int]
    L.orc_find_contours.restype = C.c_void_p
    L.orc_contours_free.argtypes = [C.c_void_p]
    L.orc_contours_count.argtypes = [C.c_void_p]
    L.orc_contour_npts.argtypes = [C.c_void_p, C.c_int]
    L.orc_contour_nchain.argtypes = [C.c_void_p, C.c_int]
    L.orc_contour_points.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int)]
    L.orc_blob_from_contour.argtypes = [C.POINTER(C.c_int), C.c_int, C.c_int, C.POINTER(Blob)]
    L.orc_analyzer_create.argtypes = [_u8p, C.c_int, C.c_int, C.c_int, _u8p, _u8p, C.c_int, _u8p,
                                      _u8p, C.c_int, C.c_int, _u8p, C.c_int, C.c_int]
    L.orc_analyzer_create.restype = C.c_void_p
    L.orc_analyzer_destroy.argtypes = [C.c_void_p]
    L.orc_significance.argtypes = [C.c_void_p, _u32p, C.c_int]
    L.orc_significance.restype = C.c_double
    L.orc_find_trigger.argtypes = [C.c_void_p, C.c_int]
    L.orc_localize.argtypes = [C.c_void_p]
    L.orc_any_cam_analysis.argtypes = [C.c_void_p]
    L.orc_any_cam_analysis.restype = C.c_int
    for f in ("trig_frame", "status", "ok", "loc_thres", "nbubbles"):
        getattr(L, "orc_get_" + f).argtypes = [C.c_void_p]
        getattr(L, "orc_get_" + f).restype = C.c_int
    L.orc_get_bubble_ndesc.argtypes = [C.c_void_p, C.c_int]
    L.orc_get_bubble_desc.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(Blob)]
    L.orc_get_bubble_ndz.argtypes = [C.c_void_p, C.c_int]
    L.orc_get_bubble_dz.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.orc_get_bubble_dz.restype = C.c_float
    L.orc_get_bubble_dzdt.argtypes = [C.c_void_p, C.c_int]
    L.orc_get_bubble_dzdt.restype = C.c_float
    L.orc_get_bubble_drdt.argtypes = [C.c_void_p, C.c_int]
    L.orc_get_bubble_drdt.restype = C.c_float
    L.orc_get_sig_trace.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_int]
    L.orc_get_pixcount_len.argtypes = [C.c_void_p, C.c_int]
    L.orc_analyzer_set_bellows_template.argtypes = [C.c_void_p, _u8p, C.c_int, C.c_int]
    L.orc_match_template_ccorr_normed.argtypes = [_u8p, C.c_int, C.c_int, _u8p, C.c_int, C.c_int, C.POINTER(C.c_float)]
    L.orc_track_feature.argtypes = [_u8p, C.c_int, C.c_int, _u8p, C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.orc_format_header.argtypes = [C.c_char_p, C.c_int]
    L.orc_format_event.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.c_int, C.c_char_p, C.c_int, C.c_int,
                                   C.c_char_p, C.c_int]
    L.orc_bench_trigger_pass.argtypes = [_u8p, C.c_int, C.c_int, C.c_int, _u8p, C.c_int, C.c_int,
                                         C.c_int, _u32p]
    L.orc_bench_trigger_pass.restype = C.c_uint64
    _lib = L
    return L


def _p(a):
    return a.ctypes.data_as(_u8p)


def _img(a):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    return a


def welford(slab):
    slab = _img(slab)
    N, H, W = slab.shape
    mu = np.empty((H, W), np.uint8)
    sg = np.empty((H, W), np.uint8)
    lib().orc_welford(_p(slab), N, W, H, _p(mu), _p(sg))
    return mu, sg


def pair_entropy16(f1, f0):
    f1, f0 = _img(f1), _img(f0)
    H, W = f1.shape
    return float(lib().orc_pair_entropy16(_p(f1), _p(f0), W, H))


def entropy16(img):
    img = _img(img)
    return float(lib().orc_entropy16(_p(img), img.shape[1], img.shape[0]))


def entropy128(img):
    img = _img(img)
    return float(lib().orc_entropy128(_p(img), img.shape[1], img.shape[0]))


def process_frame(cur, ref, sigma, roi=None):
    cur, ref, sigma = _img(cur), _img(ref), _img(sigma)
    H, W = cur.shape
    D = np.empty((H, W), np.uint8)
    if roi is None:
        lib().orc_process_frame(_p(cur), _p(ref), _p(sigma), W, H, _p(D))
    else:
        rx, ry, rw, rh = roi
        lib().orc_process_frame_roi(_p(cur), _p(ref), _p(sigma), W, H, rx, ry, rw, rh, _p(D))
    return D


def hist256(img):
    img = _img(img)
    h = np.zeros(256, np.uint32)
    lib().orc_hist256(_p(img), img.size, h.ctypes.data_as(_u32p))
    return h


def posttrig_frame(frame, mu, sigma):
    frame, mu, sigma = _img(frame), _img(mu), _img(sigma)
    H, W = frame.shape
    O = np.empty((H, W), np.uint8)
    lib().orc_posttrig_frame(_p(frame), _p(mu), _p(sigma), W, H, _p(O))
    return O


def otsu(hist, P=None):
    hist = np.ascontiguousarray(hist, dtype=np.uint32)
    if P is None:
        P = int(hist.sum())
    return int(lib().orc_otsu(hist.ctypes.data_as(_u32p), P))


def binarize(img, thr):
    img = _img(img)
    H, W = img.shape
    m = np.empty((H, W), np.uint8)
    T = lib().orc_binarize(_p(img), W, H, int(thr), _p(m))
    return m, int(T)


def find_contours(mask):
    """-> list of (points ndarray [n,2] (x,y), nchain) in cv::findContours output order."""
    mask = _img(mask)
    H, W = mask.shape
    L = lib()
    h = L.orc_find_contours(_p(mask), W, H)
    out = []
    try:
        for i in range(L.orc_contours_count(h)):
            n = L.orc_contour_npts(h, i)
            xy = np.zeros((n, 2), np.int32)
            L.orc_contour_points(h, i, xy.ctypes.data_as(C.POINTER(C.c_int)))
            out.append((xy, L.orc_contour_nchain(h, i)))
    finally:
        L.orc_contours_free(h)
    return out


def blob_from_contour(xy, genesis_fallback=True):
    xy = np.ascontiguousarray(xy, dtype=np.int32)
    b = Blob()
    lib().orc_blob_from_contour(xy.ctypes.data_as(C.POINTER(C.c_int)), len(xy),
                                1 if genesis_fallback else 0, C.byref(b))
    return b.as_dict()


class Analyzer:
    """One in-memory (event, camera) driven through the oracle's AnalyzerUnit/L3Localizer restatement."""

    def __init__(self, frames, mu, sigma, training_set_size, frame_ok=None, fid_mask=None,
                 bel_mask=None, bel_template=None):
        self._keep = []
        self.frames = _img(frames)
        F, H, W = self.frames.shape
        self.mu, self.sigma = _img(mu), _img(sigma)

        def opt(m):
            if m is None:
                return None, 0, 0
            m = _img(m)
            self._keep.append(m)
            return _p(m), m.shape[1], m.shape[0]

        fo = None
        if frame_ok is not None:
            fo_arr = _img(frame_ok)
            self._keep.append(fo_arr)
            fo = _p(fo_arr)
        fm, fw, fh = opt(fid_mask)
        bm, bw, bh = opt(bel_mask)
        self.F = F
        self._h = lib().orc_analyzer_create(_p(self.frames), F, W, H, _p(self.mu), _p(self.sigma),
                                            int(training_set_size), fo, fm, fw, fh, bm, bw, bh)
        if bel_template is not None:
            t = _img(bel_template)
            self._keep.append(t)
            lib().orc_analyzer_set_bellows_template(self._h, _p(t), t.shape[1], t.shape[0])

    def close(self):
        if self._h:
            lib().orc_analyzer_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def significance(self, hist, store):
        hist = np.ascontiguousarray(hist, dtype=np.uint32)
        return float(lib().orc_significance(self._h, hist.ctypes.data_as(_u32p), 1 if store else 0))

    def find_trigger(self, startframe=1):
        lib().orc_find_trigger(self._h, int(startframe))
        return self.state()

    def localize(self):
        lib().orc_localize(self._h)
        return self.bubbles()

    def any_cam_analysis(self):
        staged = lib().orc_any_cam_analysis(self._h)
        return int(staged), self.state(), self.bubbles()

    def state(self):
        L = lib()
        return {
            "trig": L.orc_get_trig_frame(self._h),
            "status": L.orc_get_status(self._h),
            "ok": bool(L.orc_get_ok(self._h)),
            "loc_thres": L.orc_get_loc_thres(self._h),
        }

    def sig_trace(self):
        s = np.zeros(self.F, np.float64)
        lib().orc_get_sig_trace(self._h, s.ctypes.data_as(C.POINTER(C.c_double)), self.F)
        return s

    def pixcount_len(self, b):
        return lib().orc_get_pixcount_len(self._h, b)

    def bubbles(self):
        L = lib()
        out = []
        for b in range(L.orc_get_nbubbles(self._h)):
            descs = []
            for d in range(L.orc_get_bubble_ndesc(self._h, b)):
                bl = Blob()
                L.orc_get_bubble_desc(self._h, b, d, C.byref(bl))
                descs.append(bl.as_dict())
            dz = [L.orc_get_bubble_dz(self._h, b, i) for i in range(L.orc_get_bubble_ndz(self._h, b))]
            out.append({"desc": descs, "dz": dz,
                        "dzdt": L.orc_get_bubble_dzdt(self._h, b),
                        "drdt": L.orc_get_bubble_drdt(self._h, b)})
        return out


def bench_trigger_pass(frames, sigma, ref_offset, first, count, want_hists=False):
    frames, sigma = _img(frames), _img(sigma)
    F, H, W = frames.shape
    hists = np.zeros((count, 256), np.uint32) if want_hists else None
    ck = lib().orc_bench_trigger_pass(_p(frames), F, W, H, _p(sigma), ref_offset, first, count,
                                      hists.ctypes.data_as(_u32p) if want_hists else None)
    return int(ck), hists


def format_header():
    buf = C.create_string_buffer(1 << 14)
    n = lib().orc_format_header(buf, len(buf))
    assert n >= 0
    return buf.raw[:n].decode()


def format_event(analyzers, staged, run_number, event, frame_offset):
    """analyzers: list of Analyzer (one per camera, already through any_cam_analysis)."""
    n = len(analyzers)
    hs = (C.c_void_p * n)(*[a._h for a in analyzers])
    st = (C.c_int * n)(*staged)
    buf = C.create_string_buffer(1 << 20)
    w = lib().orc_format_event(hs, st, n, run_number.encode(), int(event), int(frame_offset), buf, len(buf))
    assert w >= 0
    return buf.raw[:w].decode()


def match_template(img, tmpl):
    img, tmpl = _img(img), _img(tmpl)
    H, W = img.shape
    th, tw = tmpl.shape
    res = np.zeros((H - th + 1, W - tw + 1), np.float32)
    lib().orc_match_template_ccorr_normed(_p(img), W, H, _p(tmpl), tw, th, res.ctypes.data_as(C.POINTER(C.c_float)))
    return res


def track_feature(img, tmpl):
    img, tmpl = _img(img), _img(tmpl)
    H, W = img.shape
    th, tw = tmpl.shape
    bx, by = C.c_float(), C.c_float()
    lib().orc_track_feature(_p(img), W, H, _p(tmpl), tw, th, C.byref(bx), C.byref(by))
    return bx.value, by.value
