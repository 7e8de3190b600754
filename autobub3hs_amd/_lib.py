"""ctypes loader of the in-tree gfx950 library (autobub3hs_amd/libabub_hip.so).

The product path has NO CPU fallback: if the HIP library is missing or cannot be loaded this module
raises, loudly.  (No CPU fallback exists in this package.)
"""
import ctypes as C
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "libabub_hip.so")
HEADER = os.path.join(os.path.dirname(HERE), "include", "abub_hip.h")


class Job(C.Structure):
    _fields_ = [("cur", C.c_uint32), ("ref", C.c_uint32), ("model", C.c_uint32), ("out", C.c_uint32)]


def build(force=False):
    """Compile the HIP library for gfx950 (cross-compiles without a GPU)."""
    srcs = [os.path.join(HERE, "csrc", f) for f in os.listdir(os.path.join(HERE, "csrc"))
            if f.endswith((".hip", ".h", ".hpp")) or f == "Makefile"] + [HEADER]
    stale = force or not os.path.exists(SO) or any(os.path.getmtime(s) > os.path.getmtime(SO) for s in srcs)
    if stale:
        subprocess.check_call(["make", "-C", os.path.join(HERE, "csrc"), "-s"])
    return SO


_vp, _i, _sz = C.c_void_p, C.c_int, C.c_size_t
SIGNATURES = {
    "abub_last_error": (C.c_char_p, []),
    "abub_device_count": (_i, []),
    "abub_device_info": (_i, [_i, C.c_char_p, _i, C.POINTER(_i), C.POINTER(C.c_uint64)]),
    "abub_sigma6_dev": (_i, [_vp, _vp, _sz, _vp]),
    "abub_fill_stack_jobs_dev": (_i, [_vp, _i, _i, _i, _i, _i, _i, _vp]),
    "abub_diff_hist_dev": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp, _i, _vp]),
    "abub_diff_roi_dev": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "abub_train_dev": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp]),
    "abub_pair_hist_dev": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp]),
    "abub_posttrig_dev": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp]),
    "abub_fg_compact_dev": (_i, [_vp, _i, _i, _i, _vp, _vp, _i, _vp, _vp]),
    "abub_fast_path": (_i, [_i]),
    "abub_scratch_release": (_i, [_vp]),
    "abub_bound_counts_dev": (_i, [_vp, C.POINTER(C.c_uint32)]),
    "abub_diff_hist_chained_dev": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _i, _i, _vp]),
    "abub_diff_hist_chained_store_dev": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp, _i, _i, _vp]),
    "abub_k2_set_option": (_i, [C.c_char_p, _i]),
    "abub_k2_pieces_cap": (_sz, [_i, _i, _i]),
    "abub_png_raw_stride": (_sz, [_i, _i]),
    "abub_png_decode_dev": (_i, [_vp, _sz, _vp, _i, _vp, _i, _vp, _i, _i, _i, _vp, _sz, _vp, _sz, _vp, _sz, _vp, _vp]),
    "abub_diff_hist_chained_deferred_dev": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _i, _i, _vp, C.c_uint32, _vp, _vp, _vp]),
    "abub_diff_hist_pieces_dev": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "abub_diff_hist_compact_dev": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, C.c_uint32, _vp, C.c_uint32, _vp]),
    "abub_posttrig_compact_dev": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, C.c_uint32, _vp, C.c_uint32, _vp]),
    "abub_pairs_group_dev": (_i, [_vp, _vp, C.c_uint32, _i, _vp, _vp, _vp, _vp, _vp]),
    "abub_pairs_group_hist_dev": (_i, [_vp, _vp, C.c_uint32, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "abub_fg_compact_pairs_dev": (_i, [_vp, _i, _i, _i, _vp, _vp, C.c_uint32, _vp, _vp]),
    "abub_match_ccorr_dev": (_i, [_vp, _i, _i, _vp, _i, _i, _vp, _vp, _vp]),
    "abub_subsat_hist_dev": (_i, [_vp, _vp, _i, _i, _vp, _vp]),
    "abub_ctx_create": (_i, [C.POINTER(_vp), _i, _i, _i, _i]),
    "abub_ctx_destroy": (None, [_vp]),
    "abub_ctx_train": (_i, [_vp, C.POINTER(_vp), _i, _vp, _vp]),
    "abub_ctx_pair_hist": (_i, [_vp, _vp, _vp, _vp]),
    "abub_ctx_set_model": (_i, [_vp, _vp, _vp]),
    "abub_ctx_upload_stack": (_i, [_vp, C.POINTER(_vp), _i]),
    "abub_ctx_diff_hist_batch": (_i, [_vp, _i, _i, _i, _vp]),
    "abub_ctx_diff_frame": (_i, [_vp, _i, _i, _vp, _vp]),
    "abub_ctx_diff_frame_roi": (_i, [_vp, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "abub_ctx_posttrig": (_i, [_vp, _i, _vp, _vp]),
    "abub_ctx_foreground": (_i, [_vp, _i, _vp, _i, C.POINTER(_i)]),
    "abub_ctx_match_template": (_i, [_vp, _i, _vp, _i, _i, _vp, _vp]),
    "abub_ctx_subtract_image": (_i, [_vp, _vp, _vp]),
    "abub_ctx_set_image": (_i, [_vp, _vp]),
    "abub_ctx_fetch_image": (_i, [_vp, _vp]),
}

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(SO):
            raise RuntimeError(
                f"{SO} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(there is no CPU fallback for the hot path)")
        L = C.CDLL(SO)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError if the ABI and the header drift apart
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


class AbubError(RuntimeError):
    pass


def check(rc, what=""):
    if rc != 0:
        raise AbubError(f"{what} failed rc={rc}: {lib().abub_last_error().decode(errors='replace')}")
