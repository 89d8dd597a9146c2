"""ctypes binding of libporeseg.so (include/poreseg.h) -- the thin shim between the Python
class surface and the HIP kernels.  There is NO CPU fallback: if the shared library is missing
or a compute entry point is called without a GPU, this module raises.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PORESEG_LIB") or os.path.join(_HERE, "libporeseg.so")   # PORESEG_LIB: diagnostic builds

PS_OK = 0
PS_ERR_ARG, PS_ERR_ASSERT_WIDTH, PS_ERR_ASSERT_WINDOW, PS_ERR_ASSERT_CUTOFF = -1, -2, -3, -4
PS_ERR_CAPACITY, PS_ERR_OFF_GRID, PS_ERR_HIP, PS_ERR_NO_DEVICE, PS_ERR_INTERNAL = -5, -6, -7, -8, -9
PS_DTYPE_F32, PS_DTYPE_I16 = 0, 1
PS_DTYPE_F64 = 2          # float64 pA on no grid: ps_filter_bessel input only
PS_ALIGN_OK, PS_ALIGN_VALUE_ERROR, PS_ALIGN_INDEX_ERROR, PS_ALIGN_ZERO_DIVISION, PS_ALIGN_UNDEFINED = 0, 1, 2, 3, 4

EXPORTS = ["ps_version", "ps_device_count", "ps_create", "ps_destroy", "ps_last_error", "ps_set_tiling", "ps_set_option",
           "ps_synchronize", "ps_min_gain", "ps_segment_batch", "ps_segment_batch_ex", "ps_segment_events", "ps_segment_exact_f64", "ps_detect_events", "ps_detect_segment_trace", "ps_bounds_capacity",
           "ps_best_single_split", "ps_score_window", "ps_get_timings", "ps_synth_trace", "ps_filter_bessel",
           "ps_requantise", "ps_filter_requantise_batch", "ps_align_batch", "ps_audit_bounds", "ps_counters"]


class SplitParams(ctypes.Structure):
    _fields_ = [("min_width", ctypes.c_int32), ("max_width", ctypes.c_int32), ("window_width", ctypes.c_int32),
                ("min_gain_per_sample", ctypes.c_double), ("false_positive_rate", ctypes.c_double),
                ("prior_segments_per_second", ctypes.c_double), ("sampling_freq", ctypes.c_double),
                ("cutoff_freq", ctypes.c_double)]


class SampleFormat(ctypes.Structure):
    _fields_ = [("dtype", ctypes.c_int32), ("offset_counts", ctypes.c_int32), ("quantum", ctypes.c_double)]


_lib = None


def lib():
    """Loads libporeseg.so; raises RuntimeError (never falls back) when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "pypore_amd: %s not found -- build it with `make -C pypore_amd/csrc` (hipcc, gfx950). "
            "There is no CPU fallback." % LIB_PATH)
    # torch bundles its own libamdhip64.so.7 (same soname as /opt/rocm's): import it FIRST so the
    # process has exactly one HIP runtime, shared by torch's allocator and our kernels.
    import torch  # noqa: F401
    L = ctypes.CDLL(LIB_PATH)
    vp, i32, i64, dbl = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_double
    P = ctypes.POINTER
    L.ps_version.restype = ctypes.c_char_p
    L.ps_device_count.restype = ctypes.c_int
    L.ps_create.argtypes = [ctypes.c_int, vp, P(vp)]
    L.ps_destroy.argtypes = [vp]
    L.ps_destroy.restype = None
    L.ps_last_error.argtypes = [vp]
    L.ps_last_error.restype = ctypes.c_char_p
    L.ps_set_tiling.argtypes = [vp, i64, i64]
    L.ps_set_option.argtypes = [vp, ctypes.c_char_p, i64]
    L.ps_synchronize.argtypes = [vp]
    L.ps_min_gain.argtypes = [P(SplitParams), P(dbl)]
    L.ps_segment_batch.argtypes = [vp, vp, P(SampleFormat), P(i64), i32, P(SplitParams), vp, i64, P(i64), vp]
    L.ps_segment_batch_ex.argtypes = [vp, vp, P(SampleFormat), P(i64), i32, P(SplitParams), vp, i64, P(i64), vp, vp]
    L.ps_segment_events.argtypes = [vp, vp, P(SampleFormat), P(i64), P(i64), i32, P(SplitParams), vp, i64, P(i64), vp, vp]
    L.ps_segment_exact_f64.argtypes = [vp, vp, P(i64), P(i64), i32, P(SplitParams), vp, i64, P(i64)]
    L.ps_detect_events.argtypes = [vp, vp, P(SampleFormat), i64, dbl, i64, dbl, P(i64), P(i64), i64, P(i64)]
    L.ps_detect_segment_trace.argtypes = [vp, vp, P(SampleFormat), i64, dbl, i64, dbl, P(SplitParams), P(i64), P(i64), i64, P(i64),
                                          vp, i64, P(i64), vp]
    L.ps_bounds_capacity.argtypes = [P(i64), i32, i32]
    L.ps_bounds_capacity.restype = i64
    L.ps_best_single_split.argtypes = [vp, vp, P(SampleFormat), i64, P(dbl), P(i32)]
    L.ps_score_window.argtypes = [vp, vp, P(SampleFormat), i64, i32, dbl, vp, P(i32)]
    L.ps_get_timings.argtypes = [vp, P(dbl), i32, P(i64), i32]
    L.ps_synth_trace.argtypes = [vp, vp, i32, i64, ctypes.c_uint64, P(i64), P(i32), i64]
    L.ps_filter_bessel.argtypes = [vp, vp, P(SampleFormat), i64, i32, dbl, dbl, vp]
    L.ps_requantise.argtypes = [vp, vp, i64, vp, P(dbl), P(dbl)]
    L.ps_filter_requantise_batch.argtypes = [vp, vp, P(SampleFormat), P(i64), P(i64), ctypes.c_int32, ctypes.c_int32, dbl, dbl,
                                             vp, vp, P(dbl), P(dbl)]
    L.ps_align_batch.argtypes = [vp, P(dbl), P(dbl), P(dbl), i32, dbl, dbl, vp, vp, vp, P(i64), i32, vp, vp, vp]
    L.ps_counters.argtypes = [vp]
    L.ps_counters.restype = P(i64)
    L.ps_audit_bounds.argtypes = [vp, vp, P(SampleFormat), i64, P(SplitParams), P(i32), i32, P(dbl)]
    _lib = L
    return L


def split_params(min_width=100, max_width=1000000, window_width=10000, min_gain_per_sample=None,
                 false_positive_rate=None, prior_segments_per_second=None, sampling_freq=1.e5,
                 cutoff_freq=None):
    """The eight FastStatSplit constructor arguments (cparsers.pyx:55-57) as a C struct;
    None and 0 both mean "not given", as in the reference (`if not x`)."""
    return SplitParams(int(min_width), int(max_width), int(window_width),
                       float(min_gain_per_sample or 0.0), float(false_positive_rate or 0.0),
                       float(prior_segments_per_second or 0.0), float(sampling_freq),
                       float(cutoff_freq or 0.0))


_ASSERT_MSG = {
    PS_ERR_ASSERT_WIDTH: "Maximum width must be greater than minimum width.",
    PS_ERR_ASSERT_WINDOW: "Window width must be greater than twice the minimum width.",
    PS_ERR_ASSERT_CUTOFF: "Cutoff freq must be less than half the sampling frequency.",
}


def check(rc, ctx=None):
    """Maps a status code to the exception the reference would raise (AssertionError for the
    three constructor assertions, ValueError for bad buffers) or RuntimeError."""
    if rc == PS_OK:
        return
    msg = ""
    if ctx is not None:
        m = lib().ps_last_error(ctx)
        msg = m.decode() if m else ""
    if rc in _ASSERT_MSG:
        raise AssertionError(_ASSERT_MSG[rc])
    if rc in (PS_ERR_ARG, PS_ERR_OFF_GRID):
        raise ValueError("poreseg: %s (status %d)" % (msg or "invalid argument", rc))
    if rc == PS_ERR_NO_DEVICE:
        raise RuntimeError("poreseg: no MI355X (gfx950) device available -- there is no CPU fallback")
    raise RuntimeError("poreseg: %s (status %d)" % (msg or "error", rc))


def min_gain(**kw):
    out = ctypes.c_double()
    p = split_params(**kw)
    check(lib().ps_min_gain(ctypes.byref(p), ctypes.byref(out)))
    return out.value
