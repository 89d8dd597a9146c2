"""TEST INFRASTRUCTURE -- ctypes front end of the CPU restatement (statsplit_oracle.c).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package.  The product package (pypore_amd) never does.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libstatsplit_oracle.so")
_lib = None


def build(force=False):
    src = os.path.join(_HERE, "statsplit_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "libstatsplit_oracle.so"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_SO)
        dp = ctypes.POINTER(ctypes.c_double)
        ip = ctypes.POINTER(ctypes.c_int)
        lp = ctypes.POINTER(ctypes.c_long)
        L.so_min_gain.argtypes = [ctypes.c_int] * 3 + [ctypes.c_double] * 5 + [dp]
        L.so_min_gain.restype = ctypes.c_int
        L.so_parse.argtypes = [dp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                               ctypes.c_double, ip, ctypes.c_long, ctypes.POINTER(ctypes.c_longlong)]
        L.so_parse.restype = ctypes.c_long
        L.so_parse_flags.argtypes = [dp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double, ip,
                                     ctypes.POINTER(ctypes.c_ubyte), ctypes.c_long]
        L.so_parse_flags.restype = ctypes.c_long
        L.so_best_single_split.argtypes = [dp, ctypes.c_int, dp, ip]
        L.so_best_single_split.restype = ctypes.c_int
        L.so_score_window.argtypes = [dp, ctypes.c_int, ctypes.c_int, ctypes.c_double, dp]
        L.so_score_window.restype = ctypes.c_int
        L.so_segment_stats.argtypes = [dp, ip, ctypes.c_long, ctypes.c_int, dp]
        L.so_segment_stats.restype = None
        L.so_lambda_events.argtypes = [dp, ctypes.c_long, ctypes.c_double, ctypes.c_long,
                                       ctypes.c_double, lp, lp, ctypes.c_long]
        L.so_lambda_events.restype = ctypes.c_long
        L.so_bessel1_filtfilt.argtypes = [dp, ctypes.c_long, ctypes.c_double, ctypes.c_double, dp]
        L.so_bessel1_filtfilt.restype = ctypes.c_int
        L.so_bessel_filtfilt.argtypes = [dp, ctypes.c_long, ctypes.c_int, ctypes.c_double, ctypes.c_double, dp]
        L.so_bessel_filtfilt.restype = ctypes.c_int
        L.so_bessel_ba.argtypes = [ctypes.c_int, ctypes.c_double, dp, dp]
        L.so_bessel_ba.restype = ctypes.c_int
        L.so_align.argtypes = [dp, dp, dp, ctypes.c_int, ctypes.c_double, ctypes.c_double, dp, dp, dp, ctypes.c_int,
                               dp, ctypes.POINTER(ctypes.c_uint)]
        L.so_align.restype = ctypes.c_int
        _lib = L
    return _lib


def _dptr(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def min_gain(min_width=100, max_width=1000000, window_width=10000, min_gain_per_sample=None,
             false_positive_rate=None, prior_segments_per_second=None, sampling_freq=1.e5,
             cutoff_freq=None):
    out = ctypes.c_double()
    rc = lib().so_min_gain(int(min_width), int(max_width), int(window_width),
                           float(min_gain_per_sample or 0.0), float(false_positive_rate or 0.0),
                           float(prior_segments_per_second or 0.0), float(sampling_freq),
                           float(cutoff_freq or 0.0), ctypes.byref(out))
    if rc:
        raise AssertionError("reference assertion %d (cparsers.pyx:69-76)" % -rc)
    return out.value


def parse(x, min_width=100, max_width=1000000, window_width=10000, min_gain=None, stats=False, **kw):
    """Breakpoints (int32 array, excluding 0 and n) of FastStatSplit.parse(x)."""
    x = np.ascontiguousarray(x, dtype=np.float64)
    if min_gain is None:
        min_gain = globals()["min_gain"](min_width, max_width, window_width, **kw)
    cap = max(16, x.size // max(1, int(min_width)) + 16)
    out = np.empty(cap, dtype=np.int32)
    st = (ctypes.c_longlong * 2)()
    n = lib().so_parse(_dptr(x), x.size, int(min_width), int(max_width), int(window_width),
                       float(min_gain), out.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), cap, st)
    if n < 0 or n > cap:
        raise RuntimeError("oracle so_parse failed (%d)" % n)
    b = out[:n].copy()
    return (b, (int(st[0]), int(st[1]))) if stats else b


def parse_flags(x, min_width=100, max_width=1000000, window_width=10000, min_gain=None, **kw):
    """(breakpoints, is_spine flags) -- spine anchors are the breakpoints of the top-level chain."""
    x = np.ascontiguousarray(x, dtype=np.float64)
    if min_gain is None:
        min_gain = globals()["min_gain"](min_width, max_width, window_width, **kw)
    cap = max(16, x.size // max(1, int(min_width)) + 16)
    out = np.empty(cap, dtype=np.int32)
    fl = np.zeros(cap, dtype=np.uint8)
    n = lib().so_parse_flags(_dptr(x), x.size, int(min_width), int(max_width), int(window_width), float(min_gain),
                             out.ctypes.data_as(ctypes.POINTER(ctypes.c_int)),
                             fl.ctypes.data_as(ctypes.POINTER(ctypes.c_ubyte)), cap)
    if n < 0 or n > cap:
        raise RuntimeError("oracle so_parse_flags failed (%d)" % n)
    return out[:n].copy(), fl[:n].copy()


def best_single_split(x):
    x = np.ascontiguousarray(x, dtype=np.float64)
    g = ctypes.c_double()
    i = ctypes.c_int()
    lib().so_best_single_split(_dptr(x), x.size, ctypes.byref(g), ctypes.byref(i))
    return g.value, i.value


def score_window(x, min_width=100, min_gain=0.0):
    x = np.ascontiguousarray(x, dtype=np.float64)
    s = np.empty(x.size, dtype=np.float64)
    r = lib().so_score_window(_dptr(x), x.size, int(min_width), float(min_gain), _dptr(s))
    return r, s


def segment_stats(x, bounds):
    x = np.ascontiguousarray(x, dtype=np.float64)
    b = np.ascontiguousarray(bounds, dtype=np.int32)
    out = np.empty((b.size + 1, 4), dtype=np.float64)
    lib().so_segment_stats(_dptr(x), b.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), b.size, x.size, _dptr(out))
    return out


def lambda_events(x, threshold=90.0, min_duration=100000, min_current=-0.5):
    x = np.ascontiguousarray(x, dtype=np.float64)
    cap = 1024
    while True:
        st = np.empty(cap, dtype=np.int64)
        ln = np.empty(cap, dtype=np.int64)
        n = lib().so_lambda_events(_dptr(x), x.size, float(threshold), int(min_duration), float(min_current),
                                   st.ctypes.data_as(ctypes.POINTER(ctypes.c_long)),
                                   ln.ctypes.data_as(ctypes.POINTER(ctypes.c_long)), cap)
        if n <= cap:
            return st[:n].copy(), ln[:n].copy()
        cap = n


def bessel_filtfilt(x, cutoff=2000.0, second=1.0e5, order=1):
    """Event.filter (DataTypes.py:258-274): Bessel low-pass of `order`, filtfilt semantics; float64 in, float64 out.
    order 1 takes the closed form (so_bessel1_filtfilt), any order 1..8 the general restatement (so_bessel_filtfilt)."""
    x = np.ascontiguousarray(x, dtype=np.float64)
    out = np.empty_like(x)
    if order == 1:
        rc = lib().so_bessel1_filtfilt(_dptr(x), x.size, float(cutoff), float(second), _dptr(out))
    else:
        rc = lib().so_bessel_filtfilt(_dptr(x), x.size, int(order), float(cutoff), float(second), _dptr(out))
    if rc != 0:
        pad = 3 * (int(order) + 1)
        raise ValueError("The length of the input vector x must be greater than padlen, which is %d." % pad if x.size <= pad
                         else "cutoff must lie strictly between 0 and the Nyquist frequency (orders 1..8)")
    return out


def bessel_ba(order, wn):
    """(b, a) of scipy.signal.bessel(order, wn, 'low', analog=False, output='ba') as the oracle restates it."""
    b = np.zeros(order + 1); a = np.zeros(order + 1)
    if lib().so_bessel_ba(int(order), float(wn), _dptr(b), _dptr(a)):
        raise ValueError("order must be 1..8 and 0 < wn < 1")
    return b, a


ALIGN_ERRORS = {1: ValueError, 2: IndexError, 3: ZeroDivisionError, 4: IndexError}


def align_raw(model_means, model_stds, model_durs, skip_penalty, backslip_penalty, seq_means, seq_stds, seq_durs):
    """cSegmentAligner(model..., penalties).align(seq...) (calignment.pyx:20-100) -> (rc, score[s-1, m-1], path uint32)."""
    a = [np.ascontiguousarray(v, dtype=np.float64) for v in (model_means, model_stds, model_durs,
                                                              seq_means, seq_stds, seq_durs)]
    m, s = a[0].size, a[3].size
    path = np.zeros(max(s, 1), dtype=np.uint32)
    score = ctypes.c_double(0.0)
    rc = lib().so_align(_dptr(a[0]), _dptr(a[1]), _dptr(a[2]), m, float(skip_penalty), float(backslip_penalty),
                        _dptr(a[3]), _dptr(a[4]), _dptr(a[5]), s, ctypes.byref(score),
                        path.ctypes.data_as(ctypes.POINTER(ctypes.c_uint)))
    return rc, score.value, path[:s]


def align(model_means, model_stds, model_durs, skip_penalty, backslip_penalty, seq_means, seq_stds, seq_durs):
    """What the reference returns: (score[s-1, m-1] / np.sum(seq_durs), float64 array of model indices); raises the
    exception class the compiled reference raises (rc 4 = undefined there, IndexError here)."""
    rc, score, path = align_raw(model_means, model_stds, model_durs, skip_penalty, backslip_penalty,
                                seq_means, seq_stds, seq_durs)
    if rc:
        raise ALIGN_ERRORS[rc]("cSegmentAligner.align: reference raises here (code %d)" % rc)
    return score / np.sum(np.asarray(seq_durs, dtype=np.float64)), path.astype(np.float64)
