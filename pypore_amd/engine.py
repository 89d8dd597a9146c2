"""Device-side plumbing between Python and libporeseg.so: one Context per GPU (a ps_ctx handle
plus torch-allocated HBM buffers).  torch is used for device memory and streams only; all
compute is in the HIP kernels behind the C ABI.
"""
import collections
import ctypes
import os
import threading
import warnings

import numpy as np
import torch

from . import _lib

_contexts = {}

# Options every new Context of this process gets (ps_set_option name -> value) and its tiling (ps_set_tiling).  EMPTY in
# the product: neither this package nor libporeseg.so reads a PORESEG_* variable on its own (round 6; the one exception is
# PORESEG_LIB / PORESEG_COMM_LIB, the path of the library to load).  tests/conftest.py and the scripts under tools/ fill
# them from the environment through apply_env_defaults(), so that `PORESEG_MODE=2 pytest -m gpu` still runs the whole
# suite in verify mode on the product library.
DEFAULT_OPTIONS = {}
DEFAULT_TILING = [0, 0]
# options a StreamPool applies to its contexts on top of "shared_device" while it runs (experiments: tools/)
POOL_OVERRIDES = {}

EVENT_CAP = 1 << 16      # events the detector's host arrays hold at first (Context.detect_events / detect_segment_trace grow them when a trace has more)

_ENV_OPTIONS = {
    "PORESEG_MODE": "mode", "PORESEG_SPINE_NT": "spine_nt", "PORESEG_TREE_NT": "tree_nt", "PORESEG_PRUNE": "prune",
    "PORESEG_SCAN_BS": "scan_bs", "PORESEG_GROUPS": "groups", "PORESEG_TREE_PAR": "tree_par", "PORESEG_K0_WAVES": "k0_waves",
    "PORESEG_K0_SHARED": "k0_shared", "PORESEG_K0_MAX": "k0_admit", "PORESEG_K0_SETS": "k0_sets", "PORESEG_WIDE_BS": "wide_bs",
    "PORESEG_BRIDGE_SINGLE": "bridge_single", "PORESEG_TREE_TAIL": "tree_tail_pct", "PORESEG_FILTER_FUSED": "filter_fused",
    "PORESEG_UPLOAD": "upload_by_kernel", "PORESEG_TIMING": "timing", "PORESEG_TREE_MW": "tree_mw",
    "PORESEG_TREE_JPW": "tree_jobs_per_wave", "PORESEG_SLOTS_PCT": "slots_pct", "PORESEG_BRIDGE_EXT": "bridge_ext",
    "PORESEG_LAT_HELP": "lat_help", "PORESEG_BRIDGE_BUDGET": "bridge_budget", "PORESEG_DEBUG": "debug",
    "PORESEG_GATHER_FUSED": "gather_fused", "PORESEG_SINGLE_PASS": "single_pass", "PORESEG_DOWNLOAD": "download_by_kernel", "PORESEG_K0_UNALIGNED": "k0_unaligned",
    # libporeseg_diag.so only (PORESEG_LIB=.../libporeseg_diag.so): stale or partial results, never the product
    "PORESEG_DBG_PHASE": "dbg_phase", "PORESEG_DBG_K0_NOGRP": "dbg_k0_nogrp", "PORESEG_SCAN_LDS_PAD": "scan_lds_pad", "PORESEG_REP_STAGE": "rep_stage",
}


def options_from_env(environ=None):
    """(options, tiling, pool overrides) named by PORESEG_* variables -- for tests and tools, which call it explicitly."""
    env = os.environ if environ is None else environ
    opts = {}
    for var, name in _ENV_OPTIONS.items():
        if var in env and env[var] != "":
            opts[name] = int(env[var])
    if env.get("PORESEG_STITCH"):
        opts["stitch_host"] = 1 if env["PORESEG_STITCH"] == "host" else 0
    if "PORESEG_NOISE_K" in env:
        opts["noise_k_ppm"] = int(round(float(env["PORESEG_NOISE_K"]) * 1e6))
    tiling = [int(env.get("PORESEG_TILE", "0") or 0), int(env.get("PORESEG_HALO", "0") or 0)]
    pool = {}
    if "PORESEG_POOL_K0_WAVES" in env:
        pool["k0_waves"] = int(env["PORESEG_POOL_K0_WAVES"])
    if "PORESEG_POOL_K0_MAX" in env:
        pool["k0_admit"] = int(env["PORESEG_POOL_K0_MAX"])
    if env.get("PORESEG_POOL_SHARED", "0") == "1":
        pool["k0_shared"] = 1
    return opts, tiling, pool


def apply_env_defaults(environ=None):
    """Contexts created from now on start with the settings the PORESEG_* variables name (tests/conftest.py, tools/)."""
    opts, tiling, pool = options_from_env(environ)
    DEFAULT_OPTIONS.update(opts)
    if tiling[0] or tiling[1]:
        DEFAULT_TILING[:] = tiling
    POOL_OVERRIDES.update(pool)
    return opts, tiling, pool


def _serialised(method):
    """The method runs under its Context's lock."""
    import functools

    @functools.wraps(method)
    def call(self, *a, **kw):
        with self.lock:
            return method(self, *a, **kw)
    return call


class Context(object):
    """Owns a ps_ctx bound to GPU `device` (one process per GPU; one host thread at a time)."""

    def __init__(self, device=0):
        L = _lib.lib()
        if not torch.cuda.is_available():
            raise RuntimeError("pypore_amd: no GPU visible to torch -- the segmenter has no CPU fallback")
        self.device = int(device)
        h = ctypes.c_void_p()
        _lib.check(L.ps_create(self.device, None, ctypes.byref(h)))
        self.handle = h
        self.L = L
        self.lock = threading.RLock()                   # one call at a time per ps_ctx (see engine.context)
        if DEFAULT_TILING[0] or DEFAULT_TILING[1]:
            self.set_tiling(*DEFAULT_TILING)
        for name, value in DEFAULT_OPTIONS.items():
            self.set_option(name, value)

    def close(self):
        if self.handle:
            self.L.ps_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_tiling(self, tile_len=0, halo=0):
        with self.lock:
            _lib.check(self.L.ps_set_tiling(self.handle, int(tile_len), int(halo)), self.handle)

    def set_option(self, name, value):
        """ps_set_option, under the context's lock: an option never changes in the middle of another thread's call."""
        with self.lock:
            _lib.check(self.L.ps_set_option(self.handle, name.encode(), int(value)), self.handle)

    def seq_ms(self):
        """Device time of the most recent segment call (HIP events, first upload .. last result copy), in ms: the
        light accessor for timed loops (timings() builds a dict of everything)."""
        if not hasattr(self, "_tm"):
            self._tm = ((ctypes.c_double * 8)(), (ctypes.c_int64 * 8)())
        self.L.ps_get_timings(self.handle, self._tm[0], 8, self._tm[1], 8)
        return self._tm[0][7]

    def timings(self):
        ms = (ctypes.c_double * 8)()
        cnt = (ctypes.c_int64 * 14)()
        _lib.check(self.L.ps_get_timings(self.handle, ms, 8, cnt, 14))
        return dict(spine_ms=ms[0], tree_ms=ms[1], gather_ms=ms[2], total_ms=ms[3], stitch_ms=ms[4], bridge_ms=ms[5], blocksum_ms=ms[6], seq_ms=ms[7],
                    windows=cnt[0], candidates=cnt[1], tiles=cnt[2], tree_jobs=cnt[3], repairs=cnt[4],
                    exact_rescans=cnt[5], full_exact_scans=cnt[6], wide_redo=cnt[7],
                    windows_spine=cnt[8], windows_bridge=cnt[9], windows_tree=cnt[10], near_ties=cnt[11],
                    helper_chunks_published=cnt[12], helper_chunks_taken=cnt[13])

    def near_ties(self):
        """Windows of the most recent segment call that were decided among fp64 contenders with a margin inside the
        noise of the device logarithm against glibc's (1e-9 relative): the reference could have decided
        them the other way.  0 in every golden vector except the constructed exact tie.  -1: the call ran on the LDS-window
        kernels, which do not count them (callers that redo near ties on the exact route treat it as "some")."""
        if not hasattr(self, "_cnt"):
            self._cnt = self.L.ps_counters(self.handle)      # the context's counters in place: no call per look
        return int(self._cnt[11])

    # ---- the hot path ---------------------------------------------------------------------------
    @_serialised
    def segment_batch(self, samples, ev_off, params, quantum, offset_counts=0, want_stats=True, cap=None,
                      want_spine=False, lead=0, out=None):
        """ps_segment_batch on device-resident `samples` (torch float32 or int16 CUDA tensor).
        Returns (bounds int32 CUDA tensor [total], bounds_off int64 numpy [n_ev+1],
        stats float64 CUDA tensor [total+n_ev, 4] or None).  lead > 0: the boundaries are a view that starts `lead`
        elements into their buffer (dist.BoundaryGather puts its header there and sends the buffer as it is).
        out: int32 CUDA tensor to receive the boundaries (its size is the capacity; ValueError if they do not fit)."""
        assert samples.is_cuda and samples.is_contiguous() and samples.dim() == 1
        if samples.dtype == torch.float32:
            dtype = _lib.PS_DTYPE_F32
        elif samples.dtype == torch.int16:
            dtype = _lib.PS_DTYPE_I16
        else:
            raise ValueError("samples must be float32 or int16, got %s" % samples.dtype)
        ev_off = np.ascontiguousarray(ev_off, dtype=np.int64)
        n_ev = ev_off.size - 1
        fmt = _lib.SampleFormat(dtype, int(offset_counts), float(quantum))
        off_p = ev_off.ctypes.data_as(ctypes.POINTER(ctypes.c_int64))
        if cap is None and out is None:
            cap = int(self.L.ps_bounds_capacity(off_p, n_ev, int(params.min_width)))
            if cap < 0:
                raise ValueError("min_width must be >= 1")
        dev = samples.device
        if out is not None:
            assert out.is_cuda and out.is_contiguous() and out.dtype == torch.int32 and out.dim() == 1
            bounds, cap = out, int(out.numel())
        else:
            bounds = torch.empty(max(cap, 1) + lead, dtype=torch.int32, device=dev)[lead:]
        stats = torch.empty((max(cap, 1) + n_ev, 4), dtype=torch.float64, device=dev) if want_stats else None
        boff = np.zeros(n_ev + 1, dtype=np.int64)
        torch.cuda.current_stream(dev).synchronize()      # inputs produced on torch's stream are ready
        spine = torch.empty(max(cap, 1), dtype=torch.uint8, device=dev) if want_spine else None
        rc = self.L.ps_segment_batch_ex(self.handle, ctypes.c_void_p(samples.data_ptr()), ctypes.byref(fmt), off_p,
                                        n_ev, ctypes.byref(params), ctypes.c_void_p(bounds.data_ptr()), cap,
                                        boff.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)),
                                        ctypes.c_void_p(stats.data_ptr()) if want_stats else None,
                                        ctypes.c_void_p(spine.data_ptr()) if want_spine else None)
        _lib.check(rc, self.handle)
        if NEAR_TIE_WARNING:
            nt = self.near_ties()
            if nt > 0:
                import warnings
                warnings.warn("%d window(s) were decided by a margin inside the reference's own rounding noise (near tie: below 1e-9 "
                              "relative on grid data -- the logarithm's last bit; on re-quantised float64 data, e.g. a filtered event, "
                              "within the noise of the reference's cumsums): the reference could place such a boundary elsewhere, "
                              "typically one sample beside it" % nt, NearTieWarning, stacklevel=2)
        total = int(boff[-1])
        if want_spine:
            return bounds[:total], boff, (stats[:total + n_ev] if want_stats else None), spine[:total]
        return bounds[:total], boff, (stats[:total + n_ev] if want_stats else None)

    def _fmt(self, samples, quantum, offset_counts):
        if samples.dtype == torch.float32:
            return _lib.SampleFormat(_lib.PS_DTYPE_F32, int(offset_counts), float(quantum))
        if samples.dtype == torch.int16:
            return _lib.SampleFormat(_lib.PS_DTYPE_I16, int(offset_counts), float(quantum))
        raise ValueError("samples must be float32 or int16, got %s" % samples.dtype)

    @_serialised
    def detect_events(self, samples, quantum, threshold=90.0, min_duration=100000, min_current=-0.5,
                      offset_counts=0):
        """ps_detect_events on a device-resident trace: (starts, lengths) int64 numpy arrays of the
        events lambda_event_parser(threshold) keeps with its default rules (parsers.py:124-155)."""
        assert samples.is_cuda and samples.is_contiguous() and samples.dim() == 1
        fmt = self._fmt(samples, quantum, offset_counts)
        n = samples.numel()
        # (room for every event the rules can keep is n / min_duration + 2 -- gigabytes of host memory for a long trace and a
        #  small min_duration; a file has a few hundred: start with EVENT_CAP and take the count the library reports if it is more)
        cap = min(n // max(1, int(min_duration)) + 2, EVENT_CAP)
        cnt = ctypes.c_int64()
        torch.cuda.current_stream(samples.device).synchronize()
        P64 = ctypes.POINTER(ctypes.c_int64)
        while True:
            st = np.zeros(cap, dtype=np.int64)
            ln = np.zeros(cap, dtype=np.int64)
            rc = self.L.ps_detect_events(self.handle, ctypes.c_void_p(samples.data_ptr()), ctypes.byref(fmt), n,
                                         float(threshold), int(min_duration), float(min_current),
                                         st.ctypes.data_as(P64), ln.ctypes.data_as(P64), cap, ctypes.byref(cnt))
            if rc == _lib.PS_ERR_CAPACITY and cnt.value > cap:
                cap = int(cnt.value)
                continue
            _lib.check(rc, self.handle)
            return st[:cnt.value].copy(), ln[:cnt.value].copy()

    @_serialised
    def detect_segment_trace(self, samples, quantum, params, threshold=90.0, min_duration=100000, min_current=-0.5,
                             offset_counts=0, want_stats=False):
        """ps_detect_segment_trace: detect_events + segment_events of a device-resident file trace in one call and one pass
        over its samples (include/poreseg.h).  Returns (starts, lengths, bounds, bounds_off, stats or None)."""
        assert samples.is_cuda and samples.is_contiguous() and samples.dim() == 1
        fmt = self._fmt(samples, quantum, offset_counts)
        n = samples.numel()
        ev_cap = min(n // max(1, int(min_duration)) + 2, EVENT_CAP)      # (see detect_events)
        cnt = ctypes.c_int64()
        dev = samples.device
        P64 = ctypes.POINTER(ctypes.c_int64)
        torch.cuda.current_stream(dev).synchronize()
        while True:
            # (every event holds at most length / min_width boundaries, all of them together n / min_width)
            cap = n // int(params.min_width) + 1
            st = np.zeros(ev_cap, dtype=np.int64)
            ln = np.zeros(ev_cap, dtype=np.int64)
            boff = np.zeros(ev_cap + 1, dtype=np.int64)
            bounds = torch.empty(max(cap, 1), dtype=torch.int32, device=dev)
            stats = torch.empty((max(cap, 1) + ev_cap, 4), dtype=torch.float64, device=dev) if want_stats else None
            rc = self.L.ps_detect_segment_trace(self.handle, ctypes.c_void_p(samples.data_ptr()), ctypes.byref(fmt), n,
                                                float(threshold), int(min_duration), float(min_current), ctypes.byref(params),
                                                st.ctypes.data_as(P64), ln.ctypes.data_as(P64), ev_cap, ctypes.byref(cnt),
                                                ctypes.c_void_p(bounds.data_ptr()), cap, boff.ctypes.data_as(P64),
                                                ctypes.c_void_p(stats.data_ptr()) if want_stats else None)
            if rc == _lib.PS_ERR_CAPACITY and cnt.value > ev_cap:        # more events than EVENT_CAP: once more with room for all
                ev_cap = int(cnt.value)
                continue
            _lib.check(rc, self.handle)
            break
        n_ev = int(cnt.value)
        total = int(boff[n_ev])
        return st[:n_ev].copy(), ln[:n_ev].copy(), bounds[:total], boff[:n_ev + 1].copy(), (stats[:total + n_ev] if want_stats else None)

    @_serialised
    def segment_events(self, samples, ev_start, ev_len, params, quantum, offset_counts=0, want_stats=False):
        """ps_segment_events: events are sub-ranges [start, start+len) of one device-resident trace."""
        assert samples.is_cuda and samples.is_contiguous() and samples.dim() == 1
        fmt = self._fmt(samples, quantum, offset_counts)
        ev_start = np.ascontiguousarray(ev_start, dtype=np.int64)
        ev_len = np.ascontiguousarray(ev_len, dtype=np.int64)
        n_ev = ev_start.size
        cap = int(np.sum(ev_len // int(params.min_width) + 1)) if n_ev else 0
        dev = samples.device
        bounds = torch.empty(max(cap, 1), dtype=torch.int32, device=dev)
        stats = torch.empty((max(cap, 1) + n_ev, 4), dtype=torch.float64, device=dev) if want_stats else None
        boff = np.zeros(n_ev + 1, dtype=np.int64)
        P64 = ctypes.POINTER(ctypes.c_int64)
        torch.cuda.current_stream(dev).synchronize()
        _lib.check(self.L.ps_segment_events(self.handle, ctypes.c_void_p(samples.data_ptr()), ctypes.byref(fmt),
                                            ev_start.ctypes.data_as(P64), ev_len.ctypes.data_as(P64), n_ev,
                                            ctypes.byref(params), ctypes.c_void_p(bounds.data_ptr()), cap,
                                            boff.ctypes.data_as(P64),
                                            ctypes.c_void_p(stats.data_ptr()) if want_stats else None, None),
                   self.handle)
        total = int(boff[-1])
        return bounds[:total], boff, (stats[:total + n_ev] if want_stats else None)

    @_serialised
    def segment_exact_f64(self, current, ev_start, ev_len, params):
        """ps_segment_exact_f64: events [start, start + len) of a float64 CUDA tensor (pA, on no grid) segmented from the
        reference's own sequential prefix sums (include/poreseg.h).  Returns (bounds int32 CUDA tensor, bounds_off)."""
        assert current.is_cuda and current.is_contiguous() and current.dim() == 1 and current.dtype == torch.float64
        ev_start = np.ascontiguousarray(ev_start, dtype=np.int64)
        ev_len = np.ascontiguousarray(ev_len, dtype=np.int64)
        n_ev = ev_start.size
        cap = int(np.sum(ev_len // int(params.min_width) + 1)) if n_ev else 0
        bounds = torch.empty(max(cap, 1), dtype=torch.int32, device=current.device)
        boff = np.zeros(n_ev + 1, dtype=np.int64)
        P64 = ctypes.POINTER(ctypes.c_int64)
        torch.cuda.current_stream(current.device).synchronize()
        _lib.check(self.L.ps_segment_exact_f64(self.handle, ctypes.c_void_p(current.data_ptr()), ev_start.ctypes.data_as(P64),
                                               ev_len.ctypes.data_as(P64), n_ev, ctypes.byref(params),
                                               ctypes.c_void_p(bounds.data_ptr()), cap, boff.ctypes.data_as(P64)), self.handle)
        return bounds[:int(boff[-1])], boff

    @_serialised
    def best_single_split(self, samples, quantum, offset_counts=0):
        fmt = _lib.SampleFormat(_lib.PS_DTYPE_F32 if samples.dtype == torch.float32 else _lib.PS_DTYPE_I16,
                                int(offset_counts), float(quantum))
        g = ctypes.c_double()
        i = ctypes.c_int32()
        torch.cuda.current_stream(samples.device).synchronize()
        _lib.check(self.L.ps_best_single_split(self.handle, ctypes.c_void_p(samples.data_ptr()), ctypes.byref(fmt),
                                               samples.numel(), ctypes.byref(g), ctypes.byref(i)), self.handle)
        return g.value, i.value

    @_serialised
    def audit_bounds(self, samples, quantum, params, windows, offset_counts=0):
        """ps_audit_bounds (diagnostic): the pruning bounds of the block-sum scan -- corner, two-boundary, group -- against
        the gains they cover, evaluated on the device from the raw samples, for the given windows [(ps, pe), ...] of the
        trace taken as one event.  Returns a dict of counts, violations and smallest margins per kind."""
        fmt = self._fmt(samples, quantum, offset_counts)
        w = np.ascontiguousarray(np.asarray(windows, dtype=np.int32).reshape(-1, 2))
        out = (ctypes.c_double * 12)()
        torch.cuda.current_stream(samples.device).synchronize()
        _lib.check(self.L.ps_audit_bounds(self.handle, ctypes.c_void_p(samples.data_ptr()), ctypes.byref(fmt), samples.numel(),
                                          ctypes.byref(params), w.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), int(w.shape[0]), out),
                   self.handle)
        o = list(out)
        return {"corner": {"blocks": int(o[0]), "violations": int(o[1]), "min_margin": o[6]},
                "two_boundary": {"blocks": int(o[2]), "violations": int(o[3]), "min_margin": o[7]},
                "group": {"groups": int(o[4]), "violations": int(o[5]), "min_margin": o[8]},
                "windows_with_coarse_pass": int(o[9])}

    @_serialised
    def score_window(self, samples, quantum, min_width, min_gain, offset_counts=0):
        fmt = _lib.SampleFormat(_lib.PS_DTYPE_F32 if samples.dtype == torch.float32 else _lib.PS_DTYPE_I16,
                                int(offset_counts), float(quantum))
        scores = torch.empty(max(1, samples.numel()), dtype=torch.float64, device=samples.device)
        i = ctypes.c_int32()
        torch.cuda.current_stream(samples.device).synchronize()
        _lib.check(self.L.ps_score_window(self.handle, ctypes.c_void_p(samples.data_ptr()), ctypes.byref(fmt),
                                          samples.numel(), int(min_width), float(min_gain),
                                          ctypes.c_void_p(scores.data_ptr()), ctypes.byref(i)), self.handle)
        return i.value, scores[:samples.numel()]

    @_serialised
    def filter_bessel(self, samples, quantum, cutoff=2000., sampling_freq=1.e5, order=1, offset_counts=0):
        """ps_filter_bessel: Event.filter (DataTypes.py:258-274) -- Bessel low-pass of order 1..8, forward and backward
        (scipy filtfilt semantics); returns the filtered current in pA as a float64 CUDA tensor.  `samples`: float32 pA on
        the grid `quantum` / int16 counts (what a file holds), or a float64 tensor of pA on no grid at all -- the current
        of an event that was filtered before (quantum is ignored then)."""
        assert samples.is_cuda and samples.is_contiguous() and samples.dim() == 1
        if samples.dtype == torch.float64:
            fmt = _lib.SampleFormat(_lib.PS_DTYPE_F64, 0, 1.0)
        else:
            fmt = self._fmt(samples, quantum, offset_counts)
        out = torch.empty(max(1, samples.numel()), dtype=torch.float64, device=samples.device)
        torch.cuda.current_stream(samples.device).synchronize()
        _lib.check(self.L.ps_filter_bessel(self.handle, ctypes.c_void_p(samples.data_ptr()), ctypes.byref(fmt),
                                           samples.numel(), int(order), float(cutoff), float(sampling_freq),
                                           ctypes.c_void_p(out.data_ptr())), self.handle)
        return out[:samples.numel()]

    @_serialised
    def filter_requantise_batch(self, samples, ev_start, ev_len, quantum, cutoff=2000., sampling_freq=1.e5, order=1, offset_counts=0):
        """ps_filter_requantise_batch: Event.filter + the re-quantisation for many events of ONE device-resident trace in one
        call (two host synchronisations for the batch instead of three per event).  Returns (filtered float64 CUDA tensor,
        rounded float32 CUDA tensor -- both the events back to back --, offsets int64 numpy [n_ev + 1], centres, steps)."""
        assert samples.is_cuda and samples.is_contiguous() and samples.dim() == 1
        fmt = _lib.SampleFormat(_lib.PS_DTYPE_F64, 0, 1.0) if samples.dtype == torch.float64 else self._fmt(samples, quantum, offset_counts)
        ev_start = np.ascontiguousarray(ev_start, dtype=np.int64)
        ev_len = np.ascontiguousarray(ev_len, dtype=np.int64)
        n_ev = ev_start.size
        off = np.concatenate(([0], np.cumsum(ev_len))).astype(np.int64)
        total = int(off[-1])
        filtered = torch.empty(max(1, total), dtype=torch.float64, device=samples.device)
        rounded = torch.empty(max(1, total), dtype=torch.float32, device=samples.device)
        centre = np.zeros(max(1, n_ev), dtype=np.float64)
        step = np.zeros(max(1, n_ev), dtype=np.float64)
        P64, PD = ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_double)
        torch.cuda.current_stream(samples.device).synchronize()
        _lib.check(self.L.ps_filter_requantise_batch(self.handle, ctypes.c_void_p(samples.data_ptr()), ctypes.byref(fmt),
                                                     ev_start.ctypes.data_as(P64), ev_len.ctypes.data_as(P64), n_ev, int(order),
                                                     float(cutoff), float(sampling_freq), ctypes.c_void_p(filtered.data_ptr()),
                                                     ctypes.c_void_p(rounded.data_ptr()), centre.ctypes.data_as(PD), step.ctypes.data_as(PD)),
                   self.handle)
        return filtered[:total], rounded[:total], off, centre[:n_ev], step[:n_ev]

    @_serialised
    def requantise(self, current):
        """ps_requantise: a filtered current (float64 CUDA tensor, pA) centred and rounded to the finest power-of-two
        grid that keeps its counts below 2**22.  Returns (float32 CUDA tensor on that grid, centre, step): segment it
        with quantum = step."""
        assert current.is_cuda and current.is_contiguous() and current.dim() == 1 and current.dtype == torch.float64
        out = torch.empty(max(1, current.numel()), dtype=torch.float32, device=current.device)
        centre, step = ctypes.c_double(), ctypes.c_double()
        torch.cuda.current_stream(current.device).synchronize()
        _lib.check(self.L.ps_requantise(self.handle, ctypes.c_void_p(current.data_ptr()), current.numel(),
                                        ctypes.c_void_p(out.data_ptr()), ctypes.byref(centre), ctypes.byref(step)), self.handle)
        return out[:current.numel()], centre.value, step.value

    @_serialised
    def align_batch(self, model_means, model_stds, model_durs, skip_penalty, backslip_penalty,
                    seq_means, seq_stds, seq_durs, seq_off):
        """ps_align_batch: cSegmentAligner.align (calignment.pyx:20-100) for a batch of sequences.  Model arrays: host
        (numpy); sequence arrays: float64 CUDA tensors, sequence q = [seq_off[q], seq_off[q+1]).  Returns CUDA tensors
        (scores float64 [n_seq] = score[s-1][m-1], paths uint32-as-int64 view avoided: int32 bits of the reference's
        unsigned j [total], status int32 [n_seq])."""
        mm, ms, md = (np.ascontiguousarray(a, dtype=np.float64) for a in (model_means, model_stds, model_durs))
        assert mm.size == ms.size == md.size
        off = np.ascontiguousarray(seq_off, dtype=np.int64)
        n_seq = off.size - 1
        for t in (seq_means, seq_stds, seq_durs):
            assert t.is_cuda and t.is_contiguous() and t.dtype == torch.float64 and t.numel() >= int(off[-1])
        dev = seq_means.device
        scores = torch.zeros(max(n_seq, 1), dtype=torch.float64, device=dev)
        paths = torch.zeros(max(int(off[-1]), 1), dtype=torch.int32, device=dev)
        status = torch.zeros(max(n_seq, 1), dtype=torch.int32, device=dev)
        torch.cuda.current_stream(dev).synchronize()
        dp = ctypes.POINTER(ctypes.c_double)
        _lib.check(self.L.ps_align_batch(self.handle, mm.ctypes.data_as(dp), ms.ctypes.data_as(dp), md.ctypes.data_as(dp),
                                         mm.size, float(skip_penalty), float(backslip_penalty),
                                         ctypes.c_void_p(seq_means.data_ptr()), ctypes.c_void_p(seq_stds.data_ptr()),
                                         ctypes.c_void_p(seq_durs.data_ptr()),
                                         off.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)), n_seq,
                                         ctypes.c_void_p(scores.data_ptr()), ctypes.c_void_p(paths.data_ptr()),
                                         ctypes.c_void_p(status.data_ptr())), self.handle)
        return scores[:n_seq], paths[:int(off[-1])], status[:n_seq]

    @_serialised
    def synth_trace(self, n, seed, seg_end, level_counts, dtype=torch.float32, start=0):
        """Synthetic step trace generated directly in HBM (csrc synth_kernel == pypore_amd.synth).  start > 0: the
        samples [start, start + n) of the trace the table describes (a rank's piece of one long trace): the noise hash
        of sample i is splitmix64(seed + (i + 1) * GOLDEN), so the piece is the same generator with a shifted seed and a
        table cut at `start`."""
        out = torch.empty(n, dtype=dtype, device="cuda:%d" % self.device)
        seg_end = np.ascontiguousarray(seg_end, dtype=np.int64)
        level_counts = np.ascontiguousarray(level_counts, dtype=np.int32)
        if start:
            first = int(np.searchsorted(seg_end, start, side="right"))
            seg_end = np.ascontiguousarray(seg_end[first:] - start)
            level_counts = np.ascontiguousarray(level_counts[first:])
            seed = (int(seed) + int(start) * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
        torch.cuda.current_stream(out.device).synchronize()
        _lib.check(self.L.ps_synth_trace(self.handle, ctypes.c_void_p(out.data_ptr()),
                                         _lib.PS_DTYPE_F32 if dtype == torch.float32 else _lib.PS_DTYPE_I16,
                                         n, ctypes.c_uint64(int(seed) & 0xFFFFFFFFFFFFFFFF),
                                         seg_end.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)),
                                         level_counts.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), seg_end.size),
                   self.handle)
        return out


class NearTieWarning(UserWarning):
    """A segment call decided at least one window by a margin inside the reference's own rounding noise: the logarithm's last
    bit on grid data, the noise of its uncentred cumsums on re-quantised float64 data (Context.near_ties)."""


NEAR_TIE_WARNING = True      # set False to skip the counter read after every call (one ps_get_timings, ~1 us)


class StreamPool(object):
    """T contexts on one GPU (each a ps_ctx with its own HIP stream and scratch) fed by T host threads.

    One call of the path is a chain of kernels with idle stretches -- the tails of the spine and subtree kernels, the
    single-workgroup stitch kernels, the seam bridges that occupy a fraction of the chip.  Batches are independent
    (events and files are the reference's own unit of work, DataTypes.py:968-984), so the kernels of one batch fill the
    idle stretches of another when they are submitted on different streams: two streams deliver 1.4x the batches per
    second of one on the 1e8-sample bench trace, with bit-identical results.  ctypes drops the GIL for the duration of a
    call, so plain Python threads are enough.  Every call still ends with its own stream synchronisation.

    A host that keeps the pool busy for long should take CPython's cyclic collector out of the way (`gc.collect();
    gc.freeze()` once everything is set up): a full collection holds the GIL for tens of milliseconds and every worker
    then waits for it on its way out of the C call -- 0.33 instead of 0.35-0.43 ms per 1e8-sample batch over runs of
    400-2000 batches (tools/pool_stalls.py).

    HIP maps a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4): set it to at least the number of
    contexts before the runtime starts (bench.py: 16).  Round 4, 1e8-sample trace: 4 / 8 / 16 contexts in flight deliver a
    batch every 0.216 / 0.202 / 0.189 ms."""

    def __init__(self, device=None, streams=2, overrides=None):
        """overrides: ps_set_option name -> value applied to the pool's contexts while it runs, on top of "shared_device"
        (default: engine.POOL_OVERRIDES, empty in the product)."""
        import queue
        import threading
        base = context(device)
        self.contexts = [base] + [Context(base.device) for _ in range(max(1, int(streams)) - 1)]
        self.overrides = dict(POOL_OVERRIDES if overrides is None else overrides)
        self._shared_now = {}                            # id(context) -> the "shared_device" value it was last given
        # persistent workers for contexts 1 .. T-1 (the caller's thread drives context 0): a run() costs a queue
        # hand-over per worker, not a thread start
        self._inbox = [queue.SimpleQueue() for _ in self.contexts]
        self._done = queue.SimpleQueue()
        self._threads = []
        for t in range(1, len(self.contexts)):
            th = threading.Thread(target=self._worker, args=(t,), daemon=True)
            th.start()
            self._threads.append(th)

    def __len__(self):
        return len(self.contexts)

    def _share(self, t, n_jobs, job, results, errors, ticket=None):
        try:
            if ticket is None:                         # fixed shares: job k on context k % T
                for k in range(t, n_jobs, len(self.contexts)):
                    results[k] = job(self.contexts[t], k, t)
            else:                                      # whoever is free takes the next job
                while True:
                    k = next(ticket)                   # (itertools.count: one C call, atomic under the GIL)
                    if k >= n_jobs:
                        break
                    results[k] = job(self.contexts[t], k, t)
        except BaseException as e:                     # noqa: B036 -- re-raised on the submitting thread
            errors.append(e)

    def _worker(self, t):
        while True:
            task = self._inbox[t].get()
            if task is None:
                return
            self._share(t, *task)
            self._done.put(t)

    def close(self):
        for t in range(1, len(self.contexts)):
            self._inbox[t].put(None)
        self._threads = []

    # What a context that shares the chip wants (ps_set_option "shared_device" n, include/poreseg.h; measured in round 5):
    #   * K0 as a persistent kernel at one wave per SIMD: beside the scan waves of other calls a K0 that takes every wave slot
    #     it can get is in the way (sixteen in flight: 0.193 -> 0.170 ms per step); a lone call is faster with the one-shot K0;
    #   * no look-ahead helpers: with sixteen calls in flight the other calls are the better use of an idle workgroup (a trace
    #     without steps 1.2 -> 8.6 ms per step with them on); for a lone call they are worth 2 x on sparse traces;
    #   * more than three contexts: at most three K0s in flight (K0 is bound by HBM; sixteen of them at once finish late together,
    #     with every call's scan kernels behind them: 0.188 -> 0.175 ms per step).
    # The front stream ("k0_shared") lost (0.26-0.30 ms per step) and is set by nothing; overrides can name it for the record.
    def _configure(self, cx, n):
        """context `cx` shares its device with n - 1 others (n <= 1: a lone context again)"""
        if not hasattr(cx, "set_option") or self._shared_now.get(id(cx)) == n:
            return
        cx.set_option("shared_device", n)
        if n > 1:
            for name, value in self.overrides.items():
                cx.set_option(name, value)
        else:
            if "k0_shared" in self.overrides:
                cx.set_option("k0_shared", 0)
            if _device_shared(getattr(cx, "device", None), cx):
                cx.set_option("lat_help", 0)             # (other threads' contexts are alive on the device: see _note_live)
        # (what the process was started with -- tests / tools through engine.DEFAULT_OPTIONS -- stays in force)
        for name in ("k0_waves", "lat_help", "k0_admit"):
            if name in DEFAULT_OPTIONS:
                cx.set_option(name, DEFAULT_OPTIONS[name])
        self._shared_now[id(cx)] = n

    def run(self, n_jobs, job, dynamic=False):
        """Runs job(ctx, k, t) for k = 0 .. n_jobs-1 and returns the results in job order.  Job k runs on context
        t = k % T, or -- dynamic=True -- on whichever context is free next (jobs of unequal length).  While more than one
        job is in flight the contexts are configured for a shared device; afterwards the caller's own context
        (contexts[0], the one SpeedyStatSplit.parse uses on this thread) is a lone context again, whatever happened."""
        import itertools
        T = len(self.contexts)
        results = [None] * n_jobs
        errors = []
        n_shared = T if (T > 1 and n_jobs > 1) else 1
        try:
            for cx in (self.contexts if n_shared > 1 else self.contexts[:1]):
                self._configure(cx, n_shared)
            ticket = itertools.count() if dynamic else None
            busy = [t for t in range(1, T) if t < n_jobs]
            for t in busy:
                self._inbox[t].put((n_jobs, job, results, errors, ticket))
            self._share(0, n_jobs, job, results, errors, ticket)
            for _ in busy:
                self._done.get()
        finally:
            self._configure(self.contexts[0], 1)         # (every option the run changed, k0_admit included)
        if errors:
            raise errors[0]
        return results

    def segment_many(self, batches, params, quantum, offset_counts=0, want_stats=False):
        """batches: list of (samples CUDA tensor, ev_off); one ps_segment_batch each, spread over the streams.  Returns
        the list of (bounds, bounds_off, stats) in batch order."""
        return self.run(len(batches), lambda ctx, k, t: ctx.segment_batch(batches[k][0], batches[k][1], params, quantum,
                                                                          offset_counts=offset_counts, want_stats=want_stats))


_thread_contexts = threading.local()
_contexts_lock = threading.Lock()


def context(device=None):
    """The Context of `device` (default: torch's current device) that belongs to the CALLING THREAD.

    A ps_ctx serves one call at a time (its stream, its scratch, its status block).  The reference is safe under the GIL
    (a new FastStatSplit per parse()); here ctypes drops the GIL for the duration of a call, so two threads that call
    SpeedyStatSplit.parse at once must not meet in one ps_ctx.  The main thread keeps the process-wide context of the
    device; every other thread gets one of its own on first use (its own HIP stream and scratch: calls of different threads
    overlap on the GPU like the contexts of a StreamPool) and gives it back when the thread ends.  Each Context also
    carries a lock that its calls take, for code that hands ONE Context to several threads."""
    if device is None:
        device = torch.cuda.current_device() if torch.cuda.is_available() else 0
    device = int(device)
    if threading.current_thread() is threading.main_thread():
        with _contexts_lock:
            fresh = device not in _contexts
            if fresh:
                _contexts[device] = Context(device)
            cx = _contexts[device]
        if fresh:
            _note_live(device, cx)
        return cx
    mine = getattr(_thread_contexts, "by_device", None)
    if mine is None:
        mine = _thread_contexts.by_device = {}
    if device not in mine:
        mine[device] = Context(device)
        _note_live(device, mine[device])
    return mine[device]


_live = {}                                              # device -> weak set of the contexts engine.context() handed out


def _device_shared(device, cx):
    with _contexts_lock:
        return any(c is not cx and getattr(c, "handle", None) for c in _live.get(device, ()))


def _note_live(device, cx):
    """The look-ahead kernel's helpers are for a call that has the chip to itself.  Whether a context is alone is a matter of
    how many contexts are alive on its device -- not of which thread made it: a program that does all its work on ONE worker
    thread (a notebook kernel, a server) keeps the helpers; once a second context appears on the device, all of them run
    without (the sharing may end when a thread does; the helpers then stay off for the survivors, which is the safe side)."""
    import weakref
    with _contexts_lock:
        live = _live.setdefault(device, weakref.WeakSet())
        live.add(cx)
        others = [c for c in live if c is not cx and getattr(c, "handle", None)]
    if others and "lat_help" not in DEFAULT_OPTIONS:
        for c in others + [cx]:
            if hasattr(c, "set_option"):
                c.set_option("lat_help", 0)


# ---- host-side input normalisation -------------------------------------------------------------
def detect_quantum(x, max_bits=24, full=True):
    """Largest power of two q = 2**-k (0 <= k <= max_bits) such that every sample of the numpy
    array `x` is an integer multiple of q; ValueError if there is none (off-grid data).
    Real traces are int16 ADC counts times a scale (read_abf.py:202-210), so such a q exists
    whenever the scale is a power of two; pass quantum= explicitly to skip this host pass.
    The candidate is found on a strided subset (<= 65 536 samples) and then confirmed on the whole
    array, so a large trace costs two passes instead of one per bit.  With full=False the confirmation
    is left to the device, which checks every sample anyway (PS_ERR_OFF_GRID -> ValueError)."""
    x = np.asarray(x)
    if x.size == 0:
        return 1.0

    def integral(a, k):
        y = a * (2.0 ** k)
        return bool(np.all(y == np.rint(y)))

    def smallest_k(a):
        for k in (5,) + tuple(i for i in range(0, max_bits + 1) if i != 5):
            if integral(a, k):
                while k > 0 and integral(a, k - 1):        # coarsest grid that still holds
                    k -= 1
                return k
        return None

    sub = x[::max(1, x.size // 65536)]
    k = smallest_k(sub)
    if k is None:
        raise ValueError("samples are not on a power-of-two ADC grid; pass quantum= (pA per count)")
    if not full or sub.size == x.size or integral(x, k):
        return 2.0 ** -k
    k = smallest_k(x)                                      # the subset was too coarse a witness: full search
    if k is None:
        raise ValueError("samples are not on a power-of-two ADC grid; pass quantum= (pA per count)")
    return 2.0 ** -k


Samples = collections.namedtuple("Samples", "tensor quantum offset")      # pA = count * quantum + offset


def _int_counts_tensor(counts, dev):
    """int16 CUDA tensor of integer counts; ValueError when they leave the int16 range."""
    counts = np.asarray(counts)
    if counts.dtype != np.int16:
        if counts.size and (counts.min() < -32768 or counts.max() > 32767):
            raise ValueError("ADC counts beyond int16 on a grid that is not a power of two: not supported")
        counts = counts.astype(np.int16)
    if counts.size >= _STAGE_MIN:
        return _staged_upload(counts, dev)
    with warnings.catch_warnings():                     # (a read-only memmap of an .abf: the tensor is only copied from)
        warnings.simplefilter("ignore", UserWarning)
        return torch.from_numpy(np.ascontiguousarray(counts)).to(dev)


_STAGE_MIN = 1 << 22                                    # samples: below this one pageable copy is as fast
_STAGE_CHUNK = 1 << 24                                  # int16 samples per staging buffer (32 MB)
_stage = {}
_stage_lock = threading.Lock()                          # the staging buffers of a device are shared: one upload at a time


def _staged_upload(counts, dev):
    """A long int16 array (the data section of an .abf, usually a memmap of the page cache) -> CUDA tensor through two
    pinned staging buffers: the host's copy of chunk k+1 out of the page cache runs while chunk k crosses PCIe.  One
    `np.ascontiguousarray` of the whole file plus a pageable copy of it is 2-3 x slower (tools/bench_experiment.py)."""
    key = dev.index
    with _stage_lock:                                    # (two host threads uploading to one device would overwrite each other's chunks)
        if key not in _stage:
            _stage[key] = ([torch.empty(_STAGE_CHUNK, dtype=torch.int16, pin_memory=True) for _ in range(2)],
                           [torch.cuda.Event() for _ in range(2)], torch.cuda.Stream(device=dev))
        bufs, events, stream = _stage[key]
        out = torch.empty(counts.size, dtype=torch.int16, device=dev)
        # `out` comes from the caching allocator on the CURRENT stream: work still queued there on a recycled block must
        # be finished before the side stream writes into it
        stream.wait_stream(torch.cuda.current_stream(dev))
        used = [False, False]
        with torch.cuda.stream(stream):
            for k, a in enumerate(range(0, counts.size, _STAGE_CHUNK)):
                b = min(counts.size, a + _STAGE_CHUNK)
                i = k & 1
                if used[i]:
                    events[i].synchronize()              # the copy that last read this buffer is done
                np.copyto(bufs[i].numpy()[:b - a], counts[a:b])
                out[a:b].copy_(bufs[i][:b - a], non_blocking=True)
                events[i].record(stream)
                used[i] = True
        stream.synchronize()
    return out


def to_device(current, quantum=None, offset=None, device=None, full_detect=False):
    """Anything a caller of the parser API may hand over -> Samples(CUDA tensor, quantum, offset).

    * a GridArray (what abf.read_abf returns; slices of it): its int16 counts, 2 B/sample;
    * numpy int16: ADC counts, quantum as given (default 1);
    * numpy float64 / float32 pA: with `quantum` a power of two (or none given and a power-of-two grid found) the
      values go up as float32, exactly; otherwise the counts are recovered -- rint((x - offset) / quantum) when the
      grid is given, grid.affine_grid(x) when it is not -- and go up as int16.  Data that lies on no grid at all
      raises ValueError: nothing is ever rounded silently;
    * torch tensors (float32 / float64 / int16, host or device) are taken as they are (float64 must be exact in float32).
    """
    from .grid import affine_grid, grid_of
    dev = torch.device("cuda", torch.cuda.current_device() if device is None else int(device))
    g = grid_of(current) if quantum is None else None
    if g is not None:
        from .grid import Deferred
        if isinstance(current, Deferred) and current.counts is not None:
            # (a file's counts go up once: events cut from the file are stretches of the same device tensor)
            return Samples(current.device_counts(dev, _int_counts_tensor), g[1], g[2])
        return Samples(_int_counts_tensor(g[0], dev), g[1], g[2])
    if isinstance(current, torch.Tensor) or np.asarray(current).dtype == np.int16:
        t, q = to_device_samples(current, quantum, device, full_detect)
        return Samples(t, q, 0.0 if offset is None else float(offset))
    a = np.asarray(current)
    if a.ndim != 1:
        raise ValueError("Buffer has wrong number of dimensions (expected 1, got %d)" % a.ndim)
    if a.dtype not in (np.float64, np.float32):
        raise ValueError("Buffer dtype mismatch, expected 'double' but got %s" % a.dtype)
    off = 0.0 if offset is None else float(offset)
    pow2 = quantum is not None and quantum > 0 and np.log2(quantum) == np.rint(np.log2(quantum))
    if quantum is None or pow2:
        try:
            t, q = to_device_samples(a - off if off else a, quantum, device, full_detect)
            return Samples(t, q, off)
        except ValueError:
            if quantum is not None:
                raise
    if quantum is None:                             # no power-of-two grid: any scale and offset (a real .abf header)
        q, o, k = affine_grid(a)
        return Samples(_int_counts_tensor(k, dev), q, o)
    k = (a.astype(np.float64) - off) / float(quantum)
    kr = np.rint(k)
    if a.size and np.max(np.abs(k - kr)) > 1e-6:
        raise ValueError("samples are not integer multiples of quantum=%r above offset=%r" % (quantum, off))
    return Samples(_int_counts_tensor(kr.astype(np.int64), dev), float(quantum), off)


def to_device_samples(current, quantum=None, device=None, full_detect=False):
    """numpy float64/float32 (pA), numpy int16 (ADC counts) or torch tensor -> (CUDA tensor, quantum).
    Without `quantum` the grid is detected on a subset of the samples (full_detect=True: on all of them); a
    finer grid that only shows elsewhere makes the device call fail with ValueError (off grid)."""
    dev = torch.device("cuda", torch.cuda.current_device() if device is None else int(device))
    if isinstance(current, torch.Tensor):
        t = current
        if t.dtype == torch.float64:
            t32 = t.to(torch.float32)
            if not torch.equal(t32.to(torch.float64), t):
                raise ValueError("samples are not exactly representable in float32; pass int16 counts or a coarser grid")
            t = t32
        if t.dtype not in (torch.float32, torch.int16):
            raise ValueError("Buffer dtype mismatch, expected float64/float32 pA or int16 counts")
        if quantum is None:
            quantum = 1.0 if t.dtype == torch.int16 else detect_quantum(t.detach().cpu().numpy())
        return t.to(dev).contiguous(), quantum
    a = np.asarray(current)
    if a.ndim != 1:
        raise ValueError("Buffer has wrong number of dimensions (expected 1, got %d)" % a.ndim)
    if a.dtype == np.int16:
        return torch.from_numpy(np.ascontiguousarray(a)).to(dev), (1.0 if quantum is None else quantum)
    if a.dtype not in (np.float64, np.float32):
        raise ValueError("Buffer dtype mismatch, expected 'double' but got %s" % a.dtype)
    if quantum is None:
        quantum = detect_quantum(a, full=full_detect)
    a32 = a.astype(np.float32)
    if a.dtype == np.float64 and not np.array_equal(a32.astype(np.float64), a):
        raise ValueError("samples are not exactly representable in float32; pass int16 counts or a coarser grid")
    return torch.from_numpy(np.ascontiguousarray(a32)).to(dev), quantum
