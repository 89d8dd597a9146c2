"""Perf-cliff hunt (GPU box): the 1e8-sample trace in different dwell / parameter regimes, default build.  Prints ms per
call and which pipeline ran (K0 time > 0: block-sum device stitch; repairs > 0: host-stitch fallback)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pypore_amd import _lib, engine, synth
from pypore_amd import engine as _ps_engine
_ps_engine.apply_env_defaults()           # tools take their settings from PORESEG_* variables; the product reads none
ctx = engine.context(0)
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
cases = [
    ("dwell 100-400", (100, 400), {}),
    ("dwell 500-5000", (500, 5000), {}),
    ("dwell 1e5-1e6", (100000, 1000000), {}),
    ("dwell 1e6-1e7", (1000000, 10000000), {}),
    ("dwell 1e7-5e7", (10000000, 50000000), {}),
    ("no step at all", (n + 1, n + 2), {}),
    ("no step at all, max_width 1e9", (n + 1, n + 2), dict(max_width=1000000000)),
    ("dwell 1e5-1e6 max_width 1e9", (100000, 1000000), dict(max_width=1000000000)),
    ("dwell 1000-20000 W=1000", (1000, 20000), dict(window_width=1000)),
    ("dwell 1000-20000 W=50000", (1000, 20000), dict(window_width=50000)),
    ("dwell 1000-20000 mw=8 W=2000", (1000, 20000), dict(min_width=8, window_width=2000)),
    ("dwell 1000-20000 mw=1000", (1000, 20000), dict(min_width=1000)),
    ("dwell 1000-20000 max_width=30000", (1000, 20000), dict(max_width=30000)),
    ("dwell 1000-20000 prior=1000", (1000, 20000), dict(prior_segments_per_second=1000.)),
    ("dwell 1000-20000 prior=0.01", (1000, 20000), dict(prior_segments_per_second=0.01)),
]
for name, (lo, hi), extra in cases:
    d = synth.dwell_table(77, n, lo, hi)
    ends = np.cumsum(d)
    lv = synth.LEVEL_COUNTS[np.arange(len(d)) % 5].astype(np.int32)
    t = ctx.synth_trace(n, 77, ends, lv, dtype=torch.float32)
    kw = dict(min_width=100, max_width=1000000, window_width=10000, prior_segments_per_second=10., sampling_freq=1e5)
    kw.update(extra)
    p = _lib.split_params(**kw)
    off = np.array([0, n], dtype=np.int64)
    try:
        for _ in range(2):
            b, boff, _ = ctx.segment_batch(t, off, p, synth.QUANTUM, want_stats=False)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3):
            b, boff, _ = ctx.segment_batch(t, off, p, synth.QUANTUM, want_stats=False)
        torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 3 * 1e3
        tm = ctx.timings()
        print("%-38s %8.3f ms  bounds %8d  windows %8d  K0 %.3f spine %.3f bridge %.3f tree %.3f stitch %.3f repairs %d full_exact %d"
              % (name, ms, b.numel(), tm["windows"], tm["blocksum_ms"], tm["spine_ms"], tm["bridge_ms"], tm["tree_ms"], tm["stitch_ms"],
                 tm["repairs"], tm["full_exact_scans"]))
    except Exception as ex:
        print("%-38s ERROR %r" % (name, ex))
    del t
