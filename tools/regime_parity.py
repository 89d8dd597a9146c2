"""Parity in the regimes of tools/regime_sweep.py at a size the oracle finishes quickly (GPU box)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import oracle
from pypore_amd import _lib, engine, synth
from pypore_amd import engine as _ps_engine
_ps_engine.apply_env_defaults()           # tools take their settings from PORESEG_* variables; the product reads none
ctx = engine.context(0)
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 4_000_000
cases = [
    ((100, 400), {}), ((500, 5000), {}), ((100000, 1000000), {}), ((1000, 20000), dict(window_width=1000)),
    ((1000, 20000), dict(window_width=50000)), ((1000, 20000), dict(min_width=8, window_width=2000)),
    ((1000, 20000), dict(min_width=1000)), ((1000, 20000), dict(max_width=30000)),
    ((1000, 20000), dict(prior_segments_per_second=1000.)), ((1000, 20000), dict(prior_segments_per_second=0.01)),
    ((20000, 400000), dict(max_width=100000)), ((1000, 20000), dict(window_width=100000)),
]
bad = 0
for (lo, hi), extra in cases:
    d = synth.dwell_table(91, n, lo, hi)
    ends = np.cumsum(d)
    lv = synth.LEVEL_COUNTS[np.arange(len(d)) % 5].astype(np.int32)
    t = ctx.synth_trace(n, 91, ends, lv, dtype=torch.float32)
    kw = dict(min_width=100, max_width=1000000, window_width=10000, prior_segments_per_second=10., sampling_freq=1e5)
    kw.update(extra)
    x = t.cpu().numpy().astype(np.float64)
    ref = oracle.parse(x, **kw)
    off = np.array([0, n], dtype=np.int64)
    for tile in (0, 60000, 250000):
        ctx.set_tiling(tile, 0)
        for mode in (0, 2):
            ctx.set_option("mode", mode)
            b, boff, _ = ctx.segment_batch(t, off, _lib.split_params(**kw), synth.QUANTUM, want_stats=False)
            ok = np.array_equal(b.cpu().numpy(), ref)
            if not ok:
                bad += 1
                print("MISMATCH", (lo, hi), extra, "tile", tile, "mode", mode, len(ref), b.numel())
    ctx.set_option("mode", 0); ctx.set_tiling(0, 0)
    print("ok" if not bad else "..", (lo, hi), extra, len(ref), "boundaries")
print("problems:", bad)
sys.exit(1 if bad else 0)
