#!/usr/bin/env python3
"""Per-window latency of ONE wave: a 2e6-sample trace segmented as a single tile (the spine kernel is then one wave
walking ~400 windows alone on its SIMD), spine kernel time / windows.  PORESEG_LIB selects the build."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pypore_amd import _lib, engine, synth
from pypore_amd import engine as _ps_engine
_ps_engine.apply_env_defaults()           # tools take their settings from PORESEG_* variables; the product reads none
n, seed = 2_000_000, 2024
ctx = engine.context(0)
params = _lib.split_params(min_width=100, max_width=1000000, window_width=10000, prior_segments_per_second=10., sampling_freq=1e5)
for dwell in ((1000, 20000), (150, 1500), (30000, 200000)):
    d = synth.dwell_table(seed, n, *dwell); ends = np.cumsum(d); lv = synth.LEVEL_COUNTS[np.arange(len(d)) % 5].astype(np.int32)
    t = ctx.synth_trace(n, seed, ends, lv, dtype=torch.float32)
    ctx.set_tiling(4_000_000, 0)
    ctx.set_option("timing", 2)
    acc = []
    for _ in range(5):
        b = ctx.segment_batch(t, np.array([0, n]), params, synth.QUANTUM, want_stats=False)[0]
        tm = ctx.timings(); acc.append((tm["spine_ms"], tm["tree_ms"], tm["windows"], tm["tiles"]))
    sp = np.median([a[0] for a in acc]); tr = np.median([a[1] for a in acc])
    print("dwell %s: %d boundaries, tiles %d, spine %.3f ms, subtrees %.3f ms, windows (spine + 1/TREE_W of subtrees) %d" % (dwell, b.numel(), acc[-1][3], sp, tr, acc[-1][2]))
