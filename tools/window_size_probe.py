"""Time per window and per row of the spine / subtree kernels against the window width (one call at a time, 1e8-sample
bench trace): does a window whose digest fits the L2 share of its wave (W = 5 000: 10 KB, 2.5 MB per XCD of 256 waves)
cost less PER ROW than the default (W = 10 000: 20 KB, 5 MB per XCD against 4 MB of L2)?"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pypore_amd import _lib, engine, synth
from pypore_amd import engine as _ps_engine
_ps_engine.apply_env_defaults()           # tools take their settings from PORESEG_* variables; the product reads none
ctx = engine.context(0); n = 100_000_000
d = synth.dwell_table(1, n, 1000, 20000); lv = synth.LEVEL_COUNTS[np.arange(len(d)) % 5].astype(np.int32)
t = ctx.synth_trace(n, 1, np.cumsum(d), lv, dtype=torch.float32); off = np.array([0, n], dtype=np.int64)
out = torch.empty(n // 100 + 1, dtype=torch.int32, device="cuda")
for W in (2500, 5000, 10000, 20000, 40000):
    p = _lib.split_params(window_width=W, prior_segments_per_second=10.)
    for _ in range(3): ctx.segment_batch(t, off, p, synth.QUANTUM, want_stats=False, out=out)
    ctx.set_option("timing", 2); acc = {}
    for _ in range(5):
        b = ctx.segment_batch(t, off, p, synth.QUANTUM, want_stats=False, out=out)[0]
        tm = ctx.timings()
        for k, v in tm.items(): acc[k] = acc.get(k, 0) + v / 5
    ctx.set_option("timing", 1)
    rows = W / 8 / 63
    sp, tr = acc["spine_ms"] * 1e3, acc["tree_ms"] * 1e3
    ws, wt = acc["windows_spine"], acc["windows_tree"]
    # waves live ~60 % of the kernel: slot-time per window = kernel time * slots * 0.6 / windows
    print("W %6d: %5d boundaries, spine %.3f ms for %6d windows, subtrees %.3f ms for %6d windows; per window (2 048 slots, 60 %% alive): spine %.2f us (%.2f us per row of %.1f), subtrees %.2f us"
          % (W, b.numel(), sp / 1e3, ws, tr / 1e3, wt, sp * 2048 * 0.6 / ws, sp * 2048 * 0.6 / ws / rows, rows, tr * 2048 * 0.6 / max(wt, 1)))
