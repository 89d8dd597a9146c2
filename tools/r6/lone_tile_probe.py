#!/usr/bin/env python3
"""Round 6: a LONE call's device time against the tile length (ps_set_tiling): shorter tiles = shorter sequential chains per tile
(the spine kernel's duration is one tile's chain), more speculative windows and seams.  10^8-sample bench trace, one context."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from pypore_amd import _lib, engine, synth
ctx = engine.context(0)
n = 100_000_000
params = _lib.split_params(min_width=100, max_width=1000000, window_width=10000, prior_segments_per_second=10., sampling_freq=1e5)
d = synth.dwell_table(2024, n); ends = np.cumsum(d); lv = synth.LEVEL_COUNTS[np.arange(len(d)) % 5].astype(np.int32)
trace = ctx.synth_trace(n, 2024, ends, lv, dtype=torch.float32)
ev_off = np.array([0, n], dtype=np.int64)
out = torch.empty(n // 100 + 1, dtype=torch.int32, device="cuda")
ref = None
for rep in range(2):
    for tiles in (1536, 2048, 3072, 4096, 6144, 8192, 1024):
        L = (n + tiles - 1) // tiles if tiles != 1536 else 0
        ctx.set_tiling(L, 0)
        for _ in range(3):
            b = ctx.segment_batch(trace, ev_off, params, synth.QUANTUM, want_stats=False, out=out)[0]
        if ref is None:
            ref = b.clone()
        assert torch.equal(b, ref)
        s = 0.0
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(30):
            ctx.segment_batch(trace, ev_off, params, synth.QUANTUM, want_stats=False, out=out); s += ctx.seq_ms()
        torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / 30 * 1e3
        ctx.set_option("timing", 2)
        acc = {}
        for _ in range(5):
            ctx.segment_batch(trace, ev_off, params, synth.QUANTUM, want_stats=False, out=out)
            for k, v in ctx.timings().items():
                acc[k] = acc.get(k, 0) + v / 5
        ctx.set_option("timing", 1)
        print("%5d tiles (%6d samples): lone call %.4f ms of device time (%.4f wall); K0 %.3f spine %.3f bridge %.3f stitch %.3f tree %.3f gather %.3f; windows %d (spine %d bridge %d tree %d)"
              % (int(acc["tiles"]), L or n // 1536, s / 30, wall, acc["blocksum_ms"], acc["spine_ms"], acc["bridge_ms"], acc["stitch_ms"], acc["tree_ms"], acc["gather_ms"],
                 acc["windows"], acc["windows_spine"], acc["windows_bridge"], acc["windows_tree"]), flush=True)
