#!/usr/bin/env python3
"""Round 6: BASELINE config 2 (1 024 events x 50 000 samples, one call) with the default tiling (two tiles
of 25 000 per event) against one tile per event (ps_set_tiling 50 000: no seam, no bridge).  Lone call and T batches in flight,
interleaved.  usage: python tools/r6/config2_tile_probe.py [T]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np, torch
from pypore_amd import _lib, engine, synth
engine.apply_env_defaults()
T = int(sys.argv[1]) if len(sys.argv) > 1 else 16
n_ev, ln = 1024, 50000
e2, l2 = [], []
for e_ in range(n_ev):
    for k_ in range(5):
        e2.append(e_ * ln + (k_ + 1) * 10000); l2.append(int(synth.LEVEL_COUNTS[k_]))
params = _lib.split_params(prior_segments_per_second=10.)
pool = engine.StreamPool(0, T)
ctx = pool.contexts[0]
ts = [ctx.synth_trace(n_ev * ln, 7 + 13 * t_, np.array(e2), np.array(l2, dtype=np.int32), dtype=torch.float32) for t_ in range(T)]
off = np.arange(n_ev + 1, dtype=np.int64) * ln
job = lambda cx, k, t: cx.segment_batch(ts[t], off, params, synth.QUANTUM, want_stats=False)[0]
ref = None
for rep in range(3):
    for tile in (0, 50000, 56000, 100000):
        for cx in pool.contexts:
            cx.set_tiling(tile, 0)
        for _ in range(3):
            b = job(ctx, 0, 0)
        if ref is None:
            ref = b.clone()
        assert torch.equal(b, ref), "boundaries differ with tile %d" % tile
        torch.cuda.synchronize(); t0 = time.perf_counter(); s = 0.0
        for _ in range(20):
            job(ctx, 0, 0); s += ctx.seq_ms()
        torch.cuda.synchronize(); lone = (time.perf_counter() - t0) / 20 * 1e3
        tm = ctx.timings()
        pool.run(2 * T, job)
        torch.cuda.synchronize(); t0 = time.perf_counter(); pool.run(48, job); torch.cuda.synchronize()
        fl = (time.perf_counter() - t0) / 48 * 1e3
        print("tile %6d: lone %.4f ms (device %.4f), %d in flight %.4f ms per batch; tiles %d windows %d" % (tile, lone, s / 20, T, fl, tm["tiles"], tm["windows"]))
