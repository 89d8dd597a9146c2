#!/usr/bin/env python3
"""Round 5: where the filtered half of Experiment.parse spends its time on the 1e8-sample file: the steps of
FastStatSplit.parse_filtered_batch one by one (filter, re-quantisation, segmentation on the 64-bit digest), with the default
workflow's parameters (cutoff_freq=2000 in the segmenter) and without cutoff_freq."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pypore_amd import _lib, engine, synth, pipeline
from pypore_amd import engine as _ps_engine
_ps_engine.apply_env_defaults()           # tools take their settings from PORESEG_* variables; the product reads none
n = 100_000_000
ctx = engine.context(0)
ends, lv, _ = synth.file_trace_table(n, 7)
trace = ctx.synth_trace(n, 7, ends, lv, dtype=torch.int16)
st, ln = ctx.detect_events(trace, synth.QUANTUM, threshold=90.0)
print("events", len(st), "samples in events", int(ln.sum()))
def sync(): torch.cuda.synchronize()
for rep in range(3):
    sync(); t0 = time.perf_counter()
    ys = [ctx.filter_bessel(trace[a:a + l], synth.QUANTUM, cutoff=2000., sampling_freq=1e5, order=1) for a, l in zip(st, ln)]
    sync(); t1 = time.perf_counter()
    zs = [ctx.requantise(y) for y in ys]
    sync(); t2 = time.perf_counter()
    by_step = {}
    for i, (z, c, s) in enumerate(zs):
        by_step.setdefault(s, []).append(i)
    out = {}
    for label, kw in (("cutoff_freq=2000", dict(cutoff_freq=2000.)), ("no cutoff_freq", dict())):
        params = _lib.split_params(min_width=100, max_width=1000000, window_width=10000, prior_segments_per_second=10., sampling_freq=1e5, **kw)
        sync(); t3 = time.perf_counter(); nb = 0; dev = 0.0; wins = 0
        for s, idx in by_step.items():
            lens = np.array([zs[i][0].numel() for i in idx], dtype=np.int64)
            off = np.concatenate(([0], np.cumsum(lens)))
            samples = torch.cat([zs[i][0] for i in idx]) if len(idx) > 1 else zs[idx[0]][0]
            b, boff, _ = ctx.segment_batch(samples, off, params, s, want_stats=False)
            nb += b.numel(); dev += ctx.seq_ms(); wins += ctx.timings()["windows"]
        sync(); t4 = time.perf_counter()
        out[label] = (t4 - t3, nb, dev, wins)
    print("filter %.1f ms, requantise %.1f ms (%d grid steps); " % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, len(by_step)) +
          "; ".join("segmentation %s: %.1f ms wall, %.1f ms device, %d boundaries, %d windows" % (k, v[0] * 1e3, v[2], v[1], v[3]) for k, v in out.items()))
