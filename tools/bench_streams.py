#!/usr/bin/env python3
"""Throughput of independent batches submitted from T host threads, each with its own ps_ctx (own stream, own scratch):
the kernels of different batches overlap on the GPU (tails and single-workgroup phases of one batch are filled by the
other).  usage: bench_streams.py [threads ...]"""
import sys, os, time, threading
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pypore_amd import _lib, engine, synth
from pypore_amd import engine as _ps_engine
_ps_engine.apply_env_defaults()           # tools take their settings from PORESEG_* variables; the product reads none

n, seed, steps = 100_000_000, 2024, 40
params = _lib.split_params(min_width=100, max_width=1000000, window_width=10000, prior_segments_per_second=10., sampling_freq=1e5)
ctx0 = engine.context(0)
d = synth.dwell_table(seed, n); ends = np.cumsum(d); lv = synth.LEVEL_COUNTS[np.arange(len(d)) % 5].astype(np.int32)
trace = ctx0.synth_trace(n, seed, ends, lv, dtype=torch.float32)
ev_off = np.array([0, n], dtype=np.int64)
torch.cuda.synchronize()
ref = ctx0.segment_batch(trace, ev_off, params, synth.QUANTUM, want_stats=False)[0].cpu().numpy()
for T in [int(a) for a in sys.argv[1:]] or [1, 2, 3, 4]:
    ctxs = [ctx0] + [engine.Context(0) for _ in range(T - 1)]
    outs = [torch.empty(n // 100 + 1, dtype=torch.int32, device="cuda") for _ in range(T)]
    ok = [True] * T
    def worker(t, k):
        for _ in range(k):
            b = ctxs[t].segment_batch(trace, ev_off, params, synth.QUANTUM, want_stats=False, out=outs[t])[0]
        ok[t] = bool(np.array_equal(b.cpu().numpy(), ref))
    for t in range(T):
        worker(t, 3)
    torch.cuda.synchronize()
    th = [threading.Thread(target=worker, args=(t, steps // T)) for t in range(T)]
    t0 = time.perf_counter()
    for x in th: x.start()
    for x in th: x.join()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    k = (steps // T) * T
    print("threads %d: %.4f ms per batch, %.1f Gsamples/s, results equal %s" % (T, dt / k * 1e3, n * k / dt / 1e9, all(ok)))
