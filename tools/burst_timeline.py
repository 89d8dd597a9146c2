#!/usr/bin/env python3
"""Round 5: the driver's burst -- 20 steps over 16 contexts -- on the host's clock: when every call starts and ends
(relative to the start of the run), to see the fill and the tail of the pipeline.  usage: burst_timeline.py [T] [K]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np, torch
from pypore_amd import _lib, engine, synth
from pypore_amd import engine as _ps_engine
_ps_engine.apply_env_defaults()           # tools take their settings from PORESEG_* variables; the product reads none
T = int(sys.argv[1]) if len(sys.argv) > 1 else 16
K = int(sys.argv[2]) if len(sys.argv) > 2 else 20
n = 100_000_000
params = _lib.split_params(min_width=100, max_width=1000000, window_width=10000, prior_segments_per_second=10., sampling_freq=1e5)
pool = engine.StreamPool(0, T)
ctx0 = pool.contexts[0]
traces = []
for t in range(T):
    sd = 2024 + 1000 * t
    d = synth.dwell_table(sd, n); ends = np.cumsum(d); lv = synth.LEVEL_COUNTS[np.arange(len(d)) % 5].astype(np.int32)
    traces.append(ctx0.synth_trace(n, sd, ends, lv, dtype=torch.float32))
ev_off = np.array([0, n], dtype=np.int64)
outs = [torch.empty(n // 100 + 1, dtype=torch.int32, device="cuda") for _ in range(T)]
marks = []
def job(cx, k, t):
    a = time.perf_counter()
    cx.segment_batch(traces[t], ev_off, params, synth.QUANTUM, want_stats=False, out=outs[t])
    marks.append((k, t, a, time.perf_counter(), cx.seq_ms()))
import gc; gc.collect(); gc.freeze()
pool.run(4 * T, job)
for rep in range(6):
    marks.clear()
    torch.cuda.synchronize(); t0 = time.perf_counter(); pool.run(K, job); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("run %d: %.3f ms total = %.4f ms/step" % (rep, dt * 1e3, dt / K * 1e3))
    print("   last start %.3f ms" % max((a - t0) * 1e3 for k, t, a, b, seq in marks if k < T))
    for k, t, a, b, seq in (sorted(marks) if os.environ.get("VERBOSE") else []):
        print("   step %2d ctx %2d: start %.3f end %.3f ms (%.3f), device sequence %.3f" % (k, t, (a - t0) * 1e3, (b - t0) * 1e3, (b - a) * 1e3, seq))
