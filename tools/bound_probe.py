#!/usr/bin/env python3
"""Round 5: what bounds a step with T calls in flight?  The pool's step time for the whole call, for the scan kernels alone
(PORESEG_DBG_PHASE=1: K0 skipped, the previous call's digest is still there) and for K0 alone (PORESEG_DBG_PHASE=2), each on
T contexts with their own trace.  usage: bound_probe.py [T] [steps]   (run once per PORESEG_DBG_PHASE value)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
import numpy as np, torch
from pypore_amd import _lib, engine, synth
from pypore_amd import engine as _ps_engine
_ps_engine.apply_env_defaults()           # tools take their settings from PORESEG_* variables; the product reads none
T = int(sys.argv[1]) if len(sys.argv) > 1 else 16
K = int(sys.argv[2]) if len(sys.argv) > 2 else 160
n = 100_000_000
params = _lib.split_params(min_width=100, max_width=1000000, window_width=10000, prior_segments_per_second=10., sampling_freq=1e5)
pool = engine.StreamPool(0, T)
ctx0 = pool.contexts[0]
traces = []
for t in range(T):
    sd = 2024 + 1000 * t
    d = synth.dwell_table(sd, n, int(float(os.environ.get("DWELL_LO", 1000))), int(float(os.environ.get("DWELL_HI", 20000)))); ends = np.cumsum(d); lv = synth.LEVEL_COUNTS[np.arange(len(d)) % 5].astype(np.int32)
    traces.append(ctx0.synth_trace(n, sd, ends, lv, dtype=torch.float32))
ev_off = np.array([0, n], dtype=np.int64)
outs = [torch.empty(n // 100 + 1, dtype=torch.int32, device="cuda") for _ in range(T)]
job = lambda cx, k, t: cx.segment_batch(traces[t], ev_off, params, synth.QUANTUM, want_stats=False, out=outs[t])[0].numel()
import gc; gc.collect(); gc.freeze()
pool.run(4 * T, job)
res = []
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = pool.run(K, job); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    res.append(dt / K * 1e3)
print("phase %s T=%d K=%d k0_waves=%s shared=%s: %s ms/step, boundaries %s" % (os.environ.get("PORESEG_DBG_PHASE", "0"), T, K,
      os.environ.get("PORESEG_K0_WAVES", "-"), os.environ.get("PORESEG_POOL_SHARED", "-"), " ".join("%.4f" % x for x in res), r[-1]))
