#!/usr/bin/env python3
"""bound_probe.py for BASELINE config 2 (1 024 events x 50 000 samples per call): the pool's step time with T batches in
flight for the whole call, the scan kernels alone (PORESEG_DBG_PHASE=1) and K0 alone (=2), and the work counters of one call.
usage: bound_probe_config2.py [T] [steps] [events] [samples per event]   (run once per PORESEG_DBG_PHASE value)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
import numpy as np, torch
from pypore_amd import _lib, engine, synth
from pypore_amd import engine as _ps_engine
_ps_engine.apply_env_defaults()           # tools take their settings from PORESEG_* variables; the product reads none
T = int(sys.argv[1]) if len(sys.argv) > 1 else 16
K = int(sys.argv[2]) if len(sys.argv) > 2 else 160
n_ev = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
ln = int(sys.argv[4]) if len(sys.argv) > 4 else 50000
params = _lib.split_params(min_width=100, max_width=1000000, window_width=10000, prior_segments_per_second=10., sampling_freq=1e5)
pool = engine.StreamPool(0, T)
ctx0 = pool.contexts[0]
ends, lv = [], []
for e in range(n_ev):
    for k in range(5):
        # (config 2's own shape for 50 000 samples; otherwise uneven dwells, so that no step falls on a tile start)
        ends.append(e * ln + ((k + 1) * (ln // 5) if ln == 50000 else int((0.17, 0.41, 0.58, 0.83, 1.0)[k] * ln))); lv.append(int(synth.LEVEL_COUNTS[k]))
ends[-1] = n_ev * ln
traces = [ctx0.synth_trace(n_ev * ln, 7 + 13 * t, np.array(ends), np.array(lv, dtype=np.int32), dtype=torch.float32) for t in range(T)]
off = np.arange(n_ev + 1, dtype=np.int64) * ln
outs = [torch.empty(n_ev * ln // 100 + n_ev, dtype=torch.int32, device="cuda") for _ in range(T)]
job = lambda cx, k, t: cx.segment_batch(traces[t], off, params, synth.QUANTUM, want_stats=False, out=outs[t])[0].numel()
import gc; gc.collect(); gc.freeze()
pool.run(4 * T, job)
res = []
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = pool.run(K, job); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    res.append(dt / K * 1e3)
tm = pool.contexts[0].timings()
print("config2 %d x %d phase %s T=%d K=%d: %s ms/step, boundaries %s; %s" % (n_ev, ln, os.environ.get("PORESEG_DBG_PHASE", "0"), T, K,
      " ".join("%.4f" % x for x in res), r[-1], {k: int(v) for k, v in tm.items() if not k.endswith("_ms")}))
