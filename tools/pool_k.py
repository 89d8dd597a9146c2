import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from pypore_amd import _lib, engine, synth
from pypore_amd import engine as _ps_engine
_ps_engine.apply_env_defaults()           # tools take their settings from PORESEG_* variables; the product reads none
n, seed = 100_000_000, 2024
params = _lib.split_params(min_width=100, max_width=1000000, window_width=10000, prior_segments_per_second=10., sampling_freq=1e5)
pool = engine.StreamPool(0, 4)
ctx0 = pool.contexts[0]
d = synth.dwell_table(seed, n); ends = np.cumsum(d); lv = synth.LEVEL_COUNTS[np.arange(len(d)) % 5].astype(np.int32)
trace = ctx0.synth_trace(n, seed, ends, lv, dtype=torch.float32)
ev_off = np.array([0, n], dtype=np.int64)
outs = [torch.empty(n // 100 + 1, dtype=torch.int32, device="cuda") for _ in range(4)]
job = lambda cx, k, t: cx.segment_batch(trace, ev_off, params, synth.QUANTUM, want_stats=False, out=outs[t])[0]
pool.run(8, job)
for K in (20, 20, 20, 40, 100, 20, 400, 20):
    torch.cuda.synchronize(); t0 = time.perf_counter(); pool.run(K, job); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("K=%d: %.3f ms total, %.4f ms/step" % (K, dt * 1e3, dt / K * 1e3))
