"""BASELINE config 2 timing: batch of 1024 events x 50k samples, SpeedyStatSplit on one GPU (diagnostic)."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pypore_amd import _lib, engine, synth
from pypore_amd import engine as _ps_engine
_ps_engine.apply_env_defaults()           # tools take their settings from PORESEG_* variables; the product reads none
ctx = engine.context(0)
n_ev, ln = 1024, 50000
ends = []; lv = []
for e in range(n_ev):
    for k in range(5):
        ends.append(e * ln + (k + 1) * 10000); lv.append(int(synth.LEVEL_COUNTS[k]))
# one generator call with a per-sample noise stream (seed fixed): shape of config 2, not its per-event seeds
t = ctx.synth_trace(n_ev * ln, 7, np.array(ends), np.array(lv, dtype=np.int32), dtype=torch.float32)
off = np.arange(n_ev + 1, dtype=np.int64) * ln
params = _lib.split_params(prior_segments_per_second=10.)
for _ in range(3):
    b, boff, _ = ctx.segment_batch(t, off, params, synth.QUANTUM, want_stats=False)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10):
    b, boff, _ = ctx.segment_batch(t, off, params, synth.QUANTUM, want_stats=False)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
print("config2: %d events x %d: %.3f ms/step, %.1f Msamples/s, %d boundaries" % (n_ev, ln, dt * 1e3, n_ev * ln / dt / 1e6, b.numel()))
ctx.set_option("timing", 2)
acc = {}
for _ in range(10):
    ctx.segment_batch(t, off, params, synth.QUANTUM, want_stats=False)
    for k, v in ctx.timings().items():
        acc[k] = acc.get(k, 0) + v / 10
print({k: round(v, 4) for k, v in acc.items() if k.endswith("_ms")}, {k: int(v) for k, v in acc.items() if not k.endswith("_ms")})
