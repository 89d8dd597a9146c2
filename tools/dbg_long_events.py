import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from pypore_amd import _lib, engine, synth
from pypore_amd import engine as _ps_engine
_ps_engine.apply_env_defaults()           # tools take their settings from PORESEG_* variables; the product reads none
ctx = engine.context(0)
n_ev, ln = int(sys.argv[1]), int(sys.argv[2])
params = _lib.split_params(min_width=100, max_width=1000000, window_width=10000, prior_segments_per_second=10., sampling_freq=1e5)
ends, lv = [], []
for e in range(n_ev):
    for k in range(5):
        # (config 2's own shape for 50 000 samples; otherwise uneven dwells, so that no step falls on a tile start)
        ends.append(e * ln + ((k + 1) * (ln // 5) if ln == 50000 else int((0.17, 0.41, 0.58, 0.83, 1.0)[k] * ln))); lv.append(int(synth.LEVEL_COUNTS[k]))
ends[-1] = n_ev * ln
t = ctx.synth_trace(n_ev * ln, 7, np.array(ends), np.array(lv, dtype=np.int32), dtype=torch.float32)
off = np.arange(n_ev + 1, dtype=np.int64) * ln
b, o, _ = ctx.segment_batch(t, off, params, synth.QUANTUM, want_stats=False)
print(b.numel(), {k: int(v) for k, v in ctx.timings().items() if not k.endswith("_ms")})
