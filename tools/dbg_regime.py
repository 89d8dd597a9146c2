"""One call on a random-dwell trace with PORESEG_DEBUG (which seams gave up).  usage: dbg_regime.py n dwell_lo dwell_hi"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pypore_amd import _lib, engine, synth
from pypore_amd import engine as _ps_engine
_ps_engine.apply_env_defaults()           # tools take their settings from PORESEG_* variables; the product reads none
ctx = engine.context(0)
n, lo, hi = int(float(sys.argv[1])), int(float(sys.argv[2])), int(float(sys.argv[3]))
d = synth.dwell_table(77, n, lo, hi); ends = np.cumsum(d); lv = synth.LEVEL_COUNTS[np.arange(len(d)) % 5].astype(np.int32)
t = ctx.synth_trace(n, 77, ends, lv, dtype=torch.float32)
p = _lib.split_params(min_width=100, max_width=1000000, window_width=10000, prior_segments_per_second=10., sampling_freq=1e5)
b, o, _ = ctx.segment_batch(t, np.array([0, n], dtype=np.int64), p, synth.QUANTUM, want_stats=False)
print(b.numel(), {k: (round(v, 3) if k.endswith("_ms") else int(v)) for k, v in ctx.timings().items()})
