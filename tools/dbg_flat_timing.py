"""The 1e8-sample trace without steps, one call at a time: kernel phases (timing = 2) with the look-ahead kernel's helpers off / on.
usage: dbg_flat_timing.py [0|1]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pypore_amd import _lib, engine, synth
from pypore_amd import engine as _ps_engine
_ps_engine.apply_env_defaults()           # tools take their settings from PORESEG_* variables; the product reads none
ctx = engine.context(0)
n = 100_000_000
d = synth.dwell_table(77, n, n + 1, n + 2); ends = np.cumsum(d); lv = synth.LEVEL_COUNTS[np.arange(len(d)) % 5].astype(np.int32)
t = ctx.synth_trace(n, 77, ends, lv, dtype=torch.float32)
p = _lib.split_params(min_width=100, max_width=1000000, window_width=10000, prior_segments_per_second=10., sampling_freq=1e5)
ctx.set_option("timing", 2)
for on in ([int(sys.argv[1])] if len(sys.argv) > 1 else (0, 1)):
    ctx.set_option("lat_help", on)
    for _ in range(2):
        b, o, _ = ctx.segment_batch(t, np.array([0, n], dtype=np.int64), p, synth.QUANTUM, want_stats=False)
    print(on, b.numel(), {k: (round(v, 3) if k.endswith("_ms") else int(v)) for k, v in ctx.timings().items()})
print(b.cpu().numpy()[:12])
