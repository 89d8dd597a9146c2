"""Timing of ps_filter_bessel (Event.filter on the device) on a 1e8-sample trace resident in HBM (diagnostic)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pypore_amd import engine, synth
from pypore_amd import engine as _ps_engine
_ps_engine.apply_env_defaults()           # tools take their settings from PORESEG_* variables; the product reads none
ctx = engine.context(0)
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
d = synth.dwell_table(2024, n)
ends = np.cumsum(d)
lv = np.array([synth.LEVEL_COUNTS[i % len(synth.LEVEL_COUNTS)] for i in range(len(d))], dtype=np.int32)
for dt, nbytes in ((torch.float32, 4), (torch.int16, 2)):
    t = ctx.synth_trace(n, 2024, ends, lv, dtype=dt)
    for _ in range(2):
        y = ctx.filter_bessel(t, synth.QUANTUM)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        y = ctx.filter_bessel(t, synth.QUANTUM)
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 5 * 1e3
    alg = (nbytes + 8) * n
    print("filter %s: %.3f ms per %.0e samples = %.0f Msamples/s; algorithmic %d B/sample -> %.0f GB/s = %.1f %% of 8 TB/s"
          % (str(dt), ms, n, n / ms / 1e3, nbytes + 8, alg / ms / 1e6, alg / ms / 1e6 / 80))
    del y
