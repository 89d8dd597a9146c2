import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pypore_amd import _lib, engine, synth
ctx = engine.context(0)
n = 100_000_000; seed = 2024
d = synth.dwell_table(seed, n); ends = np.cumsum(d); lv = synth.LEVEL_COUNTS[np.arange(len(d)) % 5].astype(np.int32)
t = ctx.synth_trace(n, seed, ends, lv, dtype=torch.float32)
params = _lib.split_params(min_width=100, max_width=1000000, window_width=10000, prior_segments_per_second=10.)
ctx.set_option("scan_bs", 0)
b0, _, _ = ctx.segment_batch(t, np.array([0, n]), params, synth.QUANTUM, want_stats=False)
b0 = b0.cpu().numpy()
ctx.set_option("scan_bs", 1)
b1, _, _ = ctx.segment_batch(t, np.array([0, n]), params, synth.QUANTUM, want_stats=False)
b1 = b1.cpu().numpy()
print(len(b0), len(b1), np.array_equal(b0, b1))
if not np.array_equal(b0, b1):
    i = np.nonzero(b0[:min(len(b0), len(b1))] != b1[:min(len(b0), len(b1))])[0]
    print("first diffs at", i[:5], b0[i[:5]], b1[i[:5]], "prev", b0[i[0]-2:i[0]+3], b1[i[0]-2:i[0]+3])
ctx.set_option("mode", 2)
try:
    ctx.segment_batch(t, np.array([0, n]), params, synth.QUANTUM, want_stats=False)
    print("verify ok")
except Exception as e:
    print("verify:", e)
