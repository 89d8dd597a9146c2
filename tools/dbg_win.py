import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pypore_amd import _lib, engine, synth
import oracle
ctx = engine.context(0)
n = 100_000_000; seed = 2024
d = synth.dwell_table(seed, n); ends = np.cumsum(d)
def counts(s, e):
    seg = np.searchsorted(ends, np.arange(s, e, dtype=np.int64), side="right")
    return synth.LEVEL_COUNTS[seg % 5] + synth.noise_counts(seed, s, e - s)
params = _lib.split_params(min_width=100, max_width=1000000, window_width=10000, prior_segments_per_second=10.)
for shift in range(0, 8):
    s0 = 80240605 - shift
    c = counts(s0, 80247206)
    x32 = torch.from_numpy(synth.counts_to_pa(c, np.float32)).cuda()
    # event = [shift, end) of this buffer so that the window starts at the same samples with different alignment
    for bs in (1, 0):
        ctx.set_option("scan_bs", bs)
        b, boff, _ = ctx.segment_events(x32, np.array([shift]), np.array([len(c) - shift]), params, synth.QUANTUM)
        print("shift", shift, "bs", bs, (b.cpu().numpy()[:3] + s0 + shift).tolist(), ctx.timings()["exact_rescans"], ctx.timings()["full_exact_scans"])
ref = oracle.parse(synth.counts_to_pa(counts(80240605, 80247206), np.float64), prior_segments_per_second=10.)
print("oracle", (ref[:3] + 80240605).tolist())
