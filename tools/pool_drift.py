"""Does the step time drift over a long run?  Chunks of 20 steps on a 4-context StreamPool, one after the other in one
process, ms per step of every chunk (and the GPU clock rocm-smi reports in between)."""
import os, sys, time, subprocess
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pypore_amd import _lib, engine, synth
from pypore_amd import engine as _ps_engine
_ps_engine.apply_env_defaults()           # tools take their settings from PORESEG_* variables; the product reads none
T = int(os.environ.get("STREAMS", "4")); n = 100_000_000
pool = engine.StreamPool(0, T)
d = synth.dwell_table(1, n, 1000, 20000); lv = synth.LEVEL_COUNTS[np.arange(len(d)) % 5].astype(np.int32)
traces = [ctx.synth_trace(n, 1, np.cumsum(d), lv, dtype=torch.float32) for ctx in pool.contexts]
p = _lib.split_params(prior_segments_per_second=10.); off = np.array([0, n], dtype=np.int64)
def job(ctx, k, t):
    return ctx.segment_batch(traces[t], off, p, synth.QUANTUM, want_stats=False)[0].numel()
pool.run(8, job); torch.cuda.synchronize()
out = []
sleep = float(os.environ.get("SLEEP", "0"))
for c in range(int(os.environ.get("CHUNKS", "30"))):
    K = int(os.environ.get("K", "20"))
    t0 = time.perf_counter(); pool.run(K, job); torch.cuda.synchronize(); out.append((time.perf_counter() - t0) / K * 1e3)
    if sleep: time.sleep(sleep)
print("streams %d, sleep %.2f s between chunks: ms per step of consecutive chunks of K steps:" % (T, sleep), " ".join("%.3f" % v for v in out))
