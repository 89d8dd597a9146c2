"""Where do the sporadic long steps of a sustained multi-stream run come from?  Completion time of every step of a long
run on a 4-context StreamPool; prints the steps that took more than three times the median, per context."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pypore_amd import _lib, engine, synth
from pypore_amd import engine as _ps_engine
_ps_engine.apply_env_defaults()           # tools take their settings from PORESEG_* variables; the product reads none
T = int(os.environ.get("STREAMS", "4")); n = 100_000_000; K = int(os.environ.get("K", "2000"))
pool = engine.StreamPool(0, T)
d = synth.dwell_table(1, n, 1000, 20000); lv = synth.LEVEL_COUNTS[np.arange(len(d)) % 5].astype(np.int32)
trace = pool.contexts[0].synth_trace(n, 1, np.cumsum(d), lv, dtype=torch.float32)
p = _lib.split_params(prior_segments_per_second=10.); off = np.array([0, n], dtype=np.int64)
outs = [torch.empty(n // 100 + 1, dtype=torch.int32, device="cuda") for _ in range(T)]
def job(ctx, k, t):
    t0 = time.perf_counter()
    ctx.segment_batch(trace, off, p, synth.QUANTUM, want_stats=False, out=outs[t])
    t1 = time.perf_counter()
    return (t, t0, t1, ctx.seq_ms())
pool.run(8, job); torch.cuda.synchronize()
import gc
if os.environ.get("NOGC"): gc.disable()
t00 = time.perf_counter(); res = pool.run(K, job); dt = time.perf_counter() - t00
dur = np.array([r[2] - r[1] for r in res]) * 1e3; seq = np.array([r[3] for r in res])
med = np.median(dur)
print("%d steps on %d contexts: %.3f ms per step; call duration median %.3f ms, device sequence median %.3f ms" % (K, T, dt / K * 1e3, med, np.median(seq)))
slow = [i for i in range(K) if dur[i] > 3 * med]
print("%d calls longer than 3 x median:" % len(slow))
for i in slow[:40]:
    print("  step %5d ctx %d  started %.1f ms into the run, took %.2f ms on the host, %.2f ms device sequence" % (i, res[i][0], (res[i][1] - t00) * 1e3, dur[i], seq[i]))
