"""Where does a bad SHORT run come from?  R repetitions of K steps on a T-context StreamPool (the driver's K = 20), every
repetition bracketed by a device synchronisation like the bench; for the slowest repetitions: per call the context, host
start / end and the device sequence time."""
import os, sys, time, gc
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pypore_amd import _lib, engine, synth
from pypore_amd import engine as _ps_engine
_ps_engine.apply_env_defaults()           # tools take their settings from PORESEG_* variables; the product reads none
T = int(os.environ.get("STREAMS", "12")); n = 100_000_000; K = int(os.environ.get("K", "20")); R = int(os.environ.get("R", "60"))
pool = engine.StreamPool(0, T)
traces = []
for t in range(T):
    d = synth.dwell_table(2024 + 1000 * t, n); lv = synth.LEVEL_COUNTS[np.arange(len(d)) % 5].astype(np.int32)
    traces.append(pool.contexts[0].synth_trace(n, 2024 + 1000 * t, np.cumsum(d), lv, dtype=torch.float32))
p = _lib.split_params(prior_segments_per_second=10.); off = np.array([0, n], dtype=np.int64)
outs = [torch.empty(n // 100 + 1, dtype=torch.int32, device="cuda") for _ in range(T)]
def job(ctx, k, t):
    t0 = time.perf_counter()
    ctx.segment_batch(traces[t], off, p, synth.QUANTUM, want_stats=False, out=outs[t])
    t1 = time.perf_counter()
    return (t, t0, t1, ctx.seq_ms())
pool.run(4 * T, job); torch.cuda.synchronize()
gc.collect(); gc.freeze()
runs = []
for r in range(R):
    torch.cuda.synchronize()
    t00 = time.perf_counter(); res = pool.run(K, job); torch.cuda.synchronize(); dt = time.perf_counter() - t00
    runs.append((dt / K * 1e3, t00, res))
ms = np.array([x[0] for x in runs])
print("%d runs of %d steps on %d contexts: ms per step median %.4f, mean %.4f, max %.4f, runs above 1.3 x median: %d" % (R, K, T, np.median(ms), ms.mean(), ms.max(), int((ms > 1.3 * np.median(ms)).sum())))
print("all runs:", " ".join("%.3f" % x for x in ms))
for i in np.argsort(ms)[::-1][:2]:
    v, t00, res = runs[i]
    print("run %d: %.4f ms per step" % (i, v))
    for (t, a, b, s) in sorted(res, key=lambda x: x[1]):
        print("   ctx %2d  host start %7.3f ms  end %7.3f ms  (%.3f ms)  device sequence %.3f ms" % (t, (a - t00) * 1e3, (b - t00) * 1e3, (b - a) * 1e3, s))
