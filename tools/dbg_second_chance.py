"""Second chance for seams that gave up (seg_device.hpp: EXT_MAX): small dense traces with a lowered bridge budget against the
oracle, the dense 1e8 trace and the events whose steps fall on tile starts, with the second chance on and off."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import oracle
from pypore_amd import _lib, engine, synth
from pypore_amd import engine as _ps_engine
_ps_engine.apply_env_defaults()           # tools take their settings from PORESEG_* variables; the product reads none
ctx = engine.context(0)
kw = dict(min_width=100, max_width=1000000, window_width=10000, prior_segments_per_second=10., sampling_freq=1e5)
p = _lib.split_params(**kw)

def trace(n, lo, hi, seed=77):
    d = synth.dwell_table(seed, n, lo, hi); ends = np.cumsum(d); lv = synth.LEVEL_COUNTS[np.arange(len(d)) % 5].astype(np.int32)
    return ctx.synth_trace(n, seed, ends, lv, dtype=torch.float32)

def run(t, off, reps=1):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        b, o, _ = ctx.segment_batch(t, off, p, synth.QUANTUM, want_stats=False)
    torch.cuda.synchronize()
    return b.cpu().numpy(), o, (time.perf_counter() - t0) / reps * 1e3, int(ctx.timings()["repairs"])

n = 3_000_000
for lo, hi in ((100, 400), (300, 3000), (1000, 20000)):
    t = trace(n, lo, hi)
    ref = oracle.parse(t.cpu().numpy().astype(np.float64), **kw)
    for budget in (256, 16, 8, 4, 2, 1):
        ctx.set_option("bridge_budget", budget)
        b, _, ms, rep = run(t, np.array([0, n], dtype=np.int64))
        print("n %d dwell %d-%d budget %3d: %s  %.3f ms  repairs %d" % (n, lo, hi, budget, "ok" if np.array_equal(b, ref) else "DIFFERENT (%d vs %d)" % (len(b), len(ref)), ms, rep))
ctx.set_option("bridge_budget", 256)
t = trace(100_000_000, 100, 400)
for on in (1, 0, 1):
    ctx.set_option("bridge_ext", on)
    b, _, ms, rep = run(t, np.array([0, 100_000_000], dtype=np.int64), 2)
    print("1e8 dwell 100-400, second chance %d: %d boundaries %.3f ms repairs %d" % (on, len(b), ms, rep))
    if on == 0: ref = b
    elif 'ref' in dir() and on == 1 and ref is not None and len(ref) == len(b): print("   equal to the host stitch:", np.array_equal(b, ref))
del t
n_ev, ln = 128, 400000
ends, lv = [], []
for e in range(n_ev):
    for k in range(5):
        ends.append(e * ln + (k + 1) * (ln // 5)); lv.append(int(synth.LEVEL_COUNTS[k]))
t = ctx.synth_trace(n_ev * ln, 7, np.array(ends), np.array(lv, dtype=np.int32), dtype=torch.float32)
off = np.arange(n_ev + 1, dtype=np.int64) * ln
res = {}
for on in (1, 0):
    ctx.set_option("bridge_ext", on)
    b, o, ms, rep = run(t, off, 2)
    res[on] = b
    print("128 x 400000, steps on tile starts, second chance %d: %d boundaries %.3f ms repairs %d" % (on, len(b), ms, rep))
print("   equal:", np.array_equal(res[0], res[1]))
ctx.set_option("bridge_ext", 1)
