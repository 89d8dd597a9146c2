"""Helpers of the look-ahead kernel (seg_device.hpp: LAT_D) on traces with long stretches without splits: the same call with the
helpers off and on (equal boundaries, ms per call), small traces against the oracle."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import oracle
from pypore_amd import _lib, engine, synth
from pypore_amd import engine as _ps_engine
_ps_engine.apply_env_defaults()           # tools take their settings from PORESEG_* variables; the product reads none
ctx = engine.context(0)
base = dict(min_width=100, max_width=1000000, window_width=10000, prior_segments_per_second=10., sampling_freq=1e5)

def trace(n, lo, hi, seed=77):
    d = synth.dwell_table(seed, n, lo, hi); ends = np.cumsum(d); lv = synth.LEVEL_COUNTS[np.arange(len(d)) % 5].astype(np.int32)
    return ctx.synth_trace(n, seed, ends, lv, dtype=torch.float32)

def run(t, off, p, reps=3):
    b, o, _ = ctx.segment_batch(t, off, p, synth.QUANTUM, want_stats=False)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        b, o, _ = ctx.segment_batch(t, off, p, synth.QUANTUM, want_stats=False)
    torch.cuda.synchronize()
    tm = ctx.timings()
    return b.cpu().numpy(), np.array(o), (time.perf_counter() - t0) / reps * 1e3, int(tm["repairs"]), int(tm["windows"])

small = int(sys.argv[1]) if len(sys.argv) > 1 else 0
n = 4_000_000
for name, lo, hi, extra in (("dwell 1e5-1e6", 100000, 1000000, {}), ("dwell 3e5-3e6 maxw 250000", 300000, 3000000, dict(max_width=250000)),
                            ("no step", n + 1, n + 2, {}), ("no step, maxw 123456", n + 1, n + 2, dict(max_width=123456)),
                            ("no step W=4000 maxw 50000", n + 1, n + 2, dict(window_width=4000, max_width=50000)),
                            ("dwell 2e4-2e5", 20000, 200000, {}), ("dwell 1000-20000", 1000, 20000, {})):
    kw = dict(base); kw.update(extra)
    t = trace(n, lo, hi)
    ref = oracle.parse(t.cpu().numpy().astype(np.float64), **kw)
    p = _lib.split_params(**kw)
    for on in (0, 1):
        ctx.set_option("lat_help", on)
        b, _, ms, rep, win = run(t, np.array([0, n], dtype=np.int64), p)
        print("n %d %-28s helpers %d: %s %.3f ms repairs %d windows %d" % (n, name, on, "ok" if np.array_equal(b, ref) else "DIFFERENT (%d vs %d)" % (len(b), len(ref)), ms, rep, win))
if small: sys.exit(0)
n = 100_000_000
for name, lo, hi, extra in (("dwell 1e5-1e6", 100000, 1000000, {}), ("dwell 1e6-1e7", 1000000, 10000000, {}), ("dwell 1e7-5e7", 10000000, 50000000, {}),
                            ("no step", n + 1, n + 2, {}), ("no step maxw 1e9", n + 1, n + 2, dict(max_width=1000000000)),
                            ("dwell 1000-20000", 1000, 20000, {})):
    kw = dict(base); kw.update(extra)
    t = trace(n, lo, hi)
    p = _lib.split_params(**kw)
    res = {}
    for on in (0, 1):
        ctx.set_option("lat_help", on)
        res[on] = run(t, np.array([0, n], dtype=np.int64), p)
        print("n %d %-22s helpers %d: %d boundaries %.3f ms repairs %d windows %d" % (n, name, on, len(res[on][0]), res[on][2], res[on][3], res[on][4]))
    print("   equal:", np.array_equal(res[0][0], res[1][0]))
    del t
n_ev, ln = 8, 6_400_000
ends, lv = [], []
for e in range(n_ev):
    for k in range(5):
        ends.append(e * ln + int((0.17, 0.41, 0.58, 0.83, 1.0)[k] * ln)); lv.append(int(synth.LEVEL_COUNTS[k]))
t = ctx.synth_trace(n_ev * ln, 7, np.array(ends), np.array(lv, dtype=np.int32), dtype=torch.float32)
off = np.arange(n_ev + 1, dtype=np.int64) * ln
res = {}
for on in (0, 1):
    ctx.set_option("lat_help", on)
    res[on] = run(t, off, _lib.split_params(**base))
    print("8 x 6.4e6 helpers %d: %d boundaries %.3f ms repairs %d windows %d" % (on, len(res[on][0]), res[on][2], res[on][3], res[on][4]))
print("   equal:", np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1]))
ctx.set_option("lat_help", 1)
