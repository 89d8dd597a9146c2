#!/usr/bin/env python3
"""Round 6: what do the scan kernels lose when waves that only HOLD registers sit on the SIMDs (tools/probes/squatter.hip)?  The
pool's steady state of the scan kernels alone (diagnostic library, dbg_phase 1: K0 skipped, the previous call's digest is still
there), sixteen calls in flight, beside squatters of W waves per SIMD x R registers each.  If K0's cost to the scans is the
register file it occupies while its bytes are in flight, 3 x 128 registers of squatters should cost what three admitted K0s cost,
and ONE wave of 192 registers per SIMD -- the same 24 KB in flight in half the registers -- should cost much less.
usage: PORESEG_LIB=pypore_amd/libporeseg_diag.so python tools/r6/residency_probe.py [T] [K]"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "20")
import numpy as np, torch
from pypore_amd import _lib, engine, synth
engine.apply_env_defaults()
T = int(sys.argv[1]) if len(sys.argv) > 1 else 16
K = int(sys.argv[2]) if len(sys.argv) > 2 else 100
sq = ctypes.CDLL(os.path.join(ROOT, "tools", "probes", "libsquatter.so"))
sq.squat_alive.restype = ctypes.c_longlong
n = 100_000_000
params = _lib.split_params(min_width=100, max_width=1000000, window_width=10000, prior_segments_per_second=10., sampling_freq=1e5)
pool = engine.StreamPool(0, T)
ctx0 = pool.contexts[0]
traces = []
for t in range(T):
    sd = 2024 + 1000 * t
    d = synth.dwell_table(sd, n); ends = np.cumsum(d); lv = synth.LEVEL_COUNTS[np.arange(len(d)) % 5].astype(np.int32)
    traces.append(ctx0.synth_trace(n, sd, ends, lv, dtype=torch.float32))
ev_off = np.array([0, n], dtype=np.int64)
outs = [torch.empty(n // 100 + 1, dtype=torch.int32, device="cuda") for _ in range(T)]
job = lambda cx, k, t: cx.segment_batch(traces[t], ev_off, params, synth.QUANTUM, want_stats=False, out=outs[t])[0].numel()
import gc; gc.collect(); gc.freeze()
ref = pool.run(4 * T, job)[-T:]
for cx in pool.contexts:
    cx.set_option("dbg_phase", 1)                        # the scan kernels alone from here on
pool.run(2 * T, job)


def run(waves, regs):
    torch.cuda.synchronize()
    grid = sq.squat_start(waves, regs) if waves else 0
    assert grid >= 0, grid
    t0 = time.time()
    while waves and sq.squat_alive() < grid and time.time() - t0 < 2.0:
        time.sleep(0.01)
    alive = sq.squat_alive() if waves else 0
    pool.run(T, job)
    t1 = time.perf_counter(); r = pool.run(K, job); dt = (time.perf_counter() - t1) / K * 1e3
    assert all(r[k] == ref[k % T] for k in range(len(r))), "boundary counts changed"
    if waves:
        t2 = time.time()
        sq.squat_stop()
        if time.time() - t2 > 0.5:
            print("(the squatters took %.1f s to leave: they did not see the flag)" % (time.time() - t2), flush=True)
    return dt, alive, grid


cases = [(0, 0), (1, 128), (2, 128), (3, 128), (1, 192), (1, 256), (2, 192), (3, 64), (3, 96), (0, 0)]
for rep in range(2):
    for w, r in cases:
        dt, alive, grid = run(w, r)
        print("scans alone, %d in flight, squatters %d waves/SIMD x %3d registers (%4d of %4d started): %.4f ms per step" % (T, w, r, alive, grid, dt), flush=True)
