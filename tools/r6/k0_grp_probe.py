#!/usr/bin/env python3
"""Round 6 (VERDICT r5 next #1a): do K0's instructions count towards the step?  The pool's steady state (T calls in flight,
each on its own 1e8-sample trace) with the product's K0 against K0 WITHOUT its group-record part (~37 of its ~118 vector
instructions per block, -6.8 M wave instructions per call) while the scans read the records an earlier call left --
option dbg_k0_nogrp of libporeseg_diag.so.  Interleaved pairs inside one process; mean +- spread at the end.
usage: PORESEG_LIB=pypore_amd/libporeseg_diag.so python tools/r6/k0_grp_probe.py [T] [K] [pairs]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np, torch
from pypore_amd import _lib, engine, synth
engine.apply_env_defaults()
T = int(sys.argv[1]) if len(sys.argv) > 1 else 16
K = int(sys.argv[2]) if len(sys.argv) > 2 else 100
PAIRS = int(sys.argv[3]) if len(sys.argv) > 3 else 12
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
ref = pool.run(4 * T, job)[-T:]                       # boundaries per context (job k runs on context k % T)


def measure(flag, steps):
    for cx in pool.contexts:
        cx.set_option("dbg_k0_nogrp", flag)
    pool.run(T, job)
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = pool.run(steps, job); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3, r


res = {0: [], 1: []}
res20 = {0: [], 1: []}
for p in range(PAIRS):
    for flag in ((0, 1) if p % 2 == 0 else (1, 0)):
        ms, r = measure(flag, K)
        assert all(r[k] == ref[k % T] for k in range(len(r))), "boundary counts changed"
        res[flag].append(ms)
        ms20, _ = measure(flag, 20)
        res20[flag].append(ms20)
for name, rr in (("%d steps" % K, res), ("20 steps", res20)):
    for flag in (0, 1):
        a = np.array(rr[flag])
        print("%s  K0 %s: mean %.4f  median %.4f  min %.4f  max %.4f  sd %.4f  (%s)" % (
            name, "without group records" if flag else "as shipped         ", a.mean(), np.median(a), a.min(), a.max(), a.std(),
            " ".join("%.4f" % x for x in a)))
    d = np.array(rr[0]) - np.array(rr[1])
    print("%s  paired difference shipped - without: mean %.4f ms  sd %.4f  (t = %.1f over %d pairs)" % (
        name, d.mean(), d.std(ddof=1), d.mean() / (d.std(ddof=1) / np.sqrt(len(d)) + 1e-12), len(d)))
