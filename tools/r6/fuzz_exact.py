#!/usr/bin/env python3
"""Round 6: the exact route (ps_segment_exact_f64) against the CPU oracle -- the restatement of cparsers.pyx, itself equal to the
compiled reference on every golden -- on random float64 traces that lie on NO grid: random levels, noise, lengths, parameters,
1-4 events per call, events at odd offsets.  Bit-exact boundaries expected (the two sides run the same additions in the same
order; what is left is the logarithm's last bit, which decides nothing on continuous data).  usage: fuzz_exact.py [seeds] [base]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import oracle
from pypore_amd import _lib, engine
engine.apply_env_defaults()
ctx = engine.context(0)
n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 100
base = int(sys.argv[2]) if len(sys.argv) > 2 else 0
bad = 0
nb = 0
t0 = time.time()
for seed in range(n_seeds):
    rng = np.random.RandomState(77_000 + base + seed)
    mw = int(rng.choice([2, 8, 20, 100, 250]))
    W = int(max(2 * mw + 2, rng.choice([400, 1000, 4000, 10000, 25000])))
    maxw = int(rng.choice([W, 3 * W, 50000, 1000000]))
    params = dict(min_width=mw, max_width=max(maxw, mw), window_width=W, prior_segments_per_second=float(rng.choice([1., 10., 100.])),
                  sampling_freq=1e5)
    if rng.rand() < 0.3:
        params["cutoff_freq"] = float(rng.choice([500., 2000., 10000.]))
    n_ev = int(rng.choice([1, 1, 2, 4]))
    evs, starts, pos = [], [], int(rng.randint(0, 9))
    for e in range(n_ev):
        n = int(rng.randint(500, 200000 if n_ev == 1 else 40000))
        lo = int(rng.randint(50, 3000)); hi = lo + int(rng.randint(100, 30000))
        x = np.empty(n); i = 0
        level = rng.uniform(-200, 200)
        while i < n:
            d = int(rng.randint(lo, hi))
            level = rng.uniform(-200, 200) if rng.rand() < 0.9 else level + rng.normal(0, 0.3)
            x[i:i + d] = level
            i += d
        sigma = float(rng.choice([1e-6, 0.01, 0.3, 1.0, 7.0]))
        x += rng.normal(0, sigma, n)
        if rng.rand() < 0.2:                                 # a smoothed (correlated) current, like a filtered event
            k = np.ones(9) / 9.0
            x = np.convolve(x, k, mode="same")
        if rng.rand() < 0.2:
            x += rng.uniform(1e3, 1e5)                       # a large DC level: the cumsums round hard
        evs.append(x); starts.append(pos); pos += n + int(rng.randint(0, 9))
    buf = np.zeros(pos + 8)
    for x, a in zip(evs, starts):
        buf[a:a + x.size] = x
    t = torch.from_numpy(buf).cuda()
    b, off = ctx.segment_exact_f64(t, np.array(starts), np.array([x.size for x in evs]), _lib.split_params(**params))
    b = b.cpu().numpy()
    for e, x in enumerate(evs):
        ref = oracle.parse(x, **params)
        got = b[off[e]:off[e + 1]]
        nb += ref.size
        if not np.array_equal(got, ref):
            bad += 1
            print("MISMATCH seed %d event %d (%d samples, params %s): device %d boundaries, oracle %d; first differing %s / %s"
                  % (seed, e, x.size, params, got.size, ref.size, np.setdiff1d(got, ref)[:4], np.setdiff1d(ref, got)[:4]))
print("exact route fuzz: %d seeds, %d boundaries compared with the oracle, %d mismatching events, %.0f s" % (n_seeds, nb, bad, time.time() - t0))
sys.exit(1 if bad else 0)
