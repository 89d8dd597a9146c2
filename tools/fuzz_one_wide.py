"""One seed of `FUZZ_SCALE=64 tools/fuzz_gpu.py` on both device routes (64-bit digest, LDS-window kernels), against each
other and the oracle, with the size of the reference's own prefix sums: n_event * max|k|^2 against 2^53."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import oracle
from pypore_amd import _lib, engine, synth
from pypore_amd import engine as _ps_engine
_ps_engine.apply_env_defaults()           # tools take their settings from PORESEG_* variables; the product reads none
seed, base, SCALE = int(sys.argv[1]), int(sys.argv[2]), 64
Q = synth.QUANTUM / SCALE
rng = np.random.RandomState(10_000 + seed + base)
mw = int(rng.choice([8, 20, 100, 250])); W = int(max(2 * mw, rng.choice([400, 1000, 4000, 10000, 25000])))
maxw = int(rng.choice([W, 3 * W, 50000, 1000000]))
params = dict(min_width=mw, max_width=max(maxw, mw), window_width=W, prior_segments_per_second=float(rng.choice([1., 10., 100.])))
n_ev = int(rng.choice([1, 1, 3, 7])); sigma = float(rng.choice([0.0, 1.0, 4.0, 30.0, 150.0])); dc = int(rng.choice([0, 0, 500, -3000, 9000]))
evs = []
for e in range(n_ev):
    n = int(rng.randint(3000, 300000 if n_ev == 1 else 60000))
    lo = int(rng.randint(50, 3000)); hi = lo + int(rng.randint(100, 30000))
    k = np.empty(n, dtype=np.int64); i = 0
    while i < n:
        d = int(rng.randint(lo, hi)); lvl = int(rng.randint(-2500, 2500)); k[i:i + d] = lvl; i += d
    if sigma > 0: k += np.rint(rng.normal(0.0, sigma, n)).astype(np.int64)
    k = np.clip(k + dc, -32000, 32000) * SCALE
    k = k + rng.randint(-(SCALE // 2), SCALE // 2 + 1, n)
    evs.append(k)
ctx = engine.context(0)
sp = _lib.split_params(**params)
print(params, "sigma", sigma, "dc", dc)
for e, k in enumerate(evs):
    ref = oracle.parse(k.astype(np.float64) * Q, **params)
    ref0 = oracle.parse((k - k[0]).astype(np.float64) * Q, **params)          # same data, the DC level removed: sums stay small
    t = torch.from_numpy((k.astype(np.float64) * Q).astype(np.float32)).cuda()
    got = {}
    for wide in (1, 0):
        ctx.set_option("wide_bs", wide)
        b, _, _ = ctx.segment_batch(t, np.array([0, len(k)], dtype=np.int64), sp, Q, want_stats=False)
        got[wide] = b.cpu().numpy()
    ctx.set_option("wide_bs", 1)
    print("event %d n %d  n*max|k|^2 / 2^53 = %.2f   oracle %d   oracle on k-k[0] %d   digest==oracle %s   lds==oracle %s   digest==lds %s   digest==oracle(k-k0) %s"
          % (e, len(k), len(k) * float(np.abs(k).max()) ** 2 / 2.0 ** 53, len(ref), len(ref0), np.array_equal(got[1], ref),
             np.array_equal(got[0], ref), np.array_equal(got[1], got[0]), np.array_equal(got[1], ref0)))
