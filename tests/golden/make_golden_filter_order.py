#!/usr/bin/env python3
"""Golden vectors for Event.filter with orders 2..4: what the reference computes at
DataTypes.py:258-274 -- scipy.signal.bessel(order, cutoff / nyquist, btype='low', analog=0, output='ba') followed by
scipy.signal.filtfilt(b, a, current) -- with the scipy installed in the build container (the reference pins none).

    python tests/golden/make_golden_filter_order.py

Outputs (committed): tests/golden/golden_filter_order.npz (+ the (b, a) coefficients per case) and
manifest_filter_order.json.  Inputs are regenerated from pypore_amd.synth integer specs.

A note on what these vectors can pin.  An n-th order direct-form (b, a) filter is ill-conditioned for low cutoffs:
a last-bit difference in a coefficient changes the output by 1e-13 (order 2), 2e-11 (order 3, 500 Hz at 100 kHz),
1e-9 (order 4, 500 Hz), 1e-7 (order 5) ... 2e-2 (order 8, 500 Hz) of its range -- measured against scipy with the
oracle's restatement of the same design.  The reference's own output is that uncertain; the tests use the per-case
tolerance recorded here (100 x the oracle-vs-scipy difference, at least 1e-11)."""
import json
import os
import sys

import numpy as np
import scipy
import scipy.signal as signal

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle                          # noqa: E402
from pypore_amd import synth          # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
SPECS = [
    ("O2_2k", 2, dict(n=60000, seed=71, lo=300, hi=3000), 2000., 1.e5),
    ("O3_2k", 3, dict(n=60000, seed=72, lo=300, hi=3000), 2000., 1.e5),
    ("O4_2k", 4, dict(n=60000, seed=73, lo=300, hi=3000), 2000., 1.e5),
    ("O2_500", 2, dict(n=40000, seed=74, lo=1000, hi=5000), 500., 1.e5),
    ("O3_10k", 3, dict(n=40000, seed=75, lo=300, hi=3000), 10000., 1.e5),
    ("O4_5k_50kHz", 4, dict(n=3 * 4096 + 17, seed=76, lo=300, hi=3000), 5000., 5.e4),
    ("O3_short", 3, dict(n=13, seed=77, lo=2, hi=4), 2000., 1.e5),
    ("O4_1k", 4, dict(n=50000, seed=78, lo=1000, hi=5000), 1000., 1.e5),
]
arrays, cases = {}, []
for name, order, gen, cutoff, second in SPECS:
    x = synth.random_dwell_counts(gen["n"], gen["seed"], gen["lo"], gen["hi"]).astype(np.float64) * synth.QUANTUM
    b, a = signal.bessel(order, cutoff / (second / 2.), btype='low', analog=0, output='ba')
    y = signal.filtfilt(b, a, x)
    o = oracle.bessel_filtfilt(x, cutoff, second, order)
    diff = float(np.max(np.abs(o - y)) / np.max(np.abs(y)))
    arrays[name + "/filtered"] = y
    arrays[name + "/b"] = b
    arrays[name + "/a"] = a
    cases.append(dict(name=name, order=order, gen=gen, cutoff=cutoff, second=second, n=int(len(x)),
                      oracle_vs_scipy=diff, tol=max(1e-11, 100 * diff)))
    print(name, order, cutoff, len(x), "oracle vs scipy %.1e" % diff)
np.savez_compressed(os.path.join(HERE, "golden_filter_order.npz"), **arrays)
with open(os.path.join(HERE, "manifest_filter_order.json"), "w") as f:
    json.dump(dict(scipy_version=scipy.__version__, cases=cases), f, indent=1)
