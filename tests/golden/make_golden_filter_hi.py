#!/usr/bin/env python3
"""Golden vectors for the two Event.filter cases round 4 added, from scipy -- what the reference calls
at DataTypes.py:258-274: scipy.signal.bessel(order, cutoff / nyquist, btype='low', analog=0, output='ba') and
scipy.signal.filtfilt(b, a, current):

  * orders 5..8 (DataTypes.py:266 passes any `order` on);
  * float64 INPUT on no ADC grid: a current that was filtered before (Experiment.parse twice on the same File objects:
    Event.filter filters whatever self.current holds), and un-quantised float64 noise.

    python tests/golden/make_golden_filter_hi.py

Outputs (committed): tests/golden/golden_filter_hi.npz, manifest_filter_hi.json.  The inputs are regenerated from
pypore_amd.synth integer specs; for the twice-filtered cases the golden file also holds scipy's FIRST output, which is the
float64 input of the second filter.  Tolerance per case as in make_golden_filter_order.py: 100 x the difference between
the oracle's restatement and scipy (a direct-form filter of order n amplifies last-bit coefficient differences), >= 1e-11."""
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


def filt(x, order, cutoff, second):
    b, a = signal.bessel(order, cutoff / (second / 2.), btype='low', analog=0, output='ba')
    return signal.filtfilt(b, a, x), b, a


# (name, [(order, cutoff), ...] applied one after the other, generator, sampling rate)
SPECS = [
    ("O5_2k", [(5, 2000.)], dict(kind="grid", n=20000, seed=81, lo=300, hi=3000), 1.e5),
    ("O6_5k", [(6, 5000.)], dict(kind="grid", n=20000, seed=82, lo=300, hi=3000), 1.e5),
    ("O7_10k", [(7, 10000.)], dict(kind="grid", n=12000, seed=83, lo=300, hi=3000), 1.e5),
    ("O8_10k", [(8, 10000.)], dict(kind="grid", n=12000, seed=84, lo=300, hi=3000), 1.e5),
    ("O8_5k_50kHz", [(8, 5000.)], dict(kind="grid", n=3 * 4096 + 29, seed=85, lo=300, hi=3000), 5.e4),
    ("O5_short", [(5, 2000.)], dict(kind="grid", n=19, seed=86, lo=2, hi=4), 1.e5),
    ("twice_O1_2k", [(1, 2000.), (1, 2000.)], dict(kind="grid", n=20000, seed=87, lo=300, hi=3000), 1.e5),
    ("O1_2k_then_O2_1k", [(1, 2000.), (2, 1000.)], dict(kind="grid", n=15000, seed=88, lo=1000, hi=5000), 1.e5),
    ("offgrid_O1_2k", [(1, 2000.)], dict(kind="offgrid", n=20000, seed=89, sigma=1.0), 1.e5),
    ("offgrid_O3_2k", [(3, 2000.)], dict(kind="offgrid", n=15000, seed=90, sigma=0.7), 1.e5),
    ("twice_O1_slow", [(1, 300.), (1, 300.)], dict(kind="grid", n=3 * 4096 + 5, seed=91, lo=2000, hi=9000), 1.e5),
]


def gen_input(g):
    if g["kind"] == "grid":
        return synth.random_dwell_counts(g["n"], g["seed"], g["lo"], g["hi"]).astype(np.float64) * synth.QUANTUM
    return synth.offgrid_trace(g["n"], g["seed"], sigma=g["sigma"])


if __name__ == "__main__":
    arrays, cases = {}, []
    for name, chain, gen, second in SPECS:
        x = gen_input(gen)
        xo = x.copy()
        for step, (order, cutoff) in enumerate(chain):
            if step == len(chain) - 1 and len(chain) > 1:
                arrays[name + "/input"] = x                    # scipy's first output: the float64 input of the last filter
            y, b, a = filt(x, order, cutoff, second)
            o = oracle.bessel_filtfilt(x, cutoff, second, order)
            x = y
        diff = float(np.max(np.abs(o - y)) / np.max(np.abs(y)))
        arrays[name + "/filtered"] = y
        cases.append(dict(name=name, chain=[list(c) for c in chain], gen=gen, second=second, n=int(len(xo)),
                          oracle_vs_scipy=diff, tol=max(1e-11, 100 * diff)))
        print(name, chain, len(xo), "oracle vs scipy (last filter) %.1e" % diff)
    np.savez_compressed(os.path.join(HERE, "golden_filter_hi.npz"), **arrays)
    with open(os.path.join(HERE, "manifest_filter_hi.json"), "w") as f:
        json.dump(dict(scipy_version=scipy.__version__, cases=cases), f, indent=1)
