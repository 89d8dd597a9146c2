#!/usr/bin/env python3
"""A SAMPLE of the two routes on which parity with the reference is empirical, large enough to quote a rate (VERDICT r4
next #6), recorded from the compiled, unmodified reference:

    ./oracle/build_reference.sh && python tests/golden/make_golden_sample.py

* FS000 .. FS103 -- filtered events, the reference's default workflow (Event.filter, DataTypes.py:258-274, then
  Event.parse, :276-289): a synthetic step event (integer spec) -> scipy.signal.bessel(order, cutoff / nyquist) +
  filtfilt in float64 -> cparsers.FastStatSplit.parse of the FILTERED float64 current (cparsers.pyx:103-118).  Orders
  1-4, cutoffs 500-5000 Hz, 1e5-1e6 samples, dwells from U[300, 3000) to U[5000, 50000), with and without
  cutoff_freq in the segmenter (Experiment.parse's default passes cutoff_freq=2000).
* OS00 .. OS51 -- float64 traces on no ADC grid (synth.offgrid_trace: two incommensurate noise streams), 1e4-2e6
  samples, sigma 0.2-4 pA, four parameter sets.
Stored: the reference's boundaries per case (int32).  Inputs are regenerated from the specs; the device test filters
with its own kernels (the product's workflow) and, for comparison, segments the scipy-filtered current as well.
Outputs (committed): tests/golden/golden_sample.npz + manifest_sample.json.
"""
import json
import os
import sys
import time

import numpy as np
import scipy
import scipy.signal as signal

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ref_shims          # noqa: E402
from pypore_amd import synth          # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
SECOND = 1.e5


def filtered_specs():
    specs = []
    cutoffs = [500., 1000., 2000., 3000., 5000.]
    dwells = [(300, 3000), (1000, 20000), (5000, 50000), (1000, 6000)]
    for i in range(104):
        h = int(synth.splitmix64(np.uint64(7700 + i)))
        n = int(round(10 ** (5.0 + (h % 1000) / 1000.0)))                   # 1e5 .. 1e6, log-uniform
        lo, hi = dwells[(i // 2) % 4]
        # (order, segmenter cutoff, filter cutoff and dwell range vary independently of each other)
        specs.append(dict(name="FS%03d" % i, n=n, seed=3000 + i, lo=lo, hi=hi, order=1 + i % 4,
                          cutoff=cutoffs[(i // 8) % 5], seg_cutoff=bool((i // 4) % 2)))
    return specs


def offgrid_specs():
    specs = []
    sig = [0.2, 0.5, 1.0, 2.0, 4.0]
    psets = [dict(), dict(cutoff_freq=2000.), dict(min_width=50, window_width=5000), dict(prior_segments_per_second=100.)]
    for i in range(52):
        h = int(synth.splitmix64(np.uint64(9100 + i)))
        n = int(round(10 ** (4.0 + 2.3 * (h % 1000) / 1000.0)))             # 1e4 .. 2e6
        specs.append(dict(name="OS%02d" % i, n=n, seed=5000 + i, sigma=sig[i % 5], lo=1000 if i % 3 else 300,
                          hi=20000 if i % 3 else 4000, params=psets[(i // 5) % 4]))
    return specs


def event_current(sp):
    return synth.random_dwell_counts(sp["n"], sp["seed"], sp["lo"], sp["hi"]).astype(np.float64) * synth.QUANTUM


def scipy_filter(x, order, cutoff, second=SECOND):
    (b, a) = signal.bessel(order, cutoff / (second / 2.), btype='low', analog=0, output='ba')     # DataTypes.py:266-270
    return signal.filtfilt(b, a, x)


def seg_params(sp):
    p = dict(min_width=100, max_width=1000000, window_width=10000, prior_segments_per_second=10., sampling_freq=SECOND)
    if sp.get("seg_cutoff"):
        p["cutoff_freq"] = sp["cutoff"]
    p.update(sp.get("params", {}))
    return p


def main():
    cparsers = ref_shims.load_cparsers()
    arrays, cases = {}, []
    t0 = time.time()
    for sp in filtered_specs():
        y = scipy_filter(event_current(sp), sp["order"], sp["cutoff"])
        segs = cparsers.FastStatSplit(**seg_params(sp)).parse(y)
        b = np.array([s.start for s in segs[1:]], dtype=np.int32)
        arrays[sp["name"]] = b
        cases.append(dict(sp, op="filtered", n_bounds=int(b.size), y_sum=repr(float(np.sum(y)))))
        print(sp["name"], sp["n"], sp["order"], sp["cutoff"], sp["seg_cutoff"], b.size, "%.0f s" % (time.time() - t0), flush=True)
    for sp in offgrid_specs():
        x = synth.offgrid_trace(sp["n"], sp["seed"], sp["sigma"], sp["lo"], sp["hi"])
        segs = cparsers.FastStatSplit(**seg_params(sp)).parse(x)
        b = np.array([s.start for s in segs[1:]], dtype=np.int32)
        arrays[sp["name"]] = b
        cases.append(dict(sp, op="offgrid", n_bounds=int(b.size), x_sum=repr(float(np.sum(x)))))
        print(sp["name"], sp["n"], sp["sigma"], b.size, "%.0f s" % (time.time() - t0), flush=True)
    np.savez_compressed(os.path.join(HERE, "golden_sample.npz"), **arrays)
    with open(os.path.join(HERE, "manifest_sample.json"), "w") as f:
        json.dump(dict(scipy_version=scipy.__version__, numpy_version=np.__version__, cases=cases), f, indent=1)
    print("wrote", len(arrays), "cases,", sum(int(a.size) for a in arrays.values()), "boundaries")


if __name__ == "__main__":
    main()
