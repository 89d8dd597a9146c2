#!/usr/bin/env python3
"""Golden vectors for traces on a REAL .abf scale: the boundaries the compiled, unmodified
reference cparsers.FastStatSplit finds on float64 `counts * scale + offset`, with scale and offset as the reference
reader computes them from fp32 header fields (read_abf.py:202-205) -- never a power of two -- and per-segment mean / std.

    ./oracle/build_reference.sh && python tests/golden/make_golden_scale.py

The reference's prefix sums of such values are rounded (sequential float64 cumsum of numbers that are not on a
binary grid); the device decides on exact integer sums of the counts.  These vectors pin that the two agree.
Outputs (committed): tests/golden/golden_scale.npz + manifest_scale.json; inputs are regenerated from synth specs.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ref_shims          # noqa: E402
from pypore_amd import synth          # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
cparsers = ref_shims.load_cparsers()
f32 = np.float32


def header_scale(adc_range, instrument_scale, signal_gain, programmable_gain, resolution, telegraph=None):
    s = float(f32(adc_range)) / float(f32(instrument_scale)) / float(f32(signal_gain)) / float(f32(programmable_gain)) / resolution
    return s / float(f32(telegraph)) if telegraph else s


SCALES = {
    "axopatch_x20": (header_scale(10.0, 0.0005, 1.0, 20.0, 32768), float(f32(0.25)) - float(f32(-1.5))),
    "gain_0p01": (header_scale(10.24, 0.01, 1.0, 1.0, 32768), 0.0),
    "telegraph_5": (header_scale(10.0, 0.001, 1.0, 1.0, 32768, 5.0), float(f32(-0.125))),
}
DEF = dict(min_width=100, max_width=1000000, window_width=10000, prior_segments_per_second=10.)
CASES = [
    ("S1_1e5", dict(kind="random_dwell", n=100000, seed=101), "axopatch_x20", DEF),
    ("S2_3e5_dense", dict(kind="random_dwell", n=300000, seed=102, lo=300, hi=3000), "axopatch_x20", DEF),
    ("S3_1e6", dict(kind="random_dwell", n=1000000, seed=103), "gain_0p01", DEF),
    ("S4_1e6_tel", dict(kind="random_dwell", n=1000000, seed=104), "telegraph_5", DEF),
    ("S5_step_cut2000", dict(kind="step", n=200000, dwell=10000, seed=105), "axopatch_x20",
     dict(min_width=100, max_width=1000000, window_width=10000, prior_segments_per_second=10., cutoff_freq=2000.)),
    ("S6_3e6_w4000", dict(kind="random_dwell", n=3000000, seed=106), "axopatch_x20",
     dict(min_width=50, max_width=200000, window_width=4000, prior_segments_per_second=10.)),
    ("S7_1e7", dict(kind="random_dwell", n=10000000, seed=107), "axopatch_x20", DEF),
    ("S8_1e7_gain", dict(kind="random_dwell", n=10000000, seed=108), "gain_0p01", DEF),
]


def counts_of(gen):
    if gen["kind"] == "step":
        return synth.step_counts(gen["n"], gen["dwell"], gen["seed"])
    return synth.random_dwell_counts(gen["n"], gen["seed"], gen.get("lo", 1000), gen.get("hi", 20000))


def main():
    arrays, cases = {}, []
    for name, gen, scale_name, params in CASES:
        scale, offset = SCALES[scale_name]
        k = counts_of(gen)
        x = np.array(k, dtype=np.float64) * scale + offset           # read_abf.py:210
        p = cparsers.FastStatSplit(**params)
        segs = p.parse(x)
        bounds = np.array([s.start for s in segs[1:]], dtype=np.int32)
        arrays[name + "/bounds"] = bounds
        arrays[name + "/mean"] = np.array([s.mean for s in segs], dtype=np.float64)
        arrays[name + "/std"] = np.array([s.std for s in segs], dtype=np.float64)
        cases.append(dict(name=name, gen=gen, params=params, scale=repr(scale), offset=repr(offset), n=int(len(x)),
                          n_bounds=int(len(bounds)), min_gain=repr(float(p.min_gain))))
        print(name, len(bounds))
    np.savez_compressed(os.path.join(HERE, "golden_scale.npz"), **arrays)
    with open(os.path.join(HERE, "manifest_scale.json"), "w") as f:
        json.dump({"generator": "tests/golden/make_golden_scale.py", "cases": cases}, f, indent=1)


if __name__ == "__main__":
    main()
