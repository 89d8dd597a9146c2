#!/usr/bin/env python3
"""Golden vectors for the segment aligner: what the compiled, unmodified reference
PyPore/calignment.pyx cSegmentAligner(model_means, model_stds, model_durs, skip_penalty, backslip_penalty)
.align(seq_means, seq_stds, seq_durs) returns -- (score, float64 array of model indices) -- or which exception
it raises, on seeded synthetic models and sequences.

    ./oracle/build_reference.sh && python tests/golden/make_golden_align.py

Outputs (committed): tests/golden/golden_align.npz + tests/golden/manifest_align.json.  Inputs are stored with the
outputs (they are a few hundred float64 values per case).  Cases whose final scores are all <= -1 are skipped: the
reference's double_argmax (calignment.pyx:11-18) returns an uninitialised int there.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle                         # noqa: E402  (only to recognise the reference's undefined domain)
from oracle import ref_shims          # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
calignment = ref_shims.load_calignment()


def make_case(seed, m, s, noise, dur_hi, sp, bp, zero_std=False):
    """A model of m levels and a sequence of s segments that walks through it with stays, steps, skips and
    backslips; durations U[0.001, dur_hi) seconds, stds U[0.5, 3)."""
    rng = np.random.RandomState(seed)
    mm = np.cumsum(rng.uniform(-8, 10, m)) + 40
    ms = rng.uniform(0.5, 3, m)
    md = rng.uniform(0.001, dur_hi, m)
    j = int(rng.randint(1, max(2, m // 3 + 1))) if m > 1 else 0
    idx = []
    for _ in range(s):
        idx.append(j)
        r = rng.rand()
        j = (j if r < 0.2 else min(m - 1, j + 1) if r < 0.7 else
             min(m - 1, j + int(rng.randint(2, 5))) if r < 0.85 else max(min(1, m - 1), j - int(rng.randint(1, 4))))
    idx = np.array(idx, dtype=int)
    sm = mm[idx] + rng.normal(0, noise, s) if s else np.zeros(0)
    ss = rng.uniform(0.5, 3, s)
    sd = rng.uniform(0.001, dur_hi, s)
    if zero_std and s:
        ss[int(rng.randint(0, s))] = 0.0
    return mm, ms, md, float(sp), float(bp), sm, ss, sd


SPECS = [   # name, seed, m, s, noise, dur_hi, skip_penalty, backslip_penalty
    ("A1_small", 1, 8, 6, 0.05, 0.05, 0.5, 0.5),
    ("A2_walk_30x25", 2, 30, 25, 0.3, 0.02, 2.0, 2.0),
    ("A3_walk_64x64", 3, 64, 64, 0.2, 0.01, 0.5, 10.0),
    ("A4_walk_65x40", 4, 65, 40, 0.2, 0.01, 10.0, 0.5),
    ("A5_wide_model_200x50", 5, 200, 50, 0.1, 0.005, 1.0, 1.0),
    ("A6_long_seq_40x300", 6, 40, 300, 0.05, 0.001, 0.5, 0.5),
    ("A7_m2", 7, 2, 5, 0.05, 0.05, 0.5, 0.5),
    ("A8_s1", 8, 12, 1, 0.05, 0.05, 0.5, 0.5),
    ("A9_m1_s1", 9, 1, 1, 0.05, 0.05, 0.5, 0.5),
    ("A10_m1_s3_indexerror", 10, 1, 3, 0.05, 0.05, 0.5, 0.5),
    ("A11_s0_valueerror", 11, 6, 0, 0.05, 0.05, 0.5, 0.5),
    ("A12_cheap_moves", 12, 20, 30, 0.5, 0.02, 0.1, 0.1),
    ("A13_dear_moves", 13, 20, 30, 0.05, 0.002, 100.0, 100.0),
]
arrays, cases = {}, []
names = ("model_means", "model_stds", "model_durs", "skip_penalty", "backslip_penalty", "seq_means", "seq_stds", "seq_durs")
seed_extra = 1000
for name, seed, m, s, noise, dur_hi, sp, bp in SPECS:
    case = make_case(seed, m, s, noise, dur_hi, sp, bp)
    for k, v in zip(names, case):
        arrays[name + "/" + k] = np.asarray(v, dtype=np.float64)
    rec = dict(name=name, m=m, s=s)
    assert oracle.align_raw(*case)[0] != 4, name + ": all final scores <= -1 (undefined in the reference), pick another seed"
    try:
        score, path = calignment.cSegmentAligner(*case[:5]).align(*case[5:])
        arrays[name + "/score"] = np.float64(score)
        arrays[name + "/path"] = np.asarray(path, dtype=np.float64)
        rec["raises"] = None
    except Exception as e:                                    # noqa: BLE001 -- the class is the datum
        rec["raises"] = type(e).__name__
    cases.append(rec)

# a zero std: ZeroDivisionError
case = make_case(20, 10, 8, 0.05, 0.05, 0.5, 0.5, zero_std=True)
for k, v in zip(names, case):
    arrays["A14_zero_std/" + k] = np.asarray(v, dtype=np.float64)
try:
    calignment.cSegmentAligner(*case[:5]).align(*case[5:])
    cases.append(dict(name="A14_zero_std", m=10, s=8, raises=None))
except Exception as e:                                        # noqa: BLE001
    cases.append(dict(name="A14_zero_std", m=10, s=8, raises=type(e).__name__))

# a sequence that sits on the first model segment: the traceback reaches j == 0 early -> IndexError
mm = np.array([10., 30., 50., 70.]); ms = np.ones(4); md = np.full(4, 0.05)
sm = np.array([10.01, 9.99, 10.0, 30.0]); ss = np.ones(4); sd = np.full(4, 0.05)
case = (mm, ms, md, 0.5, 0.5, sm, ss, sd)
for k, v in zip(names, case):
    arrays["A15_first_level/" + k] = np.asarray(v, dtype=np.float64)
try:
    calignment.cSegmentAligner(*case[:5]).align(*case[5:])
    cases.append(dict(name="A15_first_level", m=4, s=4, raises=None))
except Exception as e:                                        # noqa: BLE001
    cases.append(dict(name="A15_first_level", m=4, s=4, raises=type(e).__name__))

np.savez_compressed(os.path.join(HERE, "golden_align.npz"), **arrays)
json.dump(dict(reference="PyPore/calignment.pyx cSegmentAligner (compiled unmodified: oracle/build_reference.sh)",
               cases=cases), open(os.path.join(HERE, "manifest_align.json"), "w"), indent=1)
for c in cases:
    print(c)
