"""Segment aligner (calignment.pyx:20-100).

CPU part: the oracle restatement against golden vectors recorded from the compiled, unmodified reference
(tests/golden/make_golden_align.py).  GPU part: ps_align_batch through the C ABI against the oracle --
scores and paths bit-exact (the kernel keeps the reference's operation order), error classes the same."""
import json
import os

import numpy as np
import pytest

import oracle

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = np.load(os.path.join(HERE, "golden", "golden_align.npz"))
CASES = json.load(open(os.path.join(HERE, "golden", "manifest_align.json")))["cases"]
KEYS = ("model_means", "model_stds", "model_durs", "skip_penalty", "backslip_penalty", "seq_means", "seq_stds", "seq_durs")
EXC = {"IndexError": IndexError, "ValueError": ValueError, "ZeroDivisionError": ZeroDivisionError}


def case_inputs(name):
    v = [GOLD[name + "/" + k] for k in KEYS]
    return v[0], v[1], v[2], float(v[3]), float(v[4]), v[5], v[6], v[7]


def random_case(seed, m=None, s=None):
    rng = np.random.RandomState(seed)
    m = int(rng.randint(2, 90)) if m is None else m
    s = int(rng.randint(1, 120)) if s is None else s
    mm = np.cumsum(rng.uniform(-8, 10, m)) + 40
    ms = rng.uniform(0.5, 3, m)
    md = rng.uniform(0.0005, float(rng.choice([0.002, 0.01, 0.05])), m)
    j = int(rng.randint(0, m))
    idx = []
    for _ in range(s):
        idx.append(j)
        r = rng.rand()
        j = (j if r < 0.2 else min(m - 1, j + 1) if r < 0.7 else
             min(m - 1, j + int(rng.randint(2, 5))) if r < 0.85 else max(0, j - int(rng.randint(1, 4))))
    sm = mm[np.array(idx, dtype=int)] + rng.normal(0, float(rng.choice([0.02, 0.2, 1.0])), s)
    ss = rng.uniform(0.5, 3, s)
    sd = rng.uniform(0.0005, 0.02, s)
    sp, bp = float(rng.choice([0.1, 0.5, 2., 10., 100.])), float(rng.choice([0.1, 0.5, 2., 10., 100.]))
    return mm, ms, md, sp, bp, sm, ss, sd


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_oracle_matches_reference_golden(case):
    args = case_inputs(case["name"])
    if case["raises"]:
        with pytest.raises(EXC[case["raises"]]):
            oracle.align(*args)
        return
    score, path = oracle.align(*args)
    assert score == float(GOLD[case["name"] + "/score"])              # bit-exact
    assert path.dtype == np.float64 and np.array_equal(path, GOLD[case["name"] + "/path"])


def test_golden_paths_exercise_every_move():
    moves = set()
    for c in CASES:
        if c["raises"]:
            continue
        d = np.diff(GOLD[c["name"] + "/path"])
        moves |= {"stay" if x == 0 else "step" if x == 1 else "skip" if x > 1 else "back" for x in d}
    assert moves == {"stay", "step", "skip", "back"}
