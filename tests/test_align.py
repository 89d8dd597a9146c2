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


# ---- GPU: ps_align_batch through the C ABI -------------------------------------------------------------------

def _gpu_aligner(args):
    from pypore_amd.calignment import cSegmentAligner
    return cSegmentAligner(*args[:5])


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_gpu_matches_reference_golden(case):
    args = case_inputs(case["name"])
    al = _gpu_aligner(args)
    if case["raises"]:
        with pytest.raises(EXC[case["raises"]]):
            al.align(*args[5:])
        return
    score, path = al.align(*args[5:])
    assert score == float(GOLD[case["name"] + "/score"])              # bit-exact
    assert path.dtype == np.float64 and np.array_equal(path, GOLD[case["name"] + "/path"])


@pytest.mark.gpu
def test_gpu_batch_matches_oracle_bit_exact():
    """One model, 300 random sequences (lengths 1..120, every move, several noise levels) in one launch:
    raw score, path and status of every sequence equal the oracle's."""
    mm, ms, md, sp, bp, _, _, _ = random_case(99, m=70, s=1)
    seqs = []
    for q in range(300):
        c = random_case(1000 + q, m=70)
        rng = np.random.RandomState(q)
        idx = np.clip(np.cumsum(rng.choice([0, 1, 1, 1, 2, 3, -1, -2], size=len(c[5]))) + int(rng.randint(0, 20)), 0, 69)
        seqs.append((mm[idx] + rng.normal(0, 0.2, len(idx)), c[6], c[7]))
    seqs.append((np.zeros(0), np.zeros(0), np.zeros(0)))              # ValueError in the reference
    z = random_case(5, m=70, s=9)
    z[6][4] = 0.0
    seqs.append((z[5], z[6], z[7]))                                   # ZeroDivisionError
    from pypore_amd.calignment import cSegmentAligner
    al = cSegmentAligner(mm, ms, md, sp, bp)
    scores, paths, status = al.align_batch_raw(seqs)
    seen = set()
    for q, sq in enumerate(seqs):
        rc, score, path = oracle.align_raw(mm, ms, md, sp, bp, *sq)
        seen.add(rc)
        assert status[q] == rc, (q, status[q], rc)
        if rc == 0:
            assert scores[q] == score and np.array_equal(paths[q], path), q
    assert {0, 1, 3} <= seen


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(12))
def test_gpu_random_models_match_oracle(seed):
    """Model sizes 2..1024 (one LDS-resident row set), penalties 0.1..100, single align() calls and the exception class."""
    rng = np.random.RandomState(700 + seed)
    m = int(rng.choice([2, 3, 63, 64, 65, 129, 300, 1024]))
    args = random_case(800 + seed, m=m, s=int(rng.randint(1, 200)))
    al = _gpu_aligner(args)
    try:
        want = oracle.align(*args)
    except Exception as e:                                            # noqa: BLE001
        with pytest.raises(type(e)):
            al.align(*args[5:])
        return
    got = al.align(*args[5:])
    assert got[0] == want[0] and np.array_equal(got[1], want[1])


@pytest.mark.gpu
def test_gpu_wrapper_and_limits():
    from pypore_amd.alignment import SegmentAligner
    args = case_inputs("A2_walk_30x25")
    sa = SegmentAligner(*args[:5])
    score, order = sa.align(*args[5:])
    assert score == float(GOLD["A2_walk_30x25/score"]) and np.array_equal(order, GOLD["A2_walk_30x25/path"])
    assert sa.align(np.zeros(0), np.zeros(0), np.zeros(0)) == (None, None)         # alignment.py:43-46
    big = SegmentAligner(np.arange(1025.), np.ones(1025), np.ones(1025), 1., 1.)
    with pytest.raises(ValueError):                                    # PS_ERR_ARG: model longer than the LDS row set
        big.aligner.align(np.ones(3), np.ones(3), np.ones(3))
