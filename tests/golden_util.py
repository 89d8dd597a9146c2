"""Loader for the golden vectors recorded from the compiled reference (tests/golden/make_golden.py)."""
import json
import os

import numpy as np

from pypore_amd import synth

HERE = os.path.dirname(os.path.abspath(__file__))
_npz = None
_manifest = None


def npz():
    global _npz
    if _npz is None:
        _npz = np.load(os.path.join(HERE, "golden", "golden.npz"))
    return _npz


def manifest():
    global _manifest
    if _manifest is None:
        with open(os.path.join(HERE, "golden", "manifest.json")) as f:
            _manifest = json.load(f)
    return _manifest


def cases(op):
    return [c for c in manifest()["cases"] if c["op"] == op]


def case_ids(op):
    return [c["name"] for c in cases(op)]


def input_counts(case):
    gen = case["gen"]
    kind = gen["kind"]
    if kind == "step":
        c = synth.step_counts(gen["n"], gen["dwell"], gen["seed"], gen.get("level_offset", 0))
    elif kind == "random_dwell":
        c = synth.random_dwell_counts(gen["n"], gen["seed"], gen.get("lo", 1000), gen.get("hi", 20000))
    elif kind == "stored":
        c = npz()[gen["key"]]
    else:
        raise ValueError(kind)
    return np.asarray(c[case.get("offset", 0):], dtype=np.int32)


def input_pa(case, dtype=np.float64):
    return synth.counts_to_pa(input_counts(case), dtype)


_npz_off = None
_manifest_off = None


def offgrid_npz():
    global _npz_off
    if _npz_off is None:
        _npz_off = np.load(os.path.join(HERE, "golden", "golden_offgrid.npz"))
    return _npz_off


def offgrid_cases(op):
    """Cases of tests/golden/manifest_offgrid.json (make_golden_offgrid.py): op 'parse_offgrid' or 'score_samples'."""
    global _manifest_off
    if _manifest_off is None:
        with open(os.path.join(HERE, "golden", "manifest_offgrid.json")) as f:
            _manifest_off = json.load(f)
    return [c for c in _manifest_off["cases"] if c["op"] == op]
