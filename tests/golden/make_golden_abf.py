#!/usr/bin/env python3
"""Golden vectors for the ABF reader: what the reference's OWN reader,
/root/reference/PyPore/read_abf.py:22-212, returns on files written by pypore_amd.abf.write_abf.

The reference reader is imported unmodified (one shim: numpy.float = float, removed from numpy >= 1.24,
read_abf.py:210) in the build container; it cannot travel, so its outputs are committed as a fixture:
per case the writer arguments (the file is regenerated from pypore_amd.synth integer specs by the
tests), time_step_msec, n, SHA-256 of the float64 current, and its first / last 16 values.

    python tests/golden/make_golden_abf.py

Cases: the exact power-of-two scale of the other goldens; a realistic patch-clamp header
(fADCRange 10 V, fInstrumentScaleFactor 0.0005 V/pA, gain 20, 32768 counts: scale 0.030517578...,
with both offsets); telegraph gain on (read_abf.py:203); a 2-channel and a 3-channel interleave
(read_abf.py:208-210); a short file; negative counts at the int16 limits.
"""
import hashlib
import importlib.util
import json
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from pypore_amd import abf, synth          # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def load_reference_reader():
    np.float = float                       # the only shim: numpy.float was an alias of the builtin
    sys.dont_write_bytecode = True
    spec = importlib.util.spec_from_file_location("ref_read_abf", "/root/reference/PyPore/read_abf.py")
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m.read_abf


def case_counts(spec):
    """int16 counts of channel 0 (and of the other channels) from an integer spec."""
    kind = spec["kind"]
    if kind == "file_trace":
        c, _ = synth.file_trace_counts(spec["n"], spec["seed"])
    elif kind == "random_dwell":
        c = synth.random_dwell_counts(spec["n"], spec["seed"])
    elif kind == "limits":
        c = np.array([-32768, 32767, 0, -1, 1, -32768, 32767] * 11, dtype=np.int64)[:spec["n"]]
    else:
        raise ValueError(kind)
    c = np.asarray(c, dtype=np.int64) + spec.get("shift", 0)
    others = [((np.arange(c.size, dtype=np.int64) * (7 + 4 * k)) % 4001 - 2000).astype(np.int16)
              for k in range(spec.get("extra_channels", 0))]
    return c.astype(np.int16), others


CASES = [
    ("A1_pow2", dict(kind="file_trace", n=400000, seed=11), dict()),
    ("A2_realistic", dict(kind="file_trace", n=400000, seed=12),
     dict(adc_range=10.0, adc_resolution=32768, instrument_scale=0.0005, signal_gain=1.0, programmable_gain=20.0,
          instrument_offset=0.25, signal_offset=-1.5, sampling_interval_us=10.0)),
    ("A3_telegraph", dict(kind="random_dwell", n=100000, seed=13),
     dict(adc_range=10.0, adc_resolution=32768, instrument_scale=0.001, programmable_gain=1.0, telegraph_gain=5.0,
          sampling_interval_us=4.0)),
    ("A4_two_channels", dict(kind="random_dwell", n=60000, seed=14, extra_channels=1),
     dict(adc_range=10.0, adc_resolution=32768, instrument_scale=0.0005, programmable_gain=20.0, sampling_interval_us=20.0)),
    ("A5_three_channels", dict(kind="random_dwell", n=30001, seed=15, extra_channels=2), dict(signal_offset=3.0)),
    ("A6_short", dict(kind="random_dwell", n=7, seed=16), dict()),
    ("A7_limits", dict(kind="limits", n=77),
     dict(adc_range=10.24, adc_resolution=32768, instrument_scale=0.01, instrument_offset=-0.125)),
]


def main():
    ref_read = load_reference_reader()
    out = {"generator": "tests/golden/make_golden_abf.py", "reference": "PyPore/read_abf.py:22-212 (unmodified; numpy.float shim)",
           "numpy": np.__version__, "cases": []}
    with tempfile.TemporaryDirectory() as tmp:
        for name, spec, wargs in CASES:
            counts, others = case_counts(spec)
            path = os.path.join(tmp, name + ".abf")
            abf.write_abf(path, counts, other_channels=others, **wargs)
            dt, cur = ref_read(path)
            cur = np.ascontiguousarray(cur, dtype=np.float64)
            assert cur.size == counts.size
            out["cases"].append({
                "name": name, "spec": spec, "write_args": wargs, "time_step_msec": repr(float(dt)), "n": int(cur.size),
                "sha256": hashlib.sha256(cur.tobytes()).hexdigest(),
                "first": [repr(float(v)) for v in cur[:16]], "last": [repr(float(v)) for v in cur[-16:]],
            })
            print(name, dt, cur.size, cur[:3])
    with open(os.path.join(HERE, "manifest_abf.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
