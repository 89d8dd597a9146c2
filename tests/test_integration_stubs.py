"""INTEGRATION.md section 2 shows the bindings a PyPore maintainer would add (ctypes stubs against include/poreseg.h).  The
text is executed here as it stands -- only the library's path and the reference's `core` module are supplied -- and its
results are compared with the oracle and with this package's own classes: documentation that cannot rot."""
import os
import re
import sys

import numpy as np
import pytest

import oracle
from pypore_amd import _lib, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _blocks():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sec = text[text.index("## 2. The binding"):text.index("## 3. Entry points")]
    return re.findall(r"```python\n(.*?)```", sec, flags=re.S)


def test_the_stubs_are_where_the_test_expects_them():
    b = _blocks()
    assert len(b) == 3 and "class FastStatSplit" in b[0] and "class cSegmentAligner" in b[1] and "def parse_file_trace" in b[2]
    for name in re.findall(r"_L\.(ps_\w+)", "".join(b)):
        assert name in _lib.EXPORTS, name


@pytest.mark.gpu
def test_the_stubs_of_integration_md_run_and_agree_with_the_oracle(monkeypatch):
    from pypore_amd import core as ps_core
    monkeypatch.setitem(sys.modules, "core", ps_core)            # the reference's own core.py in a PyPore tree
    ns = {}
    b = _blocks()
    exec(b[0].replace('ctypes.CDLL("libporeseg.so")', "ctypes.CDLL(%r)" % _lib.LIB_PATH), ns)
    # the segmenter stub
    x = synth.config1()
    segs = ns["FastStatSplit"](prior_segments_per_second=10.).parse(x)
    np.testing.assert_array_equal([s.start for s in segs[1:]], oracle.parse(x, prior_segments_per_second=10.))
    assert abs(ns["FastStatSplit"](prior_segments_per_second=10.).min_gain - 18.4204807339517) < 1e-9
    with pytest.raises(AssertionError):
        ns["FastStatSplit"](min_width=100, window_width=150)
    # the aligner stub against this package's class
    exec(b[1], ns)
    from pypore_amd.calignment import cSegmentAligner
    rng = np.random.RandomState(3)
    mm, ms, md = rng.uniform(20, 80, 30), rng.uniform(0.5, 2, 30), rng.uniform(0.01, 0.2, 30)
    sm, ss, sd = mm[4:24] + rng.normal(0, 0.5, 20), ms[4:24], md[4:24]
    got = ns["cSegmentAligner"](mm, ms, md, 0.3, 0.1).align(sm, ss, sd)
    ref = cSegmentAligner(mm, ms, md, 0.3, 0.1).align(sm, ss, sd)
    assert got[0] == ref[0]
    np.testing.assert_array_equal(got[1], ref[1])
    # the file stub
    exec(b[2], ns)
    c, _ = synth.file_trace_counts(600000, 41, gap=30011, ev_lo=60000, ev_hi=200000)
    xf = synth.counts_to_pa(c, np.float64)
    params = ns["_Params"](100, 1000000, 10000, 0., 0., 10., 1e5, 0.)
    events = ns["parse_file_trace"](c.astype(np.int16), synth.QUANTUM, params, min_duration=1000)
    rs, rl = oracle.lambda_events(xf, threshold=90.0, min_duration=1000)
    assert [(s, l) for s, l, _ in events] == list(zip(rs.tolist(), rl.tolist())) and len(events) >= 2
    for s, l, bounds in events:
        np.testing.assert_array_equal(bounds, oracle.parse(xf[s:s + l], prior_segments_per_second=10.))
