"""CPU tests of the host side: C ABI surface, parameter logic, class surface (no GPU compute)."""
import ctypes
import json
import os
import re

import numpy as np
import pytest

import oracle
from pypore_amd import _lib, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_loads_and_exports_every_declared_symbol():
    L = _lib.lib()
    hdr = open(os.path.join(ROOT, "include", "poreseg.h")).read()
    declared = set(re.findall(r"\b(ps_[a-z_0-9]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    for name in sorted(declared):
        assert hasattr(L, name), "libporeseg.so does not export %s" % name
    assert set(_lib.EXPORTS) == declared
    assert b"gfx950" in L.ps_version()


def test_comm_library_loads_and_exports_every_declared_symbol():
    """libporeseg_comm.so (include/poreseg_comm.h: SURVEY 8(b)'s ps_comm_init_all / ps_gather_bounds); no call that needs a GPU."""
    from pypore_amd import _comm
    L = _comm.lib()
    hdr = open(os.path.join(ROOT, "include", "poreseg_comm.h")).read()
    declared = set(re.findall(r"\b(ps_[a-z_0-9]+)\s*\(", hdr))
    assert declared == set(_comm.EXPORTS)
    for name in sorted(declared):
        assert hasattr(L, name), "libporeseg_comm.so does not export %s" % name
    assert int(re.search(r"#define PS_COMM_ID_BYTES (\d+)", hdr).group(1)) == _comm.ID_BYTES
    assert int(re.search(r"#define PS_GATHER_HEADER (\d+)", hdr).group(1)) == _comm.HEADER
    from pypore_amd import dist as pdist
    assert pdist.BoundaryGather.HEADER == _comm.HEADER
    assert L.ps_comm_world(None) == 0 and L.ps_comm_rank(None) == -1          # (NULL handles are answered, not dereferenced)
    h = ctypes.c_void_p()
    assert L.ps_comm_init_rank(2, 5, b"x" * 128, 0, ctypes.byref(h)) == -1 and b"rank 5" in L.ps_comm_last_error()


def test_struct_layouts_match_header():
    assert ctypes.sizeof(_lib.SplitParams) == 56
    assert ctypes.sizeof(_lib.SampleFormat) == 16


@pytest.mark.parametrize("kw", [
    dict(), dict(prior_segments_per_second=10.), dict(prior_segments_per_second=10., cutoff_freq=2000.),
    dict(min_gain_per_sample=0.5), dict(false_positive_rate=50., prior_segments_per_second=20.),
    dict(sampling_freq=5e4, prior_segments_per_second=3.), dict(min_gain_per_sample=0.05, window_width=1000),
])
def test_min_gain_matches_oracle(kw):
    assert repr(_lib.min_gain(**kw)) == repr(oracle.min_gain(**kw))


def test_constructor_assertions_like_reference():
    from pypore_amd.cparsers import FastStatSplit
    with pytest.raises(AssertionError):
        FastStatSplit(min_width=10, max_width=5)
    with pytest.raises(AssertionError):
        FastStatSplit(min_width=100, window_width=199)
    with pytest.raises(AssertionError):
        FastStatSplit(cutoff_freq=60000.)
    f = FastStatSplit(prior_segments_per_second=10)
    assert f.min_gain == 18.4204807339517


def test_parser_json_round_trip_and_attribute_names():
    from pypore_amd import parsers
    p = parsers.SpeedyStatSplit(min_width=50, prior_segments_per_second=10., cutoff_freq=2000.)
    d = json.loads(p.to_json())
    assert set(d) == {"min_width", "max_width", "window_width", "min_gain_per_sample", "false_positive_rate",
                      "prior_segments_per_second", "sampling_freq", "cutoff_freq", "name"}
    q = parsers.parser.from_json(p.to_json())
    assert isinstance(q, parsers.SpeedyStatSplit) and q.to_dict() == p.to_dict()


def test_no_gpu_means_loud_failure_not_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from pypore_amd.parsers import SpeedyStatSplit
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        SpeedyStatSplit(prior_segments_per_second=10.).parse(synth.config1())


def test_segment_surface():
    from pypore_amd.core import MetaSegment, Segment
    x = np.arange(10, dtype=np.float64)
    s = Segment(current=x[2:6], start=2, duration=4, end=6, mean=123.0)
    assert s.mean == 3.5 and s.n == 4 and s.min == 2 and s.max == 5           # cannot override stats
    s.scale(2.0)
    assert (s.start, s.end, s.duration) == (1.0, 3.0, 2.0)
    d = json.loads(s.to_json())
    assert d["name"] == "Segment" and d["mean"] == 3.5
    s.to_meta()
    assert isinstance(s, MetaSegment) and not hasattr(s, "current") and s.mean == 3.5


def test_lambda_event_parser_matches_oracle_and_golden():
    from golden_util import npz
    from pypore_amd.parsers import lambda_event_parser
    z = npz()
    x = z["G6_events/input"].astype(np.float64) * synth.QUANTUM
    # explicit rules (equal to the defaults) take the host route of the reference; the default-rule
    # object runs on the GPU (tests/test_gpu_parity.py)
    rules = [lambda event: event.duration > 100000, lambda event: event.min > -0.5, lambda event: event.max < 90]
    evs = lambda_event_parser(threshold=90, rules=rules).parse(x)
    assert [int(e.start) for e in evs] == list(z["G6_events/starts"])
    assert [int(e.duration) for e in evs] == list(z["G6_events/lengths"])


def test_detect_quantum():
    from pypore_amd import engine
    assert engine.detect_quantum(synth.config1()) == 2.0 ** -5
    assert engine.detect_quantum(np.array([1.0, 2.0, -7.0])) == 1.0
    assert engine.detect_quantum(np.array([0.75, 1.5])) == 0.25
    with pytest.raises(ValueError):
        engine.detect_quantum(np.array([0.1, 0.2]))


def test_synth_generators_are_stable():
    # digest of the integer generator: guards the bench/golden inputs against drift
    c = synth.random_dwell_counts(100000, 2024)
    assert c[:6].tolist() == synth.random_dwell_counts(6, 2024).tolist()
    import hashlib
    assert hashlib.sha256(c.astype(np.int32).tobytes()).hexdigest() == \
        hashlib.sha256(synth.random_dwell_counts(100000, 2024).astype(np.int32).tobytes()).hexdigest()
    assert abs(float(np.std(synth.noise_counts(9, 0, 200000))) - 32.0) < 0.2


def test_abf_writer_reader_round_trip(tmp_path):
    from pypore_amd import abf
    c = synth.step_counts(20000, 3000, 4).astype(np.int16)
    p = str(tmp_path / "a.abf")
    abf.write_abf(p, c)
    dt, x = abf.read_abf(p)
    assert dt == 0.01                                    # fADCSequenceInterval 10 us (read_abf.py:155)
    np.testing.assert_array_equal(x, c.astype(np.float64) * 2.0 ** -5)
    dt, k, scale, off = abf.read_abf_counts(p)
    assert scale == 2.0 ** -5 and off == 0.0 and np.array_equal(np.asarray(k), c)
    # a realistic non-power-of-two scale with offsets: same arithmetic as read_abf.py:202-205
    abf.write_abf(p, c, adc_range=10.0, adc_resolution=32768, instrument_scale=0.01, instrument_offset=0.25,
                  signal_offset=0.125)
    _, x2 = abf.read_abf(p)
    f32 = np.float32
    scale = float(f32(10.0)) / float(f32(0.01)) / 1.0 / 1.0 / 32768
    np.testing.assert_array_equal(x2, c.astype(np.float64) * scale + 0.125)
    with open(p, "r+b") as fh:
        fh.write(b"XXXX")
    with pytest.raises(ValueError):
        abf.read_abf(p)


def test_stream_pool_scheduling_without_a_gpu():
    """engine.StreamPool.run: job k runs on context k % T, results come back in job order, an exception on a worker
    thread is re-raised on the caller's, the pool survives it, close() ends the workers.  (Fake contexts: no GPU.)"""
    import queue
    import threading
    from pypore_amd import engine

    class FakePool(engine.StreamPool):
        def __init__(self, T):
            self.contexts = ["ctx%d" % t for t in range(T)]
            self._inbox = [queue.SimpleQueue() for _ in self.contexts]
            self._done = queue.SimpleQueue()
            self._threads = []
            for t in range(1, T):
                th = threading.Thread(target=self._worker, args=(t,), daemon=True)
                th.start()
                self._threads.append(th)

    pool = FakePool(4)
    seen = pool.run(10, lambda cx, k, t: (k, cx, t, threading.get_ident()))
    assert [(k, cx, t) for k, cx, t, _ in seen] == [(k, "ctx%d" % (k % 4), k % 4) for k in range(10)]
    assert len({ident for _, _, _, ident in seen}) == 4                   # four host threads
    assert pool.run(2, lambda cx, k, t: k * k) == [0, 1] and pool.run(0, lambda *a: 1) == []
    with pytest.raises(ZeroDivisionError):
        pool.run(5, lambda cx, k, t: 1 // (k - 3))
    assert pool.run(3, lambda cx, k, t: k) == [0, 1, 2]
    # dynamic=True: every job runs exactly once, on whichever context is free; a context stuck in a long job takes fewer
    import time
    took = pool.run(40, lambda cx, k, t: (time.sleep(0.05 if t == 2 else 0.001), k, t)[1:], dynamic=True)
    assert [k for k, _ in took] == list(range(40))
    share = [sum(1 for _, t in took if t == c) for c in range(4)]
    assert sum(share) == 40 and share[2] < min(share[0], share[1], share[3])
    pool.close()
