"""The ABF reader against the reference's own reader, and traces on a real (non-power-of-two) .abf scale.

* tests/golden/manifest_abf.json: outputs of PyPore/read_abf.py (unmodified) on files written by abf.write_abf --
  power-of-two scale, a patch-clamp header (10 V / 0.0005 V/pA / x20 / 32768, both offsets), telegraph gain,
  2- and 3-channel interleave, short file, int16 limits (tests/golden/make_golden_abf.py).
* tests/golden/golden_scale.npz: boundaries / mean / std the compiled reference finds on float64
  counts * scale + offset for three such scales (tests/golden/make_golden_scale.py).
The device never sees those float64 values: File(filename) and bare arrays alike go up as int16 counts
(pypore_amd.grid), so the GPU tests here are the evidence that real files take the drop-in path with the reference's results.
"""
import hashlib
import json
import os
import sys

import numpy as np
import pytest

from pypore_amd import abf, synth
from pypore_amd.grid import GridArray, affine_grid, grid_of

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
ABF = json.load(open(os.path.join(HERE, "golden", "manifest_abf.json")))["cases"]
SCALE = json.load(open(os.path.join(HERE, "golden", "manifest_scale.json")))["cases"]
_scale_npz = None


def scale_npz():
    global _scale_npz
    if _scale_npz is None:
        _scale_npz = np.load(os.path.join(HERE, "golden", "golden_scale.npz"))
    return _scale_npz


def abf_case_file(case, tmp_path):
    import make_golden_abf as g                      # the integer specs of the inputs (no reference import at module level)
    counts, others = g.case_counts(case["spec"])
    path = os.path.join(str(tmp_path), case["name"] + ".abf")
    abf.write_abf(path, counts, other_channels=others, **case["write_args"])
    return path, counts


def scale_case_input(case):
    gen = case["gen"]
    k = synth.step_counts(gen["n"], gen["dwell"], gen["seed"]) if gen["kind"] == "step" else \
        synth.random_dwell_counts(gen["n"], gen["seed"], gen.get("lo", 1000), gen.get("hi", 20000))
    return k, float(case["scale"]), float(case["offset"])


@pytest.mark.parametrize("case", ABF, ids=[c["name"] for c in ABF])
def test_reader_equals_reference_reader(case, tmp_path):
    """abf.read_abf == PyPore/read_abf.py:22-212 bit for bit: time step, length, every float64 (SHA-256), first / last 16."""
    path, counts = abf_case_file(case, tmp_path)
    dt, cur = abf.read_abf(path)
    assert repr(float(dt)) == case["time_step_msec"] and cur.size == case["n"] and cur.dtype == np.float64
    assert hashlib.sha256(np.ascontiguousarray(cur).tobytes()).hexdigest() == case["sha256"]
    assert [repr(float(v)) for v in cur[:16]] == case["first"] and [repr(float(v)) for v in cur[-16:]] == case["last"]
    # the array still knows its counts, and slices keep them
    k, q, o = grid_of(cur)
    assert np.array_equal(k, counts) and np.array_equal(grid_of(cur[3:5])[0], counts[3:5]) and grid_of(cur * 1.0) is None
    dt2, raw, scale, offset = abf.read_abf_counts(path)
    assert (q, o) == (scale, offset) and np.array_equal(raw, counts)


def test_affine_grid_recovers_counts_of_bare_arrays():
    for case in SCALE[:4]:
        k, scale, offset = scale_case_input(case)
        x = np.array(k, dtype=np.float64) * scale + offset
        q, o, kr = affine_grid(x)
        assert abs(q / scale - 1) < 1e-10 and np.array_equal(kr - kr[0], k - k[0])
        assert np.max(np.abs(kr * q + o - x)) < 1e-9
    with pytest.raises(ValueError):
        affine_grid(np.random.default_rng(3).normal(size=5000))


@pytest.mark.parametrize("case", SCALE, ids=[c["name"] for c in SCALE])
def test_oracle_matches_reference_on_real_scales(case):
    """The CPU restatement on the float64 values the reference saw."""
    import oracle
    k, scale, offset = scale_case_input(case)
    x = np.array(k, dtype=np.float64) * scale + offset
    b = oracle.parse(x, **case["params"])
    assert np.array_equal(b, scale_npz()[case["name"] + "/bounds"])


@pytest.mark.gpu
@pytest.mark.parametrize("case", SCALE, ids=[c["name"] for c in SCALE])
def test_device_int16_route_matches_reference_on_real_scales(case):
    """Bare float64 array in (no hint): the grid is recovered, int16 counts go to the device, boundaries bit-exact,
    mean / std within 1e-5 of the reference's numpy values on the float64 current."""
    from pypore_amd.parsers import SpeedyStatSplit
    k, scale, offset = scale_case_input(case)
    x = np.array(k, dtype=np.float64) * scale + offset
    segs = SpeedyStatSplit(**case["params"]).parse(x)
    got = np.array([s.start for s in segs[1:]], dtype=np.int32)
    assert np.array_equal(got, scale_npz()[case["name"] + "/bounds"])
    np.testing.assert_allclose([s.mean for s in segs], scale_npz()[case["name"] + "/mean"], rtol=1e-5)
    np.testing.assert_allclose([s.std for s in segs], scale_npz()[case["name"] + "/std"], rtol=1e-5)
    assert segs[1].current.base is not None and segs[1].current[0] == x[got[0]]          # views of the caller's array


@pytest.mark.gpu
def test_explicit_grid_and_gridarray_routes_agree():
    from pypore_amd.parsers import SpeedyStatSplit
    case = SCALE[0]
    k, scale, offset = scale_case_input(case)
    x = np.array(k, dtype=np.float64) * scale + offset
    ref = scale_npz()[case["name"] + "/bounds"]
    for cur, kw in ((x, dict(quantum=scale, offset=offset)), (GridArray.from_counts(k.astype(np.int16), scale, offset), {})):
        segs = SpeedyStatSplit(**case["params"], **kw).parse(cur)
        assert np.array_equal([s.start for s in segs[1:]], ref)
        assert segs[0].mean == pytest.approx(float(np.mean(x[:ref[0]])), rel=1e-9)
        assert segs[0].min == pytest.approx(float(np.min(x[:ref[0]])), rel=1e-12)
    with pytest.raises(ValueError):
        SpeedyStatSplit(**case["params"], quantum=scale * 1.37, offset=offset).parse(x)


@pytest.mark.gpu
def test_file_with_real_header_parses_end_to_end(tmp_path):
    """File(filename).parse() then Event.parse(SpeedyStatSplit) on an .abf with a patch-clamp header (ADVICE r1 medium:
    this used to raise): events and boundaries equal the oracle's on the float64 current the reference reader returns."""
    import oracle
    from pypore_amd.DataTypes import File
    from pypore_amd.parsers import SpeedyStatSplit, lambda_event_parser
    case = [c for c in ABF if c["name"] == "A2_realistic"][0]
    path, counts = abf_case_file(case, tmp_path)
    f = File(path)
    assert f.second == 100000.0
    f.parse(lambda_event_parser(threshold=90))
    x = np.asarray(f.current)
    es, el = oracle.lambda_events(x, threshold=90.0)
    assert [(int(round(ev.start * f.second)), int(round(ev.duration * f.second))) for ev in f.events] == list(zip(es.tolist(), el.tolist()))
    assert len(f.events) >= 1
    for ev in f.events:
        ev.parse(SpeedyStatSplit(prior_segments_per_second=10.))
        a = int(round(ev.start * f.second))
        ref = oracle.parse(x[a:a + len(ev.current)], prior_segments_per_second=10.)
        assert np.array_equal(np.rint(np.array([s.start for s in ev.segments[1:]]) * f.second).astype(np.int64), ref)
        st = oracle.segment_stats(x[a:a + len(ev.current)], ref)
        np.testing.assert_allclose([s.mean for s in ev.segments], st[:, 0], rtol=1e-5)
        np.testing.assert_allclose([s.std for s in ev.segments], st[:, 1], rtol=1e-5)
    f.parse_events(SpeedyStatSplit(prior_segments_per_second=10.))          # one device call for all events: same result
    for ev in f.events:
        a = int(round(ev.start * f.second))
        ref = oracle.parse(x[a:a + len(ev.current)], prior_segments_per_second=10.)
        assert np.array_equal(np.rint(np.array([s.start for s in ev.segments[1:]]) * f.second).astype(np.int64), ref)
    # the counts-only pipeline (no float64 at all) finds the same events
    from pypore_amd import pipeline
    dt, st_, ln_, bl = pipeline.parse_abf(path)
    assert list(zip(st_.tolist(), ln_.tolist())) == list(zip(es.tolist(), el.tolist()))


def test_gridarray_is_read_only():
    """an in-place edit of a reader's current (a baseline subtraction, a blanked artefact, a sort) returns the
    same object without passing __array_finalize__, so the int16 counts would go stale and the device would segment the
    unmodified trace.  The array is read-only: such an edit raises, a copy is a plain writable array without counts, and
    affine_grid re-centres counts that touch both int16 rails."""
    from pypore_amd.grid import GridArray, affine_grid, grid_of
    counts = np.array([-32768, 5, 7, 32767, 100, -3], dtype=np.int16)
    a = GridArray.from_counts(counts, 0.030517578125, 1.75)
    assert grid_of(a) is not None and grid_of(a[1:4])[0].tolist() == [5, 7, 32767]
    with pytest.raises(ValueError):
        a -= 3
    with pytest.raises(ValueError):
        a[2:4] = 100
    with pytest.raises(ValueError):
        a.sort()
    b = a.copy()
    assert type(b) is np.ndarray and b.flags.writeable and grid_of(b) is None
    b -= 3.0                                      # the caller's baseline subtraction: a bare array, grid found afresh
    q, o, k = affine_grid(b)
    assert abs(q - 0.030517578125) < 1e-12 and k.min() >= -32768 and k.max() <= 32767      # both rails: still int16
    np.testing.assert_allclose(k * q + o, b, rtol=0, atol=1e-9)
