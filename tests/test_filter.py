"""Event.filter: order-1 Bessel filtfilt.  Golden vectors come from scipy (what the reference calls
at DataTypes.py:258-274) and from the compiled reference run on the filtered current (tests/golden/make_golden_filter.py)."""
import json
import os

import numpy as np
import pytest

import oracle
from pypore_amd import synth

HERE = os.path.dirname(os.path.abspath(__file__))
MAN = json.load(open(os.path.join(HERE, "golden", "manifest_filter.json")))
NPZ = np.load(os.path.join(HERE, "golden", "golden_filter.npz"))
TOL = 1e-11          # relative to max |y|: re-association of the scan / coefficient rounding, fp64 throughout


def _counts(gen):
    if gen["kind"] == "config2_event":
        return np.rint(synth.config2_event(gen["ev"], n=gen["n"], dtype=np.float64) / synth.QUANTUM).astype(np.int64)
    return synth.random_dwell_counts(gen["n"], gen["seed"], gen["lo"], gen["hi"])


@pytest.mark.parametrize("case", MAN["cases"], ids=[c["name"] for c in MAN["cases"]])
def test_oracle_filter_matches_scipy_golden(case):
    x = _counts(case["gen"]).astype(np.float64) * synth.QUANTUM
    ref = NPZ[case["name"] + "/filtered"]
    got = oracle.bessel_filtfilt(x, case["cutoff"], case["second"])
    assert np.max(np.abs(got - ref)) <= TOL * np.max(np.abs(ref))


def test_oracle_filter_rejects_what_scipy_rejects():
    with pytest.raises(ValueError):
        oracle.bessel_filtfilt(np.ones(6), 2000., 1e5)          # len(x) must exceed padlen = 6
    with pytest.raises(ValueError):
        oracle.bessel_filtfilt(np.ones(100), 60000., 1e5)       # cutoff above Nyquist


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["f32", "i16"])
@pytest.mark.parametrize("case", MAN["cases"], ids=[c["name"] for c in MAN["cases"]])
def test_filter_kernel_matches_scipy_golden(case, dtype):
    import torch
    from pypore_amd import engine
    ctx = engine.context(0)
    k = _counts(case["gen"])
    dev = torch.from_numpy(k.astype(np.int16)).cuda() if dtype == "i16" else \
        torch.from_numpy((k * synth.QUANTUM).astype(np.float32)).cuda()
    got = ctx.filter_bessel(dev, synth.QUANTUM, cutoff=case["cutoff"], sampling_freq=case["second"]).cpu().numpy()
    ref = NPZ[case["name"] + "/filtered"]
    assert got.dtype == np.float64 and got.shape == ref.shape
    assert np.max(np.abs(got - ref)) <= TOL * np.max(np.abs(ref))
    # and against the oracle on the same input (same tolerance)
    assert np.max(np.abs(got - oracle.bessel_filtfilt(k * synth.QUANTUM, case["cutoff"], case["second"]))) <= TOL * np.max(np.abs(ref))


@pytest.mark.gpu
def test_filter_large_trace_against_oracle():
    """2e7 samples (4 883 chunks: the carry pass matters), slow and fast cutoffs."""
    import torch
    from pypore_amd import engine
    ctx = engine.context(0)
    k = synth.random_dwell_counts(20_000_000, 9, 1000, 20000)
    dev = torch.from_numpy(k.astype(np.int16)).cuda()
    for cutoff in (2000., 5.):
        got = ctx.filter_bessel(dev, synth.QUANTUM, cutoff=cutoff, sampling_freq=1e5).cpu().numpy()
        ref = oracle.bessel_filtfilt(k * synth.QUANTUM, cutoff, 1e5)
        assert np.max(np.abs(got - ref)) <= 1e-10 * np.max(np.abs(ref))


@pytest.mark.gpu
def test_event_filter_contract_and_parse_of_filtered_event():
    """Event.filter replaces current by the float64 result and records the filter (DataTypes.py:270-274).  Event.parse
    of a filtered event (centred, rounded to a 2**-18 pA grid, see DataTypes.Event.parse) finds the boundaries the
    compiled reference finds on the unrounded float64 current (golden), and segments expose views / statistics of the
    unrounded current."""
    from pypore_amd.DataTypes import Event, File
    from pypore_amd.parsers import SpeedyStatSplit
    for case in MAN["cases"]:
        if case["n"] < 1000:
            continue
        x = _counts(case["gen"]).astype(np.float64) * synth.QUANTUM
        f = File(current=x, timestep=1000. / case["second"])
        ev = Event(current=x.copy(), start=0., end=len(x) / f.second, duration=len(x) / f.second, second=f.second, file=f)
        ev.filter(cutoff=case["cutoff"])                   # order=1
        assert ev.filtered and ev.filter_order == 1 and ev.filter_cutoff == case["cutoff"]
        ref_y = NPZ[case["name"] + "/filtered"]
        assert ev.current.dtype == np.float64
        assert np.max(np.abs(ev.current - ref_y)) <= TOL * np.max(np.abs(ref_y))
        ev.parse(SpeedyStatSplit(prior_segments_per_second=10, sampling_freq=case["second"]))
        got = np.array([int(round(s.start * f.second)) for s in ev.segments[1:]])
        np.testing.assert_array_equal(got, NPZ[case["name"] + "/ref_bounds_on_filtered"])
        seg = ev.segments[len(ev.segments) // 2]
        a, b = int(round(seg.start * f.second)), int(round(seg.end * f.second))
        np.testing.assert_array_equal(seg.current, ev.current[a:b])
        assert seg.mean == pytest.approx(float(np.mean(ev.current[a:b])), rel=1e-12)
    with pytest.raises(ValueError):
        Event(current=x.copy(), second=f.second, file=f).filter(order=9)       # orders 1..8 run on the device


@pytest.mark.gpu
def test_filter_rejects_bad_arguments():
    import torch
    from pypore_amd import engine
    ctx = engine.context(0)
    dev = torch.zeros(100, dtype=torch.int16, device="cuda")
    with pytest.raises(ValueError):
        ctx.filter_bessel(dev[:6].contiguous(), 1.0)
    with pytest.raises(ValueError):
        ctx.filter_bessel(dev, 1.0, order=9)
    with pytest.raises(ValueError):
        ctx.filter_bessel(dev[:12].contiguous(), 1.0, order=3)                 # padlen = 3 * (order + 1) = 12
    with pytest.raises(ValueError):
        ctx.filter_bessel(dev, 1.0, cutoff=60000., sampling_freq=1e5)


@pytest.mark.gpu
@pytest.mark.parametrize("cutoff,n", [(2000., 3_000_001), (2000., 7), (2000., 3072 * 2 + 5), (700., 500_000), (5000., 100_000),
                                      (20000., 100_000), (10000., 4096 - 12), (1200., 2_000_000), (600., 50_000)])
def test_fused_filter_equals_three_pass_scan(cutoff, n):
    """Fast filters run both directions in one kernel over tiles with halos (the state forgets within the halo to
    2^-60); the result must agree with the exact three-pass scan to rounding, for every halo size the host can pick
    (at 100 kHz: 960 for 700 Hz, 576 for 1.2 kHz, 384 for 2 kHz, 192 for 5 kHz, 64 above; 600 Hz is too slow and takes
    the exact scan either way), tiles at the ends of the sequence and inputs shorter than a tile."""
    import torch
    from pypore_amd import engine
    ctx = engine.context(0)
    k = synth.random_dwell_counts(n, 17, 2, 4) if n < 100 else synth.random_dwell_counts(n, 17, 300, 5000)
    dev = torch.from_numpy(k.astype(np.int16)).cuda()
    try:
        ctx.set_option("filter_fused", 0)
        exact = ctx.filter_bessel(dev, synth.QUANTUM, cutoff=cutoff, sampling_freq=1e5).cpu().numpy()
        ctx.set_option("filter_fused", 1)
        fused = ctx.filter_bessel(dev, synth.QUANTUM, cutoff=cutoff, sampling_freq=1e5).cpu().numpy()
    finally:
        ctx.set_option("filter_fused", 1)
    scale = np.max(np.abs(exact))
    assert np.max(np.abs(fused - exact)) <= 2e-14 * scale
    ref = oracle.bessel_filtfilt(k * synth.QUANTUM, cutoff, 1e5)
    assert np.max(np.abs(fused - ref)) <= 1e-10 * np.max(np.abs(ref))


# ---- orders 2..4 -----------------------------------------------------------------------------
MAN_O = json.load(open(os.path.join(HERE, "golden", "manifest_filter_order.json")))
NPZ_O = np.load(os.path.join(HERE, "golden", "golden_filter_order.npz"))


def _x_order(case):
    g = case["gen"]
    return synth.random_dwell_counts(g["n"], g["seed"], g["lo"], g["hi"])


@pytest.mark.parametrize("case", MAN_O["cases"], ids=[c["name"] for c in MAN_O["cases"]])
def test_oracle_filter_orders_match_scipy_golden(case):
    """Design (poles of the phase-normalised prototype, bilinear transform) and filtfilt of the oracle against scipy's
    (b, a) and output.  Tolerance per case: a direct-form filter of order n amplifies last-bit coefficient differences
    (tests/golden/make_golden_filter_order.py records how much)."""
    b, a = oracle.bessel_ba(case["order"], case["cutoff"] / (case["second"] / 2.))
    np.testing.assert_allclose(b, NPZ_O[case["name"] + "/b"], rtol=1e-14)
    np.testing.assert_allclose(a, NPZ_O[case["name"] + "/a"], rtol=0, atol=1e-13)
    x = _x_order(case).astype(np.float64) * synth.QUANTUM
    ref = NPZ_O[case["name"] + "/filtered"]
    got = oracle.bessel_filtfilt(x, case["cutoff"], case["second"], case["order"])
    assert np.max(np.abs(got - ref)) <= case["tol"] * np.max(np.abs(ref))


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["f32", "i16"])
@pytest.mark.parametrize("case", MAN_O["cases"], ids=[c["name"] for c in MAN_O["cases"]])
def test_filter_kernel_orders_match_scipy_golden(case, dtype):
    import torch
    from pypore_amd import engine
    ctx = engine.context(0)
    k = _x_order(case)
    dev = torch.from_numpy(k.astype(np.int16)).cuda() if dtype == "i16" else \
        torch.from_numpy((k * synth.QUANTUM).astype(np.float32)).cuda()
    got = ctx.filter_bessel(dev, synth.QUANTUM, cutoff=case["cutoff"], sampling_freq=case["second"], order=case["order"]).cpu().numpy()
    ref = NPZ_O[case["name"] + "/filtered"]
    assert got.dtype == np.float64 and got.shape == ref.shape
    assert np.max(np.abs(got - ref)) <= case["tol"] * np.max(np.abs(ref))
    orc = oracle.bessel_filtfilt(k * synth.QUANTUM, case["cutoff"], case["second"], case["order"])
    assert np.max(np.abs(got - orc)) <= case["tol"] * np.max(np.abs(ref))


@pytest.mark.gpu
def test_filter_order3_large_trace_and_event_api():
    """5e6 samples (hundreds of segments with halos) against the oracle; Event.filter(order=3) through the class."""
    import torch
    from pypore_amd import engine
    from pypore_amd.DataTypes import Event, File
    ctx = engine.context(0)
    k = synth.random_dwell_counts(5_000_000, 19, 1000, 20000)
    dev = torch.from_numpy(k.astype(np.int16)).cuda()
    got = ctx.filter_bessel(dev, synth.QUANTUM, cutoff=2000., sampling_freq=1e5, order=3).cpu().numpy()
    ref = oracle.bessel_filtfilt(k * synth.QUANTUM, 2000., 1e5, 3)
    assert np.max(np.abs(got - ref)) <= 1e-9 * np.max(np.abs(ref))
    x = synth.counts_to_pa(k[:200000], np.float64)
    f = File(current=x, timestep=0.01)
    ev = Event(current=x.copy(), start=0., end=2., duration=2., second=f.second, file=f)
    ev.filter(order=3, cutoff=2000.)
    assert ev.filtered and ev.filter_order == 3
    assert np.max(np.abs(ev.current - oracle.bessel_filtfilt(x, 2000., 1e5, 3))) <= 1e-9 * np.max(np.abs(x))


@pytest.mark.gpu
def test_requantise_matches_the_host_rounding_of_event_parse():
    """ps_requantise (filtered current -> centred fp32 on the finest power-of-two grid with counts below 2**22, on the
    device) against DataTypes.Event._on_fine_grid (the same on the host with numpy): same step, same rounded values --
    the two means may differ in the last bits, which can move the centre by one grid step at most (a constant shift)."""
    import torch
    from pypore_amd import engine
    from pypore_amd.DataTypes import Event, File
    ctx = engine.context(0)
    rng = np.random.default_rng(3)
    for n, scale, offset in ((13, 1.0, 0.0), (5000, 40.0, 55.0), (1_000_003, 3.0, -20.0), (262144, 1e-3, 1e3), (70000, 250.0, 0.0)):
        x = offset + scale * np.cumsum(rng.standard_normal(n)) / np.sqrt(n) + 0.01 * scale * rng.standard_normal(n)
        f = File(current=x, timestep=0.01)
        rounded, step, _ = Event(current=x, second=f.second, file=f)._on_fine_grid()
        z, centre, dstep = ctx.requantise(torch.from_numpy(x).cuda())
        z = z.cpu().numpy().astype(np.float64)
        assert dstep == step and np.abs(z).max() < 2 ** 22 * step
        np.testing.assert_array_equal(np.rint(z / step) * step, z)             # on the grid
        shift = np.unique(np.rint((rounded - z) / step))
        assert shift.size == 1 and abs(shift[0]) <= 1
        assert abs(centre - np.mean(x)) <= step
    const, c0, s0 = ctx.requantise(torch.full((100,), 7.25, dtype=torch.float64, device="cuda"))
    assert s0 == 1.0 and c0 == 7.0 and float(const.abs().max()) == 0.0        # no spread: unit grid, the mean rounded
    with pytest.raises(ValueError):
        ctx.requantise(torch.tensor([1.0, float("nan"), 2.0], dtype=torch.float64, device="cuda"))


# ---- orders 5..8 and float64 input on no grid -------------------------------------------------
MAN_H = json.load(open(os.path.join(HERE, "golden", "manifest_filter_hi.json")))
NPZ_H = np.load(os.path.join(HERE, "golden", "golden_filter_hi.npz"))


def _x_hi(case):
    """(first input as the generator gives it, float64 input of the LAST filter of the chain)"""
    g = case["gen"]
    if g["kind"] == "grid":
        x0 = synth.random_dwell_counts(g["n"], g["seed"], g["lo"], g["hi"]).astype(np.float64) * synth.QUANTUM
    else:
        x0 = synth.offgrid_trace(g["n"], g["seed"], sigma=g["sigma"])
    return x0, (NPZ_H[case["name"] + "/input"] if len(case["chain"]) > 1 else x0)


@pytest.mark.parametrize("case", MAN_H["cases"], ids=[c["name"] for c in MAN_H["cases"]])
def test_oracle_filter_hi_matches_scipy_golden(case):
    x0, xin = _x_hi(case)
    order, cutoff = case["chain"][-1]
    ref = NPZ_H[case["name"] + "/filtered"]
    got = oracle.bessel_filtfilt(xin, cutoff, case["second"], int(order))
    assert np.max(np.abs(got - ref)) <= case["tol"] * np.max(np.abs(ref))


@pytest.mark.gpu
@pytest.mark.parametrize("case", MAN_H["cases"], ids=[c["name"] for c in MAN_H["cases"]])
def test_filter_kernel_hi_matches_scipy_golden(case):
    """the kernels on what the LAST filter of the chain receives: counts (fp32 on the grid, int16) when the chain has one
    filter and the generator is on the grid, float64 (PS_DTYPE_F64) otherwise"""
    import torch
    from pypore_amd import engine
    ctx = engine.context(0)
    x0, xin = _x_hi(case)
    order, cutoff = case["chain"][-1]
    ref = NPZ_H[case["name"] + "/filtered"]
    inputs = [torch.from_numpy(np.ascontiguousarray(xin, dtype=np.float64)).cuda()]
    if len(case["chain"]) == 1 and case["gen"]["kind"] == "grid":
        k = np.rint(x0 / synth.QUANTUM)
        inputs += [torch.from_numpy(k.astype(np.int16)).cuda(), torch.from_numpy(x0.astype(np.float32)).cuda()]
    for dev in inputs:
        got = ctx.filter_bessel(dev, synth.QUANTUM, cutoff=cutoff, sampling_freq=case["second"], order=int(order)).cpu().numpy()
        assert got.dtype == np.float64 and got.shape == ref.shape
        assert np.max(np.abs(got - ref)) <= case["tol"] * np.max(np.abs(ref)), (case["name"], dev.dtype)


@pytest.mark.gpu
def test_event_filter_twice_and_offgrid_like_the_reference():
    """DataTypes.py:258-274 filters whatever self.current holds: a second filter of a filtered event (Experiment.parse
    twice on the same File objects), float64 data on no grid -- through the Event API, against scipy's chain."""
    from pypore_amd.DataTypes import Event, File
    from pypore_amd.parsers import SpeedyStatSplit
    for case in MAN_H["cases"]:
        if case["n"] < 1000:
            continue
        x0, _ = _x_hi(case)
        f = File(current=x0, timestep=1000. / case["second"])
        ev = Event(current=x0.copy(), start=0., end=len(x0) / f.second, duration=len(x0) / f.second, second=f.second, file=f)
        for order, cutoff in case["chain"]:
            ev.filter(order=int(order), cutoff=cutoff)
        ref = NPZ_H[case["name"] + "/filtered"]
        # (the first filter of a chain ran on the device too: its own deviation from scipy feeds the second)
        assert np.max(np.abs(ev.current - ref)) <= max(case["tol"], 1e-11) * 4 * np.max(np.abs(ref)), case["name"]
        assert ev.filtered and ev.filter_order == int(case["chain"][-1][0])
    # ... and the twice-filtered event still parses (re-quantised route), boundaries equal to the oracle on the same current
    case = [c for c in MAN_H["cases"] if c["name"] == "twice_O1_2k"][0]
    x0, _ = _x_hi(case)
    f = File(current=x0, timestep=0.01)
    ev = Event(current=x0.copy(), start=0., end=len(x0) / f.second, duration=len(x0) / f.second, second=f.second, file=f)
    ev.filter(); ev.filter()
    ev.parse(SpeedyStatSplit(prior_segments_per_second=10, sampling_freq=case["second"]))
    got = np.array([int(round(s.start * f.second)) for s in ev.segments[1:]])
    np.testing.assert_array_equal(got, oracle.parse(ev._on_fine_grid()[0], prior_segments_per_second=10.))


@pytest.mark.gpu
def test_experiment_parse_twice_on_the_same_files(tmp_path):
    """VERDICT r3 missing #3: the second Experiment.parse re-filters / re-parses where the reference would"""
    from pypore_amd import abf
    from pypore_amd.DataTypes import Experiment
    counts, _ = synth.file_trace_counts(1_200_000, 31)
    path = str(tmp_path / "twice.abf")
    abf.write_abf(path, counts.astype(np.int16))
    from pypore_amd.DataTypes import File
    f = File(path)
    exp = Experiment([f])
    exp.parse(verbose=False)
    first = [[(s.start, s.end) for s in ev.segments] for ev in f.events]
    exp.parse(verbose=False)                     # the same File object again: events are detected, filtered and parsed anew
    second = [[(s.start, s.end) for s in ev.segments] for ev in f.events]
    assert first == second and len(first) >= 1
    assert len(exp.files) == 2                   # (like the reference, :988: every parse appends its files)
    # filtering the already filtered events once more goes through as well (float64 route), and they still parse
    for ev in f.events:
        before = np.array(ev.current)
        ev.filter()
        assert ev.filtered and ev.current.shape == before.shape and not np.array_equal(ev.current, before)
        ev.parse()
        assert len(ev.segments) >= 1
