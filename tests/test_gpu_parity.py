"""Parity tests proper (-m gpu): the HIP path, called through the C ABI, against the golden
vectors recorded from the reference and against the CPU oracle on seeded inputs."""
import hashlib

import os

import numpy as np
import pytest

import oracle
from golden_util import case_ids, cases, input_counts, input_pa, npz, offgrid_cases, offgrid_npz
from pypore_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from pypore_amd import engine
    return engine.context(0)


def _bounds(segs):
    return np.array([s.start for s in segs[1:]], dtype=np.int32)


@pytest.mark.parametrize("case", cases("parse"), ids=case_ids("parse"))
def test_parse_matches_reference_golden(case, ctx):
    from pypore_amd.parsers import SpeedyStatSplit
    x = input_pa(case)                                  # float64, like the reference's input
    p = SpeedyStatSplit(quantum=synth.QUANTUM, **case["params"])
    segs = p.parse(x)
    # segment boundary indices: bit-exact
    np.testing.assert_array_equal(_bounds(segs), npz()[case["name"] + "/bounds"])
    assert [s.end for s in segs] == list(_bounds(segs)) + [len(x)]
    assert all(s.duration == s.end - s.start for s in segs)
    if case["name"] + "/mean" in npz() and len(x) > 0:
        # north_star tolerance: per-segment mean/std within 1e-5 relative
        np.testing.assert_allclose([s.mean for s in segs], npz()[case["name"] + "/mean"], rtol=1e-5)
        np.testing.assert_allclose([s.std for s in segs], npz()[case["name"] + "/std"], rtol=1e-5, atol=1e-9)
        np.testing.assert_array_equal([s.min for s in segs], npz()[case["name"] + "/min"])
        np.testing.assert_array_equal([s.max for s in segs], npz()[case["name"] + "/max"])
        # .current are views of the caller's array (cparsers.pyx:115)
        assert segs[0].current.base is x or segs[0].current.base is x.base or np.shares_memory(segs[0].current, x)


@pytest.mark.parametrize("case", cases("parse")[:12], ids=case_ids("parse")[:12])
def test_parse_int16_and_float32_inputs(case, ctx):
    from pypore_amd.parsers import SpeedyStatSplit
    counts = input_counts(case)
    g = npz()[case["name"] + "/bounds"]
    p = SpeedyStatSplit(quantum=synth.QUANTUM, **case["params"])
    np.testing.assert_array_equal(_bounds(p.parse(counts.astype(np.int16))), g)
    np.testing.assert_array_equal(_bounds(p.parse(synth.counts_to_pa(counts, np.float32))), g)
    # automatic grid detection
    p2 = SpeedyStatSplit(**case["params"])
    np.testing.assert_array_equal(_bounds(p2.parse(synth.counts_to_pa(counts, np.float64))), g)


@pytest.mark.parametrize("stitch_host", [0, 1])
@pytest.mark.parametrize("tile,halo", [(20000, 10000), (50000, 20000), (100000, 40000), (30000, 1), (7000, 1)])
@pytest.mark.parametrize("name", ["G8_full", "G9_rd_2M", "G9_rd_short_dwell", "G9_rd_long_dwell", "G9_cutoff",
                                  "G4_forced_flat", "G4_forced_mixed", "G4_big_window", "G4_small_windows"])
def test_tiled_spines_stitch_to_the_same_result(name, tile, halo, stitch_host, ctx):
    """Speculative tiles + stitching must not change a single boundary: device stitch (bridged seams,
    assemble kernel; falls back to the host when a bridge gives up) and host stitch (halo tiles,
    seam repairs when the halo is too short)."""
    from pypore_amd.parsers import SpeedyStatSplit
    (case,) = [c for c in cases("parse") if c["name"] == name]
    ctx.set_tiling(tile, halo)
    ctx.set_option("stitch_host", stitch_host)
    try:
        segs = SpeedyStatSplit(quantum=synth.QUANTUM, **case["params"]).parse(input_pa(case, np.float32))
    finally:
        ctx.set_tiling(0, 0)
        ctx.set_option("stitch_host", 0)
    np.testing.assert_array_equal(_bounds(segs), npz()[name + "/bounds"])


def test_verify_mode_screen_agrees_with_exact_scan(ctx):
    """mode 2: every window is scanned by the fp32 screen AND the exact fp64 path; any disagreement
    raises.  Run on the config-2 batch shape and a 2e6-sample trace."""
    from pypore_amd.parsers import SpeedyStatSplit
    ctx.set_option("mode", 2)
    try:
        p = SpeedyStatSplit(prior_segments_per_second=10., quantum=synth.QUANTUM)
        out = p.parse_batch([synth.config2_event(ev, dtype=np.float32) for ev in range(200, 232)])
        assert sum(len(s) for s in out) >= 32 * 5
        (case,) = [c for c in cases("parse") if c["name"] == "G9_rd_2M"]
        segs = p.parse(input_pa(case, np.float32))
        np.testing.assert_array_equal(_bounds(segs), npz()["G9_rd_2M/bounds"])
    finally:
        ctx.set_option("mode", int(os.environ.get("PORESEG_MODE", "0")))      # (what the context started with: tools/gpu_validate.sh)


@pytest.mark.parametrize("case", cases("score_window"), ids=case_ids("score_window"))
def test_per_candidate_gains(case, ctx):
    from pypore_amd.cparsers import FastStatSplit
    x = input_pa(case)
    f = FastStatSplit(quantum=synth.QUANTUM, **case["params"])
    s = np.array(f.score_samples(x, no_split=True))
    g = npz()[case["name"] + "/scores"]
    assert s.shape == g.shape
    np.testing.assert_array_equal(s == 0, g == 0)
    # same operation order in fp64; only the device log differs from glibc's by <= 1 ulp per call
    np.testing.assert_allclose(s, g, rtol=0, atol=1e-7)
    assert int(np.argmax(s)) == int(np.argmax(g))


@pytest.mark.parametrize("case", offgrid_cases("score_samples"), ids=lambda c: c["name"])
def test_score_samples_full_list(case, ctx):
    """score_samples(current) (cparsers.pyx:205-275): one dense gain array per window scan, in the recursion's order --
    count, order, zero pattern and every value (1e-7 absolute: same fp64 operation order, the device log differs from
    glibc's by <= 1 ulp) against the list recorded from the compiled reference."""
    from pypore_amd.cparsers import FastStatSplit
    gen = case["gen"]
    x = synth.config1() if gen["kind"] == "config1" else synth.counts_to_pa(
        synth.random_dwell_counts(gen["n"], gen["seed"], gen["lo"], gen["hi"]), np.float64)
    f = FastStatSplit(quantum=synth.QUANTUM, **case["params"])
    scans = f.score_samples(x)
    assert len(scans) == case["n_scans"]
    z = offgrid_npz()
    for k, (sc, (a, b)) in enumerate(zip(scans, case["spans"])):
        sc = np.asarray(sc)
        assert sc.shape == (len(x),)
        ref = np.zeros(len(x))
        ref[a:b] = z["%s/scan%03d" % (case["name"], k)]
        np.testing.assert_array_equal(sc == 0, ref == 0, err_msg="scan %d" % k)
        np.testing.assert_allclose(sc, ref, rtol=0, atol=1e-7, err_msg="scan %d" % k)


def test_exact_tie_first_maximum_wins_and_is_counted(ctx):
    """A palindromic noisy event A | B | A: the gains at its two steps are exactly equal in the reference's arithmetic; the
    first one wins (cparsers.pyx:175-177).  The device decides it among fp64 contenders the same way, reports the window
    in the near-tie counter (ps_get_timings counters[11]) and the Python layer warns."""
    import warnings
    from pypore_amd import engine
    from pypore_amd.parsers import SpeedyStatSplit
    (case,) = offgrid_cases("tie")
    assert case["gain_at_a"] == case["gain_at_mirror"] and case["argmax"] == case["gen"]["a"]     # the reference's own numbers
    x = synth.counts_to_pa(synth.palindrome_counts(**case["gen"]), np.float64)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        segs = SpeedyStatSplit(quantum=synth.QUANTUM, **case["params"]).parse(x)
    got = np.array([s_.start for s_ in segs[1:]], dtype=np.int32)
    np.testing.assert_array_equal(got, offgrid_npz()[case["name"] + "/bounds"])
    np.testing.assert_array_equal(got, oracle.parse(x, **case["params"]))
    import os
    if os.environ.get("PORESEG_SCAN_BS", "1") != "0":       # (the LDS-window fallback scans whole windows in fp64 and keeps no such count)
        assert engine.context().near_ties() >= 1
        assert any(issubclass(i.category, engine.NearTieWarning) for i in w)
    # an ordinary trace reports none
    SpeedyStatSplit(quantum=synth.QUANTUM, prior_segments_per_second=10.).parse(synth.config1())
    # (-1: not counted -- the suite is running with the LDS-window kernels switched on, tools/gpu_validate.sh)
    assert engine.context().near_ties() == (-1 if os.environ.get("PORESEG_SCAN_BS", "1") == "0" else 0)


@pytest.mark.parametrize("case", offgrid_cases("parse_offgrid"), ids=lambda c: c["name"])
def test_offgrid_float64_requantise_route(case, ctx):
    """Float64 input on no ADC grid (the reference takes any double buffer, cparsers.pyx:53,103-111): the default raises
    ValueError (nothing is rounded silently); off_grid="requantise" rounds on the device to a 2**-k grid with counts below
    2**22 and segments on the 64-bit digest -- every boundary the compiled reference finds on the UNROUNDED values, and the
    segments' statistics (taken from the original values) to 1e-12."""
    from pypore_amd.parsers import SpeedyStatSplit
    x = synth.offgrid_trace(**case["gen"])
    assert [repr(float(v)) for v in x[:4]] == case["x_head"] and repr(float(np.sum(x))) == case["x_sum"]
    if case["n"] <= 100000:
        with pytest.raises(ValueError):
            SpeedyStatSplit(**case["params"]).parse(x)
    segs = SpeedyStatSplit(off_grid="requantise", **case["params"]).parse(x)
    got = np.array([s.start for s in segs[1:]], dtype=np.int32)
    ref = offgrid_npz()[case["name"] + "/bounds"]
    np.testing.assert_array_equal(got, ref)
    assert segs[3 % len(segs)].current.base is x or np.shares_memory(segs[3 % len(segs)].current, x)      # views of the caller's array
    np.testing.assert_allclose([s.mean for s in segs], offgrid_npz()[case["name"] + "/mean"], rtol=1e-12)
    np.testing.assert_allclose([s.std for s in segs], offgrid_npz()[case["name"] + "/std"], rtol=1e-9)


@pytest.mark.parametrize("case", cases("best_single_split"), ids=case_ids("best_single_split"))
def test_best_single_split(case, ctx):
    from pypore_amd.parsers import SpeedyStatSplit
    g, i = SpeedyStatSplit(prior_segments_per_second=10., quantum=synth.QUANTUM).best_single_split(input_pa(case))
    assert i == case["index"]
    assert g == pytest.approx(float(case["gain"]), rel=1e-12, abs=1e-9)


@pytest.mark.parametrize("seed", range(6))
def test_random_parameters_against_oracle(seed, ctx):
    """Seeded random traces and parameter sets: HIP path == CPU oracle, bit for bit."""
    from pypore_amd.parsers import SpeedyStatSplit
    rng = np.random.RandomState(seed)
    n = int(rng.randint(20000, 400000))
    lo = int(rng.randint(50, 2000))
    hi = lo + int(rng.randint(100, 30000))
    mw = int(rng.choice([5, 20, 100, 250]))
    W = int(max(2 * mw, rng.choice([400, 1000, 4000, 10000, 25000])))
    maxw = int(rng.choice([W, 3 * W, 50000, 1000000]))
    params = dict(min_width=mw, max_width=max(maxw, mw), window_width=W,
                  prior_segments_per_second=float(rng.choice([1., 10., 100.])))
    counts = synth.random_dwell_counts(n, 1000 + seed, lo, hi)
    x = synth.counts_to_pa(counts, np.float64)
    ref = oracle.parse(x, **params)
    got = _bounds(SpeedyStatSplit(quantum=synth.QUANTUM, **params).parse(x))
    np.testing.assert_array_equal(got, ref)


def test_batch_of_events_config2_shape(ctx):
    """BASELINE config 2 shape (reduced count): a batch of 50k-sample events in one call."""
    from pypore_amd.parsers import SpeedyStatSplit
    evs = [synth.config2_event(ev, dtype=np.float32) for ev in range(100, 148)]
    evs.append(np.zeros(0, dtype=np.float32))            # empty event
    evs.append(synth.config2_event(7, n=150, dtype=np.float32))   # shorter than 2*min_width
    p = SpeedyStatSplit(prior_segments_per_second=10., quantum=synth.QUANTUM)
    out = p.parse_batch(evs)
    assert len(out) == len(evs)
    for x, segs in zip(evs, out):
        ref = oracle.parse(x.astype(np.float64), prior_segments_per_second=10.)
        np.testing.assert_array_equal(_bounds(segs), ref)
        st = oracle.segment_stats(x.astype(np.float64), ref) if len(x) else None
        if st is not None:
            np.testing.assert_allclose([s.mean for s in segs], st[:, 0], rtol=1e-5)
            np.testing.assert_allclose([s.std for s in segs], st[:, 1], rtol=1e-5, atol=1e-9)


def test_device_generator_matches_numpy(ctx):
    import torch
    n = 300000
    d = synth.dwell_table(77, n)
    ends = np.cumsum(d)
    lv = synth.LEVEL_COUNTS[np.arange(len(d)) % 5].astype(np.int32)
    ref = synth.random_dwell_counts(n, 77)
    t = ctx.synth_trace(n, 77, ends, lv, dtype=torch.float32)
    np.testing.assert_array_equal(t.cpu().numpy(), synth.counts_to_pa(ref, np.float32))
    t16 = ctx.synth_trace(n, 77, ends, lv, dtype=torch.int16)
    np.testing.assert_array_equal(t16.cpu().numpy().astype(np.int32), ref)


def test_off_grid_and_bad_dtype_fail_loudly(ctx):
    from pypore_amd.parsers import SpeedyStatSplit
    x = synth.config1(np.float64).copy()
    x[1234] += 1e-3
    with pytest.raises(ValueError):
        SpeedyStatSplit(prior_segments_per_second=10., quantum=synth.QUANTUM).parse(x.astype(np.float32))
    with pytest.raises(ValueError):
        SpeedyStatSplit(prior_segments_per_second=10.).parse(np.arange(1000, dtype=np.int32))
    with pytest.raises(AssertionError):
        SpeedyStatSplit(min_width=100, window_width=100).parse(x)


def test_event_parse_call_contract(ctx):
    """Event.parse(parser=SpeedyStatSplit(...)) / File.parse() as user code calls them."""
    from pypore_amd.DataTypes import File
    from pypore_amd.parsers import SpeedyStatSplit, lambda_event_parser
    z = npz()
    x = z["G6_events/input"].astype(np.float64) * synth.QUANTUM
    f = File(current=x, timestep=0.01)                       # 100 kHz
    f.parse(parser=lambda_event_parser(threshold=90))          # default rules: detection on the GPU
    assert [int(round(e.start * f.second)) for e in f.events] == list(z["G6_events/starts"])
    for k, ev in enumerate(f.events):
        ev.parse(parser=SpeedyStatSplit(prior_segments_per_second=10., quantum=synth.QUANTUM))
        b = np.array([int(round(s.start * f.second)) for s in ev.segments[1:]])
        np.testing.assert_array_equal(b, z["G6_events/ev%d_bounds" % k])
        assert ev.segments[0].event is ev and ev.state_parser is not None
    f.parse_events(SpeedyStatSplit(prior_segments_per_second=10., quantum=synth.QUANTUM))
    for k, ev in enumerate(f.events):
        b = np.array([int(round(s.start * f.second)) for s in ev.segments[1:]])
        np.testing.assert_array_equal(b, z["G6_events/ev%d_bounds" % k])


@pytest.mark.parametrize("mode", [0, 2], ids=["default", "verify"])
def test_1e8_trace_digest(ctx, mode):
    """BASELINE full size: the 10^8-sample trace, generated in HBM, against the digest recorded
    from the reference (count + SHA-256 of all boundaries + first/last 32).  mode 2 (verify): every one of its ~37 000
    windows is decided by the screen AND by the exact fp64 scan, and the call fails on any disagreement."""
    import torch
    from pypore_amd import _lib
    (case,) = cases("parse_digest")
    n = case["n"]
    d = synth.dwell_table(case["gen"]["seed"], n)
    ends = np.cumsum(d)
    lv = synth.LEVEL_COUNTS[np.arange(len(d)) % 5].astype(np.int32)
    t = ctx.synth_trace(n, case["gen"]["seed"], ends, lv, dtype=torch.float32)
    params = _lib.split_params(**case["params"])
    ctx.set_option("mode", mode)
    try:
        bounds, boff, _ = ctx.segment_batch(t, np.array([0, n]), params, synth.QUANTUM, want_stats=False)
        # (the counter of whole-window fp64 scans belongs to the block-sum scan of the device-stitch pipeline: the default)
        if mode == 2 and not (os.environ.get("PORESEG_SCAN_BS") or os.environ.get("PORESEG_STITCH")):
            assert ctx.timings()["full_exact_scans"] >= ctx.timings()["windows"] > 30000
    finally:
        ctx.set_option("mode", int(os.environ.get("PORESEG_MODE", "0")))      # (what the context started with: tools/gpu_validate.sh)
    b = bounds.cpu().numpy().astype(np.int32)
    assert len(b) == case["n_bounds"]
    np.testing.assert_array_equal(b[:32], npz()["G7_1e8/first32"])
    np.testing.assert_array_equal(b[-32:], npz()["G7_1e8/last32"])
    assert hashlib.sha256(b.tobytes()).hexdigest() == case["sha256"]
    # size-independent properties: sorted, strictly increasing, >= min_width apart, inside the trace
    assert np.all(np.diff(b) >= case["params"]["min_width"]) and b[0] >= 100 and b[-1] <= n - 100


def test_spine_flags_and_sharded_trace_on_device(ctx):
    """ps_segment_batch_ex's spine flags equal the oracle's, and a trace cut into 4 pieces that are
    segmented independently on the GPU stitches back to the whole-trace golden (config 5 logic)."""
    import torch
    from pypore_amd import _lib
    from pypore_amd.dist import shard_ranges, stitch_pieces
    (case,) = [c for c in cases("parse") if c["name"] == "G9_rd_2M"]
    x = input_pa(case, np.float32)
    n = len(x)
    params = _lib.split_params(**case["params"])
    t = torch.from_numpy(x).cuda()
    b, _, _, f = ctx.segment_batch(t, np.array([0, n]), params, synth.QUANTUM, want_stats=False, want_spine=True)
    rb, rf = oracle.parse_flags(x.astype(np.float64), **case["params"])
    np.testing.assert_array_equal(b.cpu().numpy(), rb)
    np.testing.assert_array_equal(f.cpu().numpy(), rf)
    pieces = []
    for lo, hi in shard_ranges(n, 4, 80000):
        pb, _, _, pf = ctx.segment_batch(t[lo:hi].contiguous(), np.array([0, hi - lo]), params, synth.QUANTUM,
                                         want_stats=False, want_spine=True)
        pieces.append((lo, hi, pb.cpu().numpy(), pf.cpu().numpy()))
    np.testing.assert_array_equal(stitch_pieces(pieces, n, 10000, 100), npz()["G9_rd_2M/bounds"])


@pytest.mark.parametrize("case", ["long_dwell", "flat"])
def test_sharded_trace_seam_repair_on_device(ctx, case):
    """pieces whose seams find no common spine anchor inside the halo (dwells of 1e5..1e6 samples; a flat stretch
    of 3e6 samples) are joined by re-running the chain from the last trusted upstream anchor over a longer stretch.
    8 pieces on one GPU == the whole trace on the GPU == the oracle."""
    import torch
    from pypore_amd import _lib
    from pypore_amd.dist import shard_ranges, stitch_pieces
    if case == "long_dwell":
        counts = synth.random_dwell_counts(6_000_000, 64, 100000, 1000000)
    else:
        counts = np.concatenate([synth.random_dwell_counts(700_000, 71), synth.LEVEL_COUNTS[2] + synth.noise_counts(72, 0, 3_000_000),
                                 synth.random_dwell_counts(800_000, 73)])
    x = synth.counts_to_pa(counts, np.float32)
    n = len(x)
    params = _lib.split_params(min_width=100, max_width=1000000, window_width=10000, prior_segments_per_second=10., sampling_freq=1e5)
    t = torch.from_numpy(x).cuda()

    def seg(lo, hi):
        pb, _, _, pf = ctx.segment_batch(t[lo:hi].contiguous(), np.array([0, hi - lo]), params, synth.QUANTUM,
                                         want_stats=False, want_spine=True)
        return pb.cpu().numpy(), pf.cpu().numpy()

    whole, _ = seg(0, n)
    ref = oracle.parse(x.astype(np.float64), prior_segments_per_second=10.)
    np.testing.assert_array_equal(whole, ref)
    calls = []

    def repair(r, lo, hi):
        calls.append((r, lo, hi))
        return seg(lo, hi)

    pieces = [(lo, hi) + seg(lo, hi) for lo, hi in shard_ranges(n, 8, 80000)]
    got = stitch_pieces(pieces, n, 10000, 100, repair=repair, halo=80000)
    np.testing.assert_array_equal(got, ref)
    assert calls


def test_counts_beyond_int16_take_the_exact_path(ctx):
    """LDS-window scan: quantum 2^-10 turns 50 pA into 51200 counts, the window does not fit the int16 LDS
    image and is scanned straight from HBM -- by the wide-count screen (fp64 sums, fp32 logs) where that decides,
    by the exact fp64 path otherwise, and by the exact path alone in mode 1 -- same boundaries each time.  The
    block-sum scan (default) centres on the first sample and handles the same input without any fallback."""
    from pypore_amd.parsers import SpeedyStatSplit
    x = synth.config2_event(5)
    ref = oracle.parse(x, prior_segments_per_second=10.)
    segs = SpeedyStatSplit(prior_segments_per_second=10., quantum=2.0 ** -10).parse(x)
    np.testing.assert_array_equal(_bounds(segs), ref)
    ctx.set_option("scan_bs", 0)
    try:
        segs = SpeedyStatSplit(prior_segments_per_second=10., quantum=2.0 ** -10).parse(x)
        np.testing.assert_array_equal(_bounds(segs), ref)
        screened = ctx.timings()["exact_rescans"]
        ctx.set_option("mode", 1)                      # exact scans only
        segs = SpeedyStatSplit(prior_segments_per_second=10., quantum=2.0 ** -10).parse(x)
        np.testing.assert_array_equal(_bounds(segs), ref)
        # (under PORESEG_MODE=2 -- tools/gpu_validate.sh -- the first call already scanned every window exactly)
        assert ctx.timings()["exact_rescans"] > screened or os.environ.get("PORESEG_MODE", "0") != "0"
        ctx.set_option("mode", 2)                      # verify: screen and exact scan must agree on every window
        segs = SpeedyStatSplit(prior_segments_per_second=10., quantum=2.0 ** -10).parse(x)
        np.testing.assert_array_equal(_bounds(segs), ref)
    finally:
        ctx.set_option("scan_bs", 1)
        ctx.set_option("mode", int(os.environ.get("PORESEG_MODE", "0")))


def test_wide_range_counts_fall_back_from_block_sums(ctx):
    """Levels 1500 pA apart (48000 counts at 2^-5): |k - m| exceeds the uint32 block-sum range, K0 flags
    it and the call is redone with the LDS-window scan (whose int16 image overflows too -> exact path)."""
    from pypore_amd.parsers import SpeedyStatSplit
    c = synth.step_counts(60000, 7000, 91).astype(np.int64)
    c[20000:41000] += 48000
    x = synth.counts_to_pa(c, np.float64)
    ref = oracle.parse(x, prior_segments_per_second=10.)
    segs = SpeedyStatSplit(prior_segments_per_second=10., quantum=synth.QUANTUM).parse(x.astype(np.float32))
    np.testing.assert_array_equal(_bounds(segs), ref)


def _wide_digest_trace(n, seed):
    """Steps whose levels lie tens of thousands of counts apart on a grid 64 times finer than the ADC's: |k - m| runs to
    ~2^21, far beyond the 32-bit digest (23 000), inside the 64-bit one (2^23).  Exact in fp32 (|k| < 2^23)."""
    k = synth.random_dwell_counts(n, seed, 800, 12000).astype(np.int64) * 64
    k += np.random.default_rng(seed).integers(-31, 32, n)                    # use the fine grid's low bits too
    assert 23000 < np.abs(k - k[0]).max() < 2 ** 23 and np.abs(k).max() < 2 ** 23
    return k


@pytest.mark.parametrize("tile", [0, 60000])
def test_wide_digest_takes_counts_beyond_the_32bit_block_sums(tile, ctx):
    """K0 refuses such a trace on the 32-bit digest (ST_WIDE_RANGE) and the call is redone with the block-sum scan on the
    64-bit digest (timings: wide_redo 1) instead of the LDS-window kernels (wide_redo 2, option wide_bs = 0): same
    boundaries as the oracle on the same float64 values each way, also in verify mode, whole trace and tiled."""
    import torch
    from pypore_amd import _lib
    n = 1_200_000
    k = _wide_digest_trace(n, 5)
    q = synth.QUANTUM / 64
    x = k.astype(np.float64) * q
    kw = dict(min_width=100, max_width=1000000, window_width=10000, prior_segments_per_second=10., sampling_freq=1e5)
    ref = oracle.parse(x, **kw)
    assert len(ref) > 100
    t = torch.from_numpy(x.astype(np.float32)).cuda()
    off = np.array([0, n], dtype=np.int64)
    try:
        ctx.set_tiling(tile, 0)
        for wide_bs, mode, want in ((1, 0, 1), (1, 2, 1), (0, 0, 2)):
            ctx.set_option("wide_bs", wide_bs)
            ctx.set_option("mode", mode)
            for _ in range(2):                          # the second call starts on the route the first one ended on
                b, _, _ = ctx.segment_batch(t, off, _lib.split_params(**kw), q, want_stats=False)
                np.testing.assert_array_equal(b.cpu().numpy(), ref)
                assert ctx.timings()["wide_redo"] == want
    finally:
        ctx.set_option("wide_bs", 1)
        ctx.set_option("mode", int(os.environ.get("PORESEG_MODE", "0")))
        ctx.set_tiling(0, 0)


def test_wide_digest_batch_of_events_with_statistics(ctx):
    """Several events of different length in one call on the 64-bit digest (int16 input whose levels span more than
    23 000 counts, and fp32 input): boundaries and per-segment statistics equal the oracle's."""
    import torch
    from pypore_amd import _lib
    rng = np.random.default_rng(12)
    lens = [50000, 1234, 180001, 99, 70000]
    kw = dict(min_width=100, max_width=1000000, window_width=10000, prior_segments_per_second=10., sampling_freq=1e5)
    ks = []
    for i, ln in enumerate(lens):
        k = synth.random_dwell_counts(ln, 40 + i, 400, 6000).astype(np.int64)
        k = (k - 1500) * 58                                           # int16 range, steps up to ~27 000 counts
        ks.append(np.clip(k, -32768, 32767))
    allk = np.concatenate(ks)
    assert max(np.abs(k - k[0]).max() for k in ks) > 24000 and np.abs(allk).max() < 32767
    off = np.concatenate(([0], np.cumsum(lens))).astype(np.int64)
    q = synth.QUANTUM
    for dtype in (np.int16, np.float32):
        t = torch.from_numpy(allk.astype(dtype) if dtype == np.int16 else (allk * q).astype(np.float32)).cuda()
        b, boff, stats = ctx.segment_batch(t, off, _lib.split_params(**kw), q, want_stats=True)
        assert ctx.timings()["wide_redo"] == 1
        b = b.cpu().numpy(); st = stats.cpu().numpy() if hasattr(stats, "cpu") else np.asarray(stats)
        si = 0
        for e, ln in enumerate(lens):
            x = ks[e].astype(np.float64) * q
            ref = oracle.parse(x, **kw)
            np.testing.assert_array_equal(b[boff[e]:boff[e + 1]], ref)
            edges = np.concatenate(([0], ref, [ln])).astype(int)
            for a, z in zip(edges[:-1], edges[1:]):
                seg = x[a:z]
                if len(seg):
                    np.testing.assert_allclose(st[si][0], seg.mean(), rtol=1e-9, atol=1e-9)
                    np.testing.assert_allclose(st[si][1], seg.std(), rtol=1e-7, atol=1e-9)
                si += 1


@pytest.mark.parametrize("scan_bs", [0, 1])
def test_both_scan_implementations_on_goldens(scan_bs, ctx):
    """The LDS-window scan and the block-sum scan give the same boundaries (subset of the goldens)."""
    from pypore_amd.parsers import SpeedyStatSplit
    ctx.set_option("scan_bs", scan_bs)
    try:
        for name in ("G1_config1", "G2_dwell500", "G4_forced_mixed", "G4_small_windows", "G4_odd_window", "G4_big_window",
                     "G4_minwidth2", "G9_rd_2M", "G9_rd_short_dwell", "G9_cutoff"):
            (case,) = [c for c in cases("parse") if c["name"] == name]
            segs = SpeedyStatSplit(quantum=synth.QUANTUM, **case["params"]).parse(input_pa(case, np.float32))
            np.testing.assert_array_equal(_bounds(segs), npz()[name + "/bounds"])
    finally:
        ctx.set_option("scan_bs", 1)


def test_event_detector_kernel_matches_reference_parsers_py(ctx):
    """K3: ps_detect_events against the events the reference's own lambda_event_parser found (G6) and
    against the oracle on a longer file-shaped trace; fp32 and int16 inputs."""
    import torch
    z = npz()
    counts = z["G6_events/input"]
    for t, q in ((torch.from_numpy(synth.counts_to_pa(counts, np.float32)).cuda(), synth.QUANTUM),
                 (torch.from_numpy(counts.astype(np.int16)).cuda(), synth.QUANTUM)):
        st, ln = ctx.detect_events(t, q, threshold=90.0)
        np.testing.assert_array_equal(st, z["G6_events/starts"])
        np.testing.assert_array_equal(ln, z["G6_events/lengths"])
    c2, evs = synth.file_trace_counts(3_000_000, 5)
    x = synth.counts_to_pa(c2, np.float64)
    rs, rl = oracle.lambda_events(x, threshold=90.0)
    st, ln = ctx.detect_events(torch.from_numpy(c2.astype(np.int16)).cuda(), synth.QUANTUM)
    np.testing.assert_array_equal(st, rs)
    np.testing.assert_array_equal(ln, rl)
    assert len(st) >= 3


def test_config3_file_pipeline_on_device(ctx, tmp_path):
    """BASELINE config 3 shape (reduced size): synthetic .abf -> events -> per-event SpeedyStatSplit,
    all on the GPU from the raw int16 counts, against the reference-shaped CPU route (oracle)."""
    from pypore_amd import abf, pipeline
    from pypore_amd.parsers import lambda_event_parser
    c, _ = synth.file_trace_counts(2_500_000, 9)
    path = str(tmp_path / "synthetic.abf")
    abf.write_abf(path, c.astype(np.int16))
    dt, x = abf.read_abf(path)
    assert dt == 0.01 and np.array_equal(x, synth.counts_to_pa(c, np.float64))
    dt2, st, ln, bl = pipeline.parse_abf(path)
    rs, rl = oracle.lambda_events(x, threshold=90.0)
    np.testing.assert_array_equal(st, rs)
    np.testing.assert_array_equal(ln, rl)
    for e in range(len(st)):
        ref = oracle.parse(x[st[e]:st[e] + ln[e]], prior_segments_per_second=10.)
        np.testing.assert_array_equal(bl[e], ref)
    # the drop-in class route gives the same events
    evs = lambda_event_parser(threshold=90).parse(x, quantum=synth.QUANTUM)
    assert [int(e.start) for e in evs] == list(rs) and [int(e.duration) for e in evs] == list(rl)


def test_events_in_arbitrary_memory_order(ctx):
    """ps_segment_events with events that are not sorted by address (and overlap): output regions are
    laid out by cumulative event length, not by address."""
    import torch
    from pypore_amd import _lib
    c = synth.random_dwell_counts(600000, 77)
    t = torch.from_numpy(synth.counts_to_pa(c, np.float32)).cuda()
    starts = np.array([400000, 0, 150000, 390000], dtype=np.int64)
    lens = np.array([200000, 120000, 60000, 100000], dtype=np.int64)
    params = _lib.split_params(prior_segments_per_second=10.)
    b, boff, _ = ctx.segment_events(t, starts, lens, params, synth.QUANTUM)
    b = b.cpu().numpy()
    x = synth.counts_to_pa(c, np.float64)
    for e in range(4):
        ref = oracle.parse(x[starts[e]:starts[e] + lens[e]], prior_segments_per_second=10.)
        np.testing.assert_array_equal(b[boff[e]:boff[e + 1]], ref)


@pytest.mark.parametrize("seed", range(16))
def test_randomised_event_batches_fp32_int16_odd_offsets(seed, ctx):
    """Small edition of tools/fuzz_gpu.py: random levels / noise (incl. noise-free) / DC offsets, 1-7 events at odd
    offsets of one device buffer, fp32 or int16, default and verify mode: boundaries equal the oracle's."""
    import torch
    from pypore_amd import _lib
    rng = np.random.RandomState(777 + seed)
    mw = int(rng.choice([8, 20, 100, 250]))
    W = int(max(2 * mw, rng.choice([400, 1000, 4000, 10000, 25000])))
    params = dict(min_width=mw, max_width=int(max(rng.choice([W, 3 * W, 1000000]), mw)), window_width=W,
                  prior_segments_per_second=float(rng.choice([1., 10., 100.])))
    n_ev = int(rng.choice([1, 3, 7]))
    sigma = float(rng.choice([0.0, 1.0, 4.0, 30.0, 150.0]))
    dc = int(rng.choice([0, 500, -3000, 9000]))
    evs, starts, pos = [], [], 0
    for _ in range(n_ev):
        n = int(rng.randint(3000, 120000 if n_ev == 1 else 40000))
        lo = int(rng.randint(50, 3000)); hi = lo + int(rng.randint(100, 30000))
        k = np.empty(n, dtype=np.int64); i = 0
        while i < n:
            d = int(rng.randint(lo, hi)); k[i:i + d] = int(rng.randint(-2500, 2500)); i += d
        if sigma > 0:
            k += np.rint(rng.normal(0.0, sigma, n)).astype(np.int64)
        evs.append(np.clip(k + dc, -32000, 32000))
        pos += int(rng.randint(0, 9)); starts.append(pos); pos += n
    buf = np.zeros(pos + 16, dtype=np.int64)
    for k, s in zip(evs, starts):
        buf[s:s + len(k)] = k
    dev = torch.from_numpy(buf.astype(np.int16)).cuda() if rng.randint(0, 2) else \
        torch.from_numpy((buf * synth.QUANTUM).astype(np.float32)).cuda()
    refs = [oracle.parse(k.astype(np.float64) * synth.QUANTUM, **params) for k in evs]
    try:
        for mode in (0, 2):
            ctx.set_option("mode", mode)
            b, boff, st = ctx.segment_events(dev, np.array(starts), np.array([len(k) for k in evs]),
                                             _lib.split_params(**params), synth.QUANTUM, want_stats=(mode == 0))
            b = b.cpu().numpy()
            for e, ref in enumerate(refs):
                np.testing.assert_array_equal(b[boff[e]:boff[e + 1]], ref)
            if st is not None:                           # per-segment statistics (K2 from the K0 digest) against numpy
                st = st.cpu().numpy()
                for e, (k, ref) in enumerate(zip(evs, refs)):
                    edges = [0] + list(ref) + [len(k)]
                    for i, (a0, b0) in enumerate(zip(edges, edges[1:])):
                        seg = k[a0:b0].astype(np.float64) * synth.QUANTUM
                        got = st[boff[e] + e + i]
                        assert got[2] == seg.min() and got[3] == seg.max()
                        np.testing.assert_allclose(got[0], seg.mean(), rtol=1e-5, atol=1e-12)
                        np.testing.assert_allclose(got[1], seg.std(), rtol=1e-5, atol=1e-9)
    finally:
        ctx.set_option("mode", int(os.environ.get("PORESEG_MODE", "0")))      # (what the context started with: tools/gpu_validate.sh)


def test_grid_detected_on_a_subset_is_confirmed_by_the_device(ctx):
    """The drop-in parse() finds the ADC grid on a strided subset of a large array; a sample on a finer grid elsewhere
    makes the device refuse (it checks every sample) and the host then searches all samples once."""
    from pypore_amd.parsers import SpeedyStatSplit
    k = synth.random_dwell_counts(300000, 21, 1000, 20000)
    x = k.astype(np.float64) * synth.QUANTUM
    x[123457] += 2.0 ** -8                               # not in the strided subset (stride 4), finer than 2**-5
    ref = oracle.parse(x, prior_segments_per_second=10.)
    got = _bounds(SpeedyStatSplit(prior_segments_per_second=10.).parse(x))
    np.testing.assert_array_equal(got, ref)
    x[123457] += 1e-7                                    # on no power-of-two grid at all
    with pytest.raises(ValueError):
        SpeedyStatSplit(prior_segments_per_second=10.).parse(x)


REGIMES = [
    ((100, 400), {}), ((100000, 1000000), {}), ((1000, 20000), dict(window_width=1000)),
    ((1000, 20000), dict(window_width=50000)), ((1000, 20000), dict(min_width=8, window_width=2000)),
    ((1000, 20000), dict(min_width=1000)), ((1000, 20000), dict(max_width=30000)),
    ((1000, 20000), dict(prior_segments_per_second=1000.)), ((20000, 400000), dict(max_width=100000)),
    ((1000, 20000), dict(window_width=100000)),
]


@pytest.mark.parametrize("regime", REGIMES, ids=[("dwell%d-%d " % r[0]) + ",".join("%s=%s" % kv for kv in r[1].items()) for r in REGIMES])
def test_dwell_and_parameter_regimes(regime, ctx):
    """Small edition of tools/regime_parity.py: dense and sparse steps (open-ended tiles, long bridges, forced splits),
    narrow and wide windows (incl. wider than the block-sum scan takes), several tilings, default and verify mode."""
    import torch
    from pypore_amd import _lib
    (lo, hi), extra = regime
    n = 1_500_000
    d = synth.dwell_table(91, n, lo, hi)
    lv = synth.LEVEL_COUNTS[np.arange(len(d)) % 5].astype(np.int32)
    t = ctx.synth_trace(n, 91, np.cumsum(d), lv, dtype=torch.float32)
    kw = dict(min_width=100, max_width=1000000, window_width=10000, prior_segments_per_second=10., sampling_freq=1e5)
    kw.update(extra)
    ref = oracle.parse(t.cpu().numpy().astype(np.float64), **kw)
    try:
        for tile in (0, 60000):
            ctx.set_tiling(tile, 0)
            for mode in (0, 2):
                ctx.set_option("mode", mode)
                b, _, _ = ctx.segment_batch(t, np.array([0, n], dtype=np.int64), _lib.split_params(**kw), synth.QUANTUM,
                                            want_stats=False)
                np.testing.assert_array_equal(b.cpu().numpy(), ref)
    finally:
        ctx.set_option("mode", int(os.environ.get("PORESEG_MODE", "0")))      # (what the context started with: tools/gpu_validate.sh)
        ctx.set_tiling(0, 0)


def test_large_dc_offset_on_a_fine_grid(ctx):
    """counts near 2^21 (a 50 pA baseline on a 2^-15 pA grid): n * max|k|^2 exceeds 2^53, so the uncentred
    sums the contender path would form are no longer exact integers in fp64 -- those windows take the whole-window
    fp64 scan instead (the block sums about the event's first sample stay exact: |k - m| < 23 000).  The reference's own
    cumsums round in this regime; boundaries still agree (oracle on the same float64 values), also in verify mode."""
    from pypore_amd.parsers import SpeedyStatSplit
    k = (synth.random_dwell_counts(400000, 77).astype(np.int64) - 1500) * 16 + 1638400
    assert np.abs(k).max() < 2 ** 23 and 10000 * float(np.abs(k).max()) ** 2 > 2.0 ** 53
    x = k.astype(np.float64) * 2.0 ** -15
    ref = oracle.parse(x, prior_segments_per_second=10.)
    for mode in (0, 2):
        ctx.set_option("mode", mode)
        try:
            segs = SpeedyStatSplit(prior_segments_per_second=10., quantum=2.0 ** -15).parse(x)
        finally:
            ctx.set_option("mode", int(os.environ.get("PORESEG_MODE", "0")))
        np.testing.assert_array_equal(_bounds(segs), ref)
        assert ctx.timings()["wide_redo"] == 0


@pytest.mark.parametrize("option,value,default", [("groups", 0, 1), ("tree_par", 0, 1), ("k0_waves", 2, 0), ("k0_waves", 1, 0)])
def test_round4_options_change_no_result(ctx, option, value, default):
    """The switches of round 4 -- the coarse pass over the group records, the deep subtree jobs shared by the waves of a
    workgroup, K0's occupancy cap -- are tuning knobs: same boundaries with each of them off, on the narrow digest (dense
    steps: many splits per window; long dwells: most windows hold none) and on the 64-bit digest (a filtered, re-quantised
    event: few deep jobs, the case tree_par_kernel is chosen for on the device), also in verify mode."""
    import torch
    from pypore_amd import _lib, engine
    p = _lib.split_params(prior_segments_per_second=10.)
    cases = []
    for seed, lo, hi in ((41, 150, 1500), (42, 3000, 60000)):
        k = synth.random_dwell_counts(700000, seed, lo, hi)
        cases.append((torch.from_numpy(synth.counts_to_pa(k, np.float32)).cuda(), synth.QUANTUM, 0))
    k = synth.random_dwell_counts(400000, 43, 1000, 20000)
    y = ctx.filter_bessel(torch.from_numpy(k.astype(np.int16)).cuda(), synth.QUANTUM)
    z, _, step = ctx.requantise(y)
    cases.append((z, step, 1))
    want = []
    for t, q, route in cases:
        b, _, _ = ctx.segment_batch(t, np.array([0, t.numel()], dtype=np.int64), p, q, want_stats=False)
        assert b.numel() > 20
        if not (os.environ.get("PORESEG_SCAN_BS") or os.environ.get("PORESEG_WIDE_BS")):
            assert ctx.timings()["wide_redo"] == route           # (0: 32-bit digest, 1: 64-bit digest)
        want.append(b.cpu().numpy())
    ctx.set_option(option, value)
    try:
        for mode in (0, 2):
            ctx.set_option("mode", mode)
            for (t, q, route), w in zip(cases, want):
                b, _, _ = ctx.segment_batch(t, np.array([0, t.numel()], dtype=np.int64), p, q, want_stats=False)
                np.testing.assert_array_equal(b.cpu().numpy(), w)
    finally:
        ctx.set_option(option, default)
        ctx.set_option("mode", int(os.environ.get("PORESEG_MODE", "0")))


@pytest.mark.parametrize("k0_waves", [0, 1])
def test_fp32_block_sums_at_the_edges_of_the_packed_conversion(ctx, k0_waves):
    """K0's fp32 fast route turns (x/q - m) into packed int16 by v_cvt_pknorm_i16_f32 (seg_bs.hpp, k0_block_sums_pkf32):
    offsets up to +-16383 stay on the 32-bit digest, anything wider must saturate, fail the range check and send the call
    to the 64-bit digest -- +-16384, +-32767, +-32768, beyond +-65536 (where a wrap-around would look narrow again) and
    counts next to 2^23; a sample half a count, or 2^-10 of a count, off the grid, a NaN and an infinity must be refused.
    Reference: cparsers.pyx:110-111 takes the samples as they are; the device's exact sums need them on the grid."""
    import torch
    from pypore_amd import _lib
    n = 600_000
    kw = dict(min_width=100, max_width=1000000, window_width=10000, prior_segments_per_second=10., sampling_freq=1e5)
    base = synth.random_dwell_counts(n, 5, 1000, 20000).astype(np.int64)
    ev = np.array([0, n], dtype=np.int64)
    first = int(base[0])

    def run(counts):
        x = torch.from_numpy(counts.astype(np.float32) * np.float32(synth.QUANTUM)).cuda()
        b, _, _ = ctx.segment_batch(x, ev, _lib.split_params(**kw), synth.QUANTUM, want_stats=False)
        return b.cpu().numpy()

    ctx.set_option("k0_waves", k0_waves)
    try:
        for amp in (16383, -16383, 16384, -16384, 32767, -32767, 32768, -32768, 65536 + 5, -65536 - 5, 131072, 3_000_000):
            k = base.copy()
            k[300_000:300_400] = first + amp               # a 400-sample excursion in the middle of whole blocks
            k[450_003] = first + amp                       # and a lone sample
            assert abs(k).max() < 2 ** 23
            ref = oracle.parse(k.astype(np.float64) * synth.QUANTUM, **kw)
            np.testing.assert_array_equal(run(k), ref, err_msg="amplitude %d" % amp)
        # counts next to 2^23 (offsets narrow): a short event, so that the reference's own fp64 sums of squares stay exact
        # (2.6e5 pA squared times 24 000 samples < 2^53; beyond that its cumsum rounding decides, tools/fuzz_gpu.py)
        n2 = 24_000
        k = synth.random_dwell_counts(n2, 6, 300, 2000).astype(np.int64)
        k += 8_380_000 - int(k.max())
        ref = oracle.parse(k.astype(np.float64) * synth.QUANTUM, **kw)
        x = torch.from_numpy(k.astype(np.float32) * np.float32(synth.QUANTUM)).cuda()
        b, _, _ = ctx.segment_batch(x, np.array([0, n2], dtype=np.int64), _lib.split_params(**kw), synth.QUANTUM, want_stats=False)
        np.testing.assert_array_equal(b.cpu().numpy(), ref)
        assert len(ref) > 5
        for bad in (0.5, 2.0 ** -10, float("nan"), float("inf"), -float("inf")):
            x = base.astype(np.float32) * np.float32(synth.QUANTUM)
            x[333_333] = (base[333_333] + bad) * synth.QUANTUM if np.isfinite(bad) else bad
            with pytest.raises(ValueError):
                ctx.segment_batch(torch.from_numpy(x).cuda(), ev, _lib.split_params(**kw), synth.QUANTUM, want_stats=False)
    finally:
        ctx.set_option("k0_waves", 0)
        ctx.set_option("wide_bs", 1)                         # (also forgets that this quantum just took the wide routes: the next tests start on the 32-bit digest)


@pytest.mark.parametrize("dtype", ["float32", "int16"])
def test_seams_out_of_anchors_get_a_second_chance_on_the_device(ctx, dtype):
    """A seam's bridge gives up after BR_MAX anchors without meeting a downstream list (densely stepped data: 4 of the 1 536
    seams of a 1e8-sample trace with dwells of 100-400 samples); the look-ahead kernel then continues those seams with room
    for more anchors before the call falls back to the host stitch (seg_device.hpp: EXT_MAX; 236 -> 7 ms on that trace).
    Here on a small dense trace with the budget lowered (option bridge_budget) so that most seams need it, against the oracle;
    with the second chance switched off the same calls take the host stitch and give the same boundaries.
    Reference: _recursive_split, cparsers.pyx:180-203 -- whatever way the chain is cut into pieces, it is one chain."""
    import torch
    from pypore_amd import _lib
    n = 3_000_000
    kw = dict(min_width=100, max_width=1000000, window_width=10000, prior_segments_per_second=10., sampling_freq=1e5)
    d = synth.dwell_table(77, n, 100, 400)
    lv = synth.LEVEL_COUNTS[np.arange(len(d)) % 5].astype(np.int32)
    t = ctx.synth_trace(n, 77, np.cumsum(d), lv, dtype=getattr(torch, dtype))
    x = t.cpu().numpy().astype(np.float64) * (synth.QUANTUM if dtype == "int16" else 1.0)
    ref = oracle.parse(x, **kw)
    ev = np.array([0, n], dtype=np.int64)
    ctx.set_option("wide_bs", 1)                             # (forget a wide route an earlier test may have left this quantum on)
    try:
        seen = {}
        for budget, ext in ((256, 1), (8, 1), (1, 1), (1, 0)):
            ctx.set_option("bridge_budget", budget)
            ctx.set_option("bridge_ext", ext)
            b, _, _ = ctx.segment_batch(t, ev, _lib.split_params(**kw), synth.QUANTUM, want_stats=False)
            np.testing.assert_array_equal(b.cpu().numpy(), ref, err_msg="budget %d, second chance %d" % (budget, ext))
            seen[(budget, ext)] = int(ctx.timings()["repairs"])
        assert seen[(256, 1)] == 0                                 # nothing to mend with the full budget on this trace
        assert 0 < seen[(8, 1)] < 1_000_000                        # mended on the device (51 seams)
        assert seen[(1, 1)] > 0                                    # (74 seams: more than get the long side buffer; either route)
        assert seen[(1, 0)] >= 1_000_000                           # (the host stitch marks its count that way)
    finally:
        ctx.set_option("bridge_budget", int(os.environ.get("PORESEG_BRIDGE_BUDGET", "256")))   # (what the context started with: tools/gpu_validate.sh)
        ctx.set_option("bridge_ext", 1)


# (name, dwell range, parameters, must an owner TAKE published chunks when the helpers stay?  Not where every stretch ends in a
#  forced split within the first helped chunk: max_width 123 456 is 24.7 windows, the listing starts at window 16 and the chunk
#  16..31 holds the forced split, which the owner works out itself)
SPARSE = [
    ("no step", None, {}, True),
    ("no step, max_width not a multiple of W/2", None, dict(max_width=123456), False),
    ("no step, W 4000, max_width 50000", None, dict(window_width=4000, max_width=50000), False),
    ("dwell 1e5-1e6", (100000, 1000000), {}, True),
    ("dwell 3e5-3e6, max_width 250000", (300000, 3000000), dict(max_width=250000), False),
]


@pytest.mark.parametrize("case", SPARSE, ids=[c[0] for c in SPARSE])
def test_long_stretches_without_splits_with_and_without_helpers(case, ctx):
    """The chain through a stretch without splits is sequential by the reference's definition (find_split, cparsers.pyx:186-201:
    the next window starts where the last anchor is); the look-ahead kernel's idle workgroups scan chunks of the stretch's
    window lattice ahead of the seam's owner, who takes their results instead of scanning (seg_device.hpp: LAT_D).  Whatever
    they publish, the boundaries are the oracle's -- across forced splits that keep the lattice (max_width a multiple of
    W/2) and ones that do not, and the same with the helpers switched off (option lat_help)."""
    import torch
    from pypore_amd import _lib
    name, dwell, extra, must_take = case
    n = 4_000_000
    kw = dict(min_width=100, max_width=1000000, window_width=10000, prior_segments_per_second=10., sampling_freq=1e5)
    kw.update(extra)
    lo, hi = dwell if dwell else (n + 1, n + 2)
    d = synth.dwell_table(77, n, lo, hi)
    lv = synth.LEVEL_COUNTS[np.arange(len(d)) % 5].astype(np.int32)
    t = ctx.synth_trace(n, 77, np.cumsum(d), lv, dtype=torch.float32)
    ref = oracle.parse(t.cpu().numpy().astype(np.float64), **kw)
    ctx.set_option("wide_bs", 1)                             # (forget a wide route an earlier test may have left this quantum on)
    windows, published, taken = {}, {}, {}
    try:
        # 2: the helpers stay until every workgroup is through with its own seams (never leave on idle polls): what they do is
        # then a property of the call, not of the timing; 1: the product's setting; 0: every seam walks alone
        for on in (2, 1, 0):
            ctx.set_option("lat_help", on)
            b, _, _ = ctx.segment_batch(t, np.array([0, n], dtype=np.int64), _lib.split_params(**kw), synth.QUANTUM, want_stats=False)
            np.testing.assert_array_equal(b.cpu().numpy(), ref, err_msg="helpers %d" % on)
            tm = ctx.timings()
            windows[on], published[on], taken[on] = int(tm["windows"]), int(tm["helper_chunks_published"]), int(tm["helper_chunks_taken"])
    finally:
        ctx.set_option("lat_help", int(os.environ.get("PORESEG_LAT_HELP", "1")))
    print("windows scanned with helpers that stay / helpers / none: %d / %d / %d; chunks published %s, taken by an owner %s"
          % (windows[2], windows[1], windows[0], published, taken))
    assert published[0] == 0 and taken[0] == 0
    if os.environ.get("PORESEG_SCAN_BS") != "0" and not os.environ.get("PORESEG_STITCH"):
        # (this used to be printed, not asserted -- with helpers that stay it no longer depends on the timing: they
        #  scanned chunks of the stretch ahead of its owner, published them, and the owner took published chunks instead of
        #  scanning -- with the oracle's boundaries above)
        assert published[2] > 0 and (taken[2] > 0 or not must_take), (published, taken)
        assert windows[2] > windows[0], windows


def test_helper_tags_survive_their_wrap_around(ctx):
    """Every listed stretch publishes under a tag of its own (24 bits, 4 096 reserved per call); before they run out the
    library clears the published results and starts over (poreseg.hip: lat_tag_next).  4 300 calls on a small trace without
    steps -- each lists stretches and is helped -- cross that point; every call must return the first call's boundaries
    (which the sparse-shapes test compares with the oracle)."""
    import torch
    from pypore_amd import _lib
    if os.environ.get("PORESEG_SCAN_BS") == "0" or os.environ.get("PORESEG_STITCH"):
        pytest.skip("the helpers belong to the block-sum device-stitch pipeline")
    n = 600_000
    kw = dict(min_width=100, max_width=1000000, window_width=10000, prior_segments_per_second=10., sampling_freq=1e5)
    d = synth.dwell_table(5, n, n + 1, n + 2)
    t = ctx.synth_trace(n, 5, np.cumsum(d), synth.LEVEL_COUNTS[np.arange(len(d)) % 5].astype(np.int32), dtype=torch.float32)
    ref = oracle.parse(t.cpu().numpy().astype(np.float64), **kw)
    ctx.set_option("wide_bs", 1)
    ev, p = np.array([0, n], dtype=np.int64), _lib.split_params(**kw)
    out = torch.empty(n // 100 + 2, dtype=torch.int32, device="cuda")
    first = None
    taken_late = 0
    ctx.set_option("lat_help", 2)                            # (helpers that stay: every call IS helped, whatever the timing)
    try:
        for k in range(4300):
            b, _, _ = ctx.segment_batch(t, ev, p, synth.QUANTUM, want_stats=False, out=out)
            if first is None:
                first = b.clone()
                np.testing.assert_array_equal(first.cpu().numpy(), ref)
            elif not torch.equal(b, first):
                raise AssertionError("call %d differs from the first" % k)
            if k >= 4200:
                taken_late += int(ctx.timings()["helper_chunks_taken"])
    finally:
        ctx.set_option("lat_help", int(os.environ.get("PORESEG_LAT_HELP", "1")))
    assert taken_late > 0                                    # published chunks are still found under their tags after the wrap-around


@pytest.mark.parametrize("k0_waves", [0, 1, 2])
@pytest.mark.parametrize("dtype", ["int16", "float32"])
def test_events_at_odd_sample_offsets_take_k0s_fast_route_correctly(ctx, dtype, k0_waves):
    """K0's fast route loads 16 bytes from addresses that are only sample-aligned (events cut out of a file trace
    start at any sample) -- through a vector type that says so, and after ps_create has probed the device.  Stretches at odd
    sample offsets of one trace, K0 one-shot and persistent, with the unaligned loads (the probe's verdict) and with the
    16-byte condition of rounds 1-4 (option k0_unaligned 0: those blocks take the general route): the oracle's boundaries
    and the same statistics every time."""
    import torch
    from pypore_amd import _lib
    n = 1_500_000
    kw = dict(min_width=100, max_width=1000000, window_width=10000, prior_segments_per_second=10., sampling_freq=1e5)
    d = synth.dwell_table(31, n, 2000, 20000)
    lv = synth.LEVEL_COUNTS[np.arange(len(d)) % 5].astype(np.int32)
    t = ctx.synth_trace(n, 31, np.cumsum(d), lv, dtype=getattr(torch, dtype))
    x = t.cpu().numpy().astype(np.float64) * (synth.QUANTUM if dtype == "int16" else 1.0)
    starts = np.array([1, 300_003, 650_007, 1_000_013], dtype=np.int64)          # odd offsets: 2 / 4 bytes from any 16-byte boundary
    lens = np.array([290_001, 340_000, 333_333, 480_000], dtype=np.int64)
    refs = [oracle.parse(x[a:a + l], **kw) for a, l in zip(starts, lens)]
    ref_stats = [oracle.segment_stats(x[a:a + l], r) for (a, l), r in zip(zip(starts, lens), refs)]
    ctx.set_option("wide_bs", 1)
    try:
        for unaligned in (1, 0):
            ctx.set_option("k0_unaligned", unaligned)
            ctx.set_option("k0_waves", k0_waves)
            b, off, st = ctx.segment_events(t, starts, lens, _lib.split_params(**kw), synth.QUANTUM, want_stats=True)
            b = b.cpu().numpy(); st = st.cpu().numpy()
            for e, r in enumerate(refs):
                np.testing.assert_array_equal(b[off[e]:off[e + 1]], r, err_msg="event %d, unaligned loads %d" % (e, unaligned))
                got = st[off[e] + e:off[e + 1] + e + 1]
                np.testing.assert_allclose(got[:, 0], ref_stats[e][:, 0], rtol=1e-5)
                np.testing.assert_allclose(got[:, 1], ref_stats[e][:, 1], rtol=1e-5, atol=1e-9)
    finally:
        ctx.set_option("k0_unaligned", 1)
        ctx.set_option("k0_waves", int(os.environ.get("PORESEG_K0_WAVES", "0")))
