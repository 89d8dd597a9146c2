"""ps_detect_segment_trace (round 6): a file trace's events and their boundaries in one call and one pass over the samples
-- against the two calls it replaces (ps_detect_events + ps_segment_events) and against the CPU oracle; events start at
every phase of the 8-sample blocks of the trace-aligned digest."""
import numpy as np
import pytest

import oracle
from pypore_amd import _lib, synth


@pytest.fixture(scope="module")
def ctx():
    from pypore_amd import engine
    return engine.context(0)


def test_library_exports_the_single_pass_entry_point():
    assert "ps_detect_segment_trace" in _lib.EXPORTS
    assert hasattr(_lib.lib(), "ps_detect_segment_trace")


def _trace(seed, n=1_200_000):
    return synth.file_trace_counts(n, seed, gap=30001 + seed, ev_lo=60000, ev_hi=200000)[0]


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["int16", "float32"])
def test_single_pass_equals_the_two_calls_and_the_oracle_at_every_start_phase(dtype, ctx):
    import torch
    phases = set()
    for seed in range(40, 48):
        c = _trace(seed)
        x = synth.counts_to_pa(c, np.float64)
        t = torch.from_numpy(c.astype(np.int16) if dtype == "int16" else synth.counts_to_pa(c, np.float32)).cuda()
        params = _lib.split_params(prior_segments_per_second=10., min_width=100, max_width=1000000, window_width=10000)
        st, ln, b, off, stats = ctx.detect_segment_trace(t, synth.QUANTUM, params, threshold=90.0, min_duration=1000,
                                                         want_stats=True)
        assert ctx.timings()["wide_redo"] != 3             # (counters[7] = 3: the call fell back to the two calls)
        st2, ln2 = ctx.detect_events(t, synth.QUANTUM, threshold=90.0, min_duration=1000)
        b2, off2, stats2 = ctx.segment_events(t, st2, ln2, params, synth.QUANTUM, want_stats=True)
        np.testing.assert_array_equal(st, st2)
        np.testing.assert_array_equal(ln, ln2)
        np.testing.assert_array_equal(off, off2)
        np.testing.assert_array_equal(b.cpu().numpy(), b2.cpu().numpy())
        # (mean and std are formed about the digest's centre -- the trace's first count here, the event's in the two calls:
        #  the same exact integer sums, last-bit differences in the final fp64 expressions; min and max are exact)
        s1, s2 = stats.cpu().numpy(), stats2.cpu().numpy()
        np.testing.assert_allclose(s1[:, 0], s2[:, 0], rtol=1e-12, atol=0)
        np.testing.assert_allclose(s1[:, 1], s2[:, 1], rtol=1e-7, atol=1e-9)
        np.testing.assert_array_equal(s1[:, 2:], s2[:, 2:])
        bh = b.cpu().numpy()
        for e in range(len(st)):                             # (event e: rows off[e] + e .. off[e + 1] + e, one more segment than boundaries)
            ed = np.concatenate(([0], bh[off[e]:off[e + 1]], [ln[e]]))
            for j in (0, len(ed) - 2):                       # its first and last segment against numpy
                seg = x[st[e] + ed[j]:st[e] + ed[j + 1]]
                np.testing.assert_allclose(s1[off[e] + e + j], [seg.mean(), seg.std(), seg.min(), seg.max()], rtol=1e-7, atol=1e-9)
        rs, rl = oracle.lambda_events(x, threshold=90.0, min_duration=1000)
        np.testing.assert_array_equal(st, rs)
        np.testing.assert_array_equal(ln, rl)
        bb = b.cpu().numpy()
        for e in range(len(st)):
            ref = oracle.parse(x[st[e]:st[e] + ln[e]], prior_segments_per_second=10.)
            np.testing.assert_array_equal(bb[off[e]:off[e + 1]], ref)
            phases.add(int(st[e]) & 7)
    assert phases == set(range(8))


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["no_events", "one_event_to_the_end", "short", "empty", "minwidth2", "offset_counts"])
def test_single_pass_edge_cases(case, ctx):
    import torch
    params = _lib.split_params(prior_segments_per_second=10.)
    kw = dict(threshold=90.0, min_duration=1000)
    oc = 0
    if case == "no_events":
        c = (synth.OPEN_COUNTS + synth.noise_counts(5, 0, 300000)).astype(np.int32)
    elif case == "one_event_to_the_end":
        c = np.concatenate([synth.OPEN_COUNTS + synth.noise_counts(6, 0, 20003), synth.random_dwell_counts(400000, 6)]).astype(np.int32)
    elif case == "short":
        c = _trace(41, 9000)
    elif case == "empty":
        c = np.zeros(0, dtype=np.int32)
    elif case == "minwidth2":                                # (the block-sum scan does not apply: the call makes the two calls)
        c = _trace(42, 400000)
        params = _lib.split_params(prior_segments_per_second=10., min_width=2, max_width=100000, window_width=500)
    else:
        c = _trace(43, 700000)
        oc = 1234                                            # ps_sample_format.offset_counts: pA = (raw value + oc) * quantum
        c = c - oc
    t = torch.from_numpy(c.astype(np.int16)).cuda()
    if case == "empty":
        st, ln, b, off, _ = ctx.detect_segment_trace(t, synth.QUANTUM, params, **kw)
        assert len(st) == 0 and list(off) == [0] and b.numel() == 0
        return
    st, ln, b, off, _ = ctx.detect_segment_trace(t, synth.QUANTUM, params, offset_counts=oc, **kw)
    st2, ln2 = ctx.detect_events(t, synth.QUANTUM, offset_counts=oc, **kw)
    b2, off2, _ = ctx.segment_events(t, st2, ln2, params, synth.QUANTUM, offset_counts=oc)
    np.testing.assert_array_equal(st, st2)
    np.testing.assert_array_equal(ln, ln2)
    np.testing.assert_array_equal(off, off2)
    np.testing.assert_array_equal(b.cpu().numpy(), b2.cpu().numpy())
    if case == "no_events":
        assert len(st) == 0
    if case == "one_event_to_the_end":
        assert len(st) == 1 and st[0] + ln[0] == len(c)


@pytest.mark.gpu
def test_single_pass_switch_and_pipeline_route(ctx):
    """option "single_pass" 0 makes the two calls inside the library; pipeline.segment_file_trace(single_pass=False) makes
    them from Python: all three give the same events and boundaries."""
    import torch
    from pypore_amd import pipeline
    c = _trace(44, 2_000_000)
    t = torch.from_numpy(c.astype(np.int16)).cuda()
    params = _lib.split_params(prior_segments_per_second=10.)
    a = pipeline.segment_file_trace(t, synth.QUANTUM, params, threshold=90.0, min_duration=1000, ctx=ctx)
    z = pipeline.segment_file_trace(t, synth.QUANTUM, params, threshold=90.0, min_duration=1000, ctx=ctx, single_pass=False)
    ctx.set_option("single_pass", 0)
    try:
        m = pipeline.segment_file_trace(t, synth.QUANTUM, params, threshold=90.0, min_duration=1000, ctx=ctx)
    finally:
        ctx.set_option("single_pass", 1)
    for r in (z, m):
        np.testing.assert_array_equal(a[0], r[0])
        np.testing.assert_array_equal(a[1], r[1])
        np.testing.assert_array_equal(a[2].cpu().numpy(), r[2].cpu().numpy())
        np.testing.assert_array_equal(a[3], r[3])
    assert len(a[0]) >= 3


@pytest.mark.gpu
def test_single_pass_wide_range_trace_takes_the_two_calls(ctx):
    """A trace whose counts span 2^14 or more around its first one: the narrow digest does not apply to the whole trace (it may to
    each event), the call falls back by itself and says so in counters[7]."""
    import torch
    c = _trace(45, 600000).astype(np.int64)
    c[300] = 30000                                          # a spike in the open channel
    c[301] = -30000
    t = torch.from_numpy(c.astype(np.int16)).cuda()
    params = _lib.split_params(prior_segments_per_second=10.)
    st, ln, b, off, _ = ctx.detect_segment_trace(t, synth.QUANTUM, params, threshold=90.0, min_duration=1000)
    assert ctx.timings()["wide_redo"] == 3
    st2, ln2 = ctx.detect_events(t, synth.QUANTUM, threshold=90.0, min_duration=1000)
    b2, off2, _ = ctx.segment_events(t, st2, ln2, params, synth.QUANTUM)
    np.testing.assert_array_equal(st, st2)
    np.testing.assert_array_equal(ln, ln2)
    np.testing.assert_array_equal(b.cpu().numpy(), b2.cpu().numpy())


@pytest.mark.gpu
def test_more_events_than_the_first_capacity_grow_the_host_arrays(ctx, monkeypatch):
    """Context.detect_events / detect_segment_trace start with room for engine.EVENT_CAP events (not n / min_duration: gigabytes
    for a long trace and a small min_duration) and take the library's count when a trace has more (PS_ERR_CAPACITY)."""
    import torch
    from pypore_amd import engine
    c = _trace(46, 1_500_000)
    t = torch.from_numpy(c.astype(np.int16)).cuda()
    params = _lib.split_params(prior_segments_per_second=10.)
    ref = ctx.detect_segment_trace(t, synth.QUANTUM, params, threshold=90.0, min_duration=1000)
    ref_ev = ctx.detect_events(t, synth.QUANTUM, threshold=90.0, min_duration=1000)
    assert len(ref[0]) >= 5
    monkeypatch.setattr(engine, "EVENT_CAP", 3)
    got = ctx.detect_segment_trace(t, synth.QUANTUM, params, threshold=90.0, min_duration=1000)
    got_ev = ctx.detect_events(t, synth.QUANTUM, threshold=90.0, min_duration=1000)
    np.testing.assert_array_equal(got_ev[0], ref_ev[0])
    np.testing.assert_array_equal(got_ev[1], ref_ev[1])
    for a, z in zip(got[:2] + (got[3],), ref[:2] + (ref[3],)):
        np.testing.assert_array_equal(a, z)
    np.testing.assert_array_equal(got[2].cpu().numpy(), ref[2].cpu().numpy())
