"""BASELINE.json's configurations at their full sizes on one GPU.

config 2: 1 024 events x 50 000 samples in one device call, every event against the oracle.
config 5: ONE 10^9-sample trace: the whole-trace result on one GPU == the same trace cut into 8 stand-alone pieces
          with halo, joined at common spine anchors (the multi-GPU decomposition, dist.stitch_pieces) == the oracle run
          on 10 overlapping ~10^8-sample chunks on the host's cores and joined the same way (the reference
          itself cannot hold 10^9 samples in one call here).
config 4: the file loop (host -> HBM on a copy stream, detector + segmenter per file) on 6 files of 7.5e6 samples with
          two of them checked event by event against the oracle; bench.py --workload files runs the 64 x 7.5e7 shape.
"""
import threading

import os

import numpy as np
import pytest

import oracle
from pypore_amd import synth

pytestmark = pytest.mark.gpu
DEF = dict(min_width=100, max_width=1000000, window_width=10000, prior_segments_per_second=10., sampling_freq=1e5)


@pytest.fixture(scope="module")
def ctx():
    from pypore_amd import engine
    return engine.context(0)


@pytest.mark.parametrize("mode", [0, 2], ids=["default", "verify"])
def test_config2_full_1024_events(ctx, mode):
    """mode 2 (verify): the screen -- coarse pass, block bounds, drain filter -- and the exact fp64 scan of EVERY window must
    agree, or the call fails (the driver's own run proves screen == exact at full size)."""
    import torch
    from pypore_amd import _lib
    n_ev, n = 1024, 50000
    ctx.set_option("mode", mode)
    counts = np.stack([synth.step_counts(n, 10000, ev) for ev in range(n_ev)])            # config2_event(ev): seed = event id
    t = torch.from_numpy(synth.counts_to_pa(counts.reshape(-1), np.float32)).cuda()
    ev_off = np.arange(n_ev + 1, dtype=np.int64) * n
    try:
        bounds, boff, stats = ctx.segment_batch(t, ev_off, _lib.split_params(**DEF), synth.QUANTUM, want_stats=True)
        # (the counter of whole-window fp64 scans belongs to the block-sum scan of the device-stitch pipeline: the default)
        if mode == 2 and not (os.environ.get("PORESEG_SCAN_BS") or os.environ.get("PORESEG_STITCH")):
            assert ctx.timings()["full_exact_scans"] >= ctx.timings()["windows"] > 2 * n_ev     # every window was scanned twice
    finally:
        ctx.set_option("mode", int(os.environ.get("PORESEG_MODE", "0")))      # (what the context started with: tools/gpu_validate.sh)
    b, st = bounds.cpu().numpy(), stats.cpu().numpy()
    total = 0
    for ev in range(n_ev):
        x = synth.counts_to_pa(counts[ev], np.float64)
        ref = oracle.parse(x, **DEF)
        np.testing.assert_array_equal(b[boff[ev]:boff[ev + 1]], ref, err_msg="event %d" % ev)
        if ev % 64 == 0:
            rs = oracle.segment_stats(x, ref)
            np.testing.assert_allclose(st[boff[ev] + ev: boff[ev + 1] + ev + 1, :2], rs[:, :2], rtol=1e-5)
        total += len(ref)
    assert total == boff[-1] and total > 4 * n_ev


def test_config5_1e9_trace_whole_vs_pieces_vs_oracle_chunks(ctx):
    import torch
    from pypore_amd import _lib
    from pypore_amd.dist import shard_ranges, stitch_pieces
    n, seed, W, mw = 1_000_000_000, 2024, DEF["window_width"], DEF["min_width"]
    d = synth.dwell_table(seed, n)
    ends = np.cumsum(d)
    lv = synth.LEVEL_COUNTS[np.arange(len(d)) % 5].astype(np.int32)
    t = ctx.synth_trace(n, seed, ends, lv, dtype=torch.float32)
    params = _lib.split_params(**DEF)
    whole, _, _, flags = ctx.segment_batch(t, np.array([0, n]), params, synth.QUANTUM, want_stats=False, want_spine=True)
    whole = whole.cpu().numpy().astype(np.int64)
    assert np.all(np.diff(whole) >= mw) and whole[0] >= mw and whole[-1] <= n - mw
    # (a) 8 stand-alone pieces on the GPU, stitched
    pieces = []
    for lo, hi in shard_ranges(n, 8, 8 * W):
        pb, _, _, pf = ctx.segment_batch(t[lo:hi], np.array([0, hi - lo]), params, synth.QUANTUM, want_stats=False, want_spine=True)
        pieces.append((lo, hi, pb.cpu().numpy(), pf.cpu().numpy()))
    np.testing.assert_array_equal(stitch_pieces(pieces, n, W, mw), whole)
    # (b) the piece generator equals the whole-trace generator (what bench.py --workload sharded-trace relies on)
    lo, hi = shard_ranges(n, 8, 8 * W)[5]
    assert torch.equal(ctx.synth_trace(hi - lo, seed, ends, lv, dtype=torch.float32, start=lo), t[lo:hi])
    # (c) the oracle on 10 overlapping chunks of ~1e8 samples (one host thread each), joined the same way
    ranges = shard_ranges(n, 10, 8 * W)
    res = [None] * len(ranges)

    def work(r):
        clo, chi = ranges[r]
        x = t[clo:chi].cpu().numpy().astype(np.float64)
        res[r] = oracle.parse_flags(x, **DEF)

    th = [threading.Thread(target=work, args=(r,)) for r in range(len(ranges))]
    for x in th:
        x.start()
    for x in th:
        x.join()
    ref = stitch_pieces([(ranges[r][0], ranges[r][1], res[r][0], res[r][1]) for r in range(len(ranges))], n, W, mw)
    np.testing.assert_array_equal(ref, whole)
    # the first chunk's spine flags are the whole trace's up to its trusted end
    lim = ranges[0][1] - 2 * W - 2 * mw
    k = int(np.searchsorted(whole, lim))
    np.testing.assert_array_equal(flags.cpu().numpy()[:k], res[0][1][:k])


def test_config4_file_loop_two_streams(ctx, tmp_path):
    """Six .abf files: read_abf_counts -> pinned host -> HBM on a copy stream while the previous file is detected and
    segmented (the loop of bench.py --workload files, from real files)."""
    import os
    import torch
    from pypore_amd import _lib, abf, pipeline
    n, params = 7_500_000, _lib.split_params(**DEF)
    paths, tables = [], []
    for f in range(6):
        counts, events = synth.file_trace_counts(n, 900 + f)
        p = os.path.join(str(tmp_path), "f%d.abf" % f)
        abf.write_abf(p, counts.astype(np.int16))
        paths.append(p)
        tables.append((counts, events))
    host = []
    for p in paths:
        dt, raw, scale, offset = abf.read_abf_counts(p)
        assert (dt, scale, offset) == (0.01, synth.QUANTUM, 0.0)
        h = torch.empty(n, dtype=torch.int16, pin_memory=True)
        h.copy_(torch.from_numpy(np.array(raw)))
        host.append(h)
    dbuf = [torch.empty(n, dtype=torch.int16, device="cuda") for _ in range(2)]
    cs = torch.cuda.Stream()
    evs = [torch.cuda.Event() for _ in paths]
    with torch.cuda.stream(cs):
        dbuf[0].copy_(host[0], non_blocking=True)
        evs[0].record(cs)
    out = []
    for j in range(len(paths)):
        evs[j].synchronize()
        if j + 1 < len(paths):
            with torch.cuda.stream(cs):
                dbuf[(j + 1) % 2].copy_(host[j + 1], non_blocking=True)
                evs[j + 1].record(cs)
        st, ln, b, o, _ = pipeline.segment_file_trace(dbuf[j % 2], synth.QUANTUM, params, threshold=90.0)
        out.append((st, ln, b.cpu().numpy(), o))
    for j, (st, ln, b, o) in enumerate(out):
        counts, events = tables[j]
        if j in (0, 5):
            es, el = oracle.lambda_events(synth.counts_to_pa(counts, np.float64), threshold=90.0)
            assert list(zip(st.tolist(), ln.tolist())) == list(zip(es.tolist(), el.tolist()))
            for e, (a, l) in enumerate(zip(st, ln)):
                ref = oracle.parse(synth.counts_to_pa(counts[a:a + l], np.float64), **DEF)
                np.testing.assert_array_equal(b[o[e]:o[e + 1]], ref)


def test_stream_pool_batches_in_flight_give_the_same_results(ctx):
    """engine.StreamPool: twelve batches (three different traces) over four contexts / streams / host threads -- every
    result equals the one-at-a-time result of its trace; segment_many keeps the batch order."""
    import torch
    from pypore_amd import _lib, engine
    params = _lib.split_params(**DEF)
    traces = []
    for seed, n in ((31, 20_000_000), (32, 5_000_000), (33, 12_345_678)):
        d = synth.dwell_table(seed, n)
        ends = np.cumsum(d)
        lv = synth.LEVEL_COUNTS[np.arange(len(d)) % 5].astype(np.int32)
        traces.append((ctx.synth_trace(n, seed, ends, lv, dtype=torch.float32), np.array([0, n], dtype=np.int64)))
    alone = [ctx.segment_batch(t, off, params, synth.QUANTUM, want_stats=False)[0].cpu().numpy() for t, off in traces]
    pool = engine.StreamPool(0, 4)
    try:
        batches = [traces[k % 3] for k in range(12)]
        res = pool.segment_many(batches, params, synth.QUANTUM)
        for k, (b, boff, st) in enumerate(res):
            np.testing.assert_array_equal(b.cpu().numpy(), alone[k % 3])
            assert boff[-1] == len(alone[k % 3]) and st is None
    finally:
        pool.close()
    ref = oracle.parse(traces[1][0].cpu().numpy().astype(np.float64), **DEF)
    np.testing.assert_array_equal(alone[1], ref)


def test_config3_full_size_abf_file_end_to_end(ctx, tmp_path):
    """BASELINE config 3 at its full size: a 10^8-sample .abf @100 kHz on disk (200 MB), read_abf_counts -> device ->
    lambda_event_parser(threshold=90) -> per-event SpeedyStatSplit, against the oracle: the events, and the boundaries
    of EVERY event (oracle on the host's cores, one thread per event)."""
    import os
    import torch
    from pypore_amd import abf, pipeline
    n, seed = 100_000_000, 4242
    ends, lv, _ = synth.file_trace_table(n, seed)
    counts = ctx.synth_trace(n, seed, ends, lv, dtype=torch.int16).cpu().numpy()
    path = os.path.join(str(tmp_path), "config3.abf")
    abf.write_abf(path, counts)
    dt, st, ln, bl = pipeline.parse_abf(path)
    assert dt == 0.01 and len(st) > 50
    x = counts.astype(np.float64) * synth.QUANTUM
    es, el = oracle.lambda_events(x, threshold=90.0)
    assert list(zip(st.tolist(), ln.tolist())) == list(zip(es.tolist(), el.tolist()))
    refs = [None] * len(st)

    def work(lo, hi):
        for e in range(lo, hi):
            refs[e] = oracle.parse(x[st[e]:st[e] + ln[e]], **DEF)

    nthr = min(32, len(st))
    cuts = np.linspace(0, len(st), nthr + 1).astype(int)
    th = [threading.Thread(target=work, args=(cuts[i], cuts[i + 1])) for i in range(nthr)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for e in range(len(st)):
        np.testing.assert_array_equal(bl[e], refs[e], err_msg="event %d" % e)
    assert sum(len(b) for b in bl) > 5000


def test_dense_steps_full_size_are_mended_on_the_device(ctx):
    """The 1e8-sample trace with dwells of 100-400 samples (400 000 segments): four of its 1 536 seams run out of anchors, the
    look-ahead kernel continues them (seg_device.hpp: EXT_MAX) -- the call must not fall back to the host stitch (236 ms
    before round 5, 6.5 with the second chance) and must give what the host stitch gives (option bridge_ext = 0: tiles with
    halos, the LDS-window kernels, the chain followed on the host; the small edition of this test in test_gpu_parity.py
    compares both with the oracle, which would take half a minute of one core here).
    Reference: _recursive_split, cparsers.pyx:180-203."""
    import torch
    from pypore_amd import _lib
    if os.environ.get("PORESEG_SCAN_BS") == "0" or os.environ.get("PORESEG_STITCH"):
        pytest.skip("the second chance belongs to the block-sum device-stitch pipeline")
    n = 100_000_000
    d = synth.dwell_table(77, n, 100, 400)
    lv = synth.LEVEL_COUNTS[np.arange(len(d)) % 5].astype(np.int32)
    t = ctx.synth_trace(n, 77, np.cumsum(d), lv, dtype=torch.float32)
    ctx.set_option("wide_bs", 1)                             # (forget a wide route an earlier test may have left this quantum on)
    ctx.set_option("bridge_budget", 256)                     # (the library's own: tools/gpu_validate.sh runs the suite with less)
    try:
        b, _, _ = ctx.segment_batch(t, np.array([0, n], dtype=np.int64), _lib.split_params(**DEF), synth.QUANTUM, want_stats=False)
        repairs = int(ctx.timings()["repairs"])
        assert 0 < repairs < 1_000_000, "the call fell back to the host stitch, or had nothing to mend"
        ctx.set_option("bridge_ext", 0)
        ref, _, _ = ctx.segment_batch(t, np.array([0, n], dtype=np.int64), _lib.split_params(**DEF), synth.QUANTUM, want_stats=False)
        assert int(ctx.timings()["repairs"]) >= 1_000_000      # (the host stitch marks its count that way)
    finally:
        ctx.set_option("bridge_ext", 1)
        ctx.set_option("bridge_budget", int(os.environ.get("PORESEG_BRIDGE_BUDGET", "256")))
    assert torch.equal(b, ref) and b.numel() > 300_000
