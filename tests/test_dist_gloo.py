"""world_size-2 gloo tests (CPU) of the N>1 path: unit sharding + the boundary gather.
The per-rank segmenter is injected; on CPU the oracle stands in for the GPU segmenter (the
distributed logic is device-agnostic: the same code runs with backend nccl/RCCL on GPUs)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import oracle
        from pypore_amd import dist as pdist
        from pypore_amd import synth
        lens = [50000, 20000, 50000, 0, 35000, 150, 50000]
        evs = [synth.counts_to_pa(synth.step_counts(n, 5000 + 1000 * i, 300 + i), np.float64) if n else np.zeros(0)
               for i, n in enumerate(lens)]

        def seg(units):
            return [oracle.parse(evs[u], prior_segments_per_second=10.) for u in units]

        out = pdist.segment_units_sharded(lens, seg)
        ok = all(np.array_equal(out[u], oracle.parse(evs[u], prior_segments_per_second=10.)) for u in range(len(lens)))
        # gather_varlen with an empty contribution
        t = torch.arange(3 * rank, dtype=torch.int32)
        g = pdist.gather_varlen(t)
        ok = ok and [x.numel() for x in g] == [3 * r for r in range(world)]
        # the single-collective gather: two batches in flight, then one that overflows its slot
        bg = pdist.BoundaryGather(8, torch.device("cpu"), torch.int32, depth=2)
        t0 = bg.submit(torch.arange(2 + rank, dtype=torch.int32) + 10 * rank)
        t1 = bg.submit(torch.zeros(0, dtype=torch.int32))
        r0, r1 = bg.result(t0), bg.result(t1)
        ok = ok and all(np.array_equal(r0[r].numpy(), np.arange(2 + r) + 10 * r) for r in range(world))
        ok = ok and all(x.numel() == 0 for x in r1)
        t2 = bg.submit(torch.arange(5 + 20 * rank, dtype=torch.int32))        # rank 1: 25 > 7 payload slots
        r2 = bg.result(t2)
        ok = ok and all(np.array_equal(r2[r].numpy(), np.arange(5 + 20 * r)) for r in range(world))
        # in place: the payload already sits HEADER elements into a buffer of at least one slot
        buf = torch.full((64,), -7, dtype=torch.int32)
        view = buf[pdist.BoundaryGather.HEADER:][:3 - rank]
        view.copy_(torch.arange(3 - rank, dtype=torch.int32) + 100 * rank)
        r3 = bg.result(bg.submit(view))
        ok = ok and all(np.array_equal(r3[r].numpy(), np.arange(3 - r) + 100 * r) for r in range(world))
        ok = ok and int(buf[0]) == 3 - rank                                    # the count went into the buffer's head
        rows = bg.result(bg.submit(view), host=False)                          # device-side consumer: rows, count in column 0
        ok = ok and rows.shape == (world, 8) and [int(c) for c in rows[:, 0]] == [3 - r for r in range(world)]
        q.put((rank, bool(ok), [len(b) for b in out]))
    finally:
        dist.destroy_process_group()


def test_sharded_units_and_boundary_gather_world2():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res)
    assert res[0][2] == res[1][2]            # every rank holds the full, identical result


def test_shard_units_balanced_and_deterministic():
    from pypore_amd.dist import shard_units
    lens = [7, 3, 9, 1, 4, 4, 8, 2]
    s = shard_units(lens, 3)
    assert sorted(sum(s, [])) == list(range(len(lens)))
    loads = [sum(lens[u] for u in r) for r in s]
    assert max(loads) - min(loads) <= max(lens)
    assert s == shard_units(lens, 3)
    assert shard_units([], 2) == [[], []]


@pytest.mark.parametrize("world,halo", [(2, 80000), (4, 60000), (8, 80000)])
def test_sharded_trace_stitch_equals_whole_trace(world, halo):
    """BASELINE config 5 logic on the CPU: pieces segmented independently (oracle as the per-rank
    segmenter) and joined at common spine anchors reproduce the whole-trace result bit for bit."""
    import oracle
    from pypore_amd import synth
    from pypore_amd.dist import shard_ranges, stitch_pieces
    n = 1_600_000
    x = synth.counts_to_pa(synth.random_dwell_counts(n, 61), np.float64)
    ref = oracle.parse(x, prior_segments_per_second=10.)
    pieces = []
    for lo, hi in shard_ranges(n, world, halo):
        b, f = oracle.parse_flags(x[lo:hi], prior_segments_per_second=10.)
        pieces.append((lo, hi, b, f))
    got = stitch_pieces(pieces, n, 10000, 100)
    np.testing.assert_array_equal(got, ref)


def test_sharded_trace_too_short_halo_fails_loudly():
    import oracle
    from pypore_amd import synth
    from pypore_amd.dist import shard_ranges, stitch_pieces
    n = 600_000
    x = synth.counts_to_pa(synth.random_dwell_counts(n, 62, 30000, 90000), np.float64)
    pieces = []
    for lo, hi in shard_ranges(n, 3, 21000):
        b, f = oracle.parse_flags(x[lo:hi], prior_segments_per_second=10.)
        pieces.append((lo, hi, b, f))
    with pytest.raises(RuntimeError, match="halo"):
        stitch_pieces(pieces, n, 10000, 100)


def _long_dwell_trace(n, seed):
    from pypore_amd import synth
    return synth.counts_to_pa(synth.random_dwell_counts(n, seed, 100000, 1000000), np.float64)


def _flat_stretch_trace():
    """steps | 3e6 samples of one level (only the forced splits at max_width cut it) | steps"""
    from pypore_amd import synth
    a = synth.random_dwell_counts(700_000, 71)
    flat = synth.LEVEL_COUNTS[2] + synth.noise_counts(72, 0, 3_000_000)
    b = synth.random_dwell_counts(800_000, 73)
    return synth.counts_to_pa(np.concatenate([a, flat, b]), np.float64)


@pytest.mark.parametrize("case,world,halo", [("short_halo", 3, 21000), ("long_dwell", 8, 80000), ("long_dwell", 4, 80000),
                                             ("flat", 8, 80000), ("flat", 2, 80000)])
def test_sharded_trace_seam_repair_equals_whole_trace(case, world, halo):
    """a seam that finds no common spine anchor inside the halo is extended and re-run (that seam only),
    and the stitched result still equals the whole-trace result -- dwells of 1e5..1e6 samples, a flat stretch of 3e6."""
    import oracle
    from pypore_amd import synth
    from pypore_amd.dist import shard_ranges, stitch_pieces
    if case == "short_halo":
        x = synth.counts_to_pa(synth.random_dwell_counts(600_000, 62, 30000, 90000), np.float64)
    elif case == "long_dwell":
        x = _long_dwell_trace(6_000_000, 64)
    else:
        x = _flat_stretch_trace()
    n = x.size
    ref = oracle.parse(x, prior_segments_per_second=10.)
    pieces = []
    for lo, hi in shard_ranges(n, world, halo):
        b, f = oracle.parse_flags(x[lo:hi], prior_segments_per_second=10.)
        pieces.append((lo, hi, b, f))
    calls = []

    def repair(r, lo, hi):
        calls.append((r, lo, hi))
        return oracle.parse_flags(x[lo:hi], prior_segments_per_second=10.)

    got = stitch_pieces(pieces, n, 10000, 100, repair=repair, halo=halo)
    np.testing.assert_array_equal(got, ref)
    assert len(calls) >= 1                       # (these cases cannot be joined inside the halo)
    assert all(hi - lo < n for _, lo, hi in calls[:1]) or world == 2     # a repair re-runs a stretch, not the trace


def test_sharded_trace_repair_that_comes_up_short():
    """a repair callback that can only reach part of the stretch it is asked for (it slices a local buffer)
    must not give silently wrong boundaries: reporting the length it really segmented either ends the piece there -- and
    the result is still the whole-trace result -- or, when that does not get past the old end, raises."""
    import oracle
    from pypore_amd.dist import shard_ranges, stitch_pieces
    x = _long_dwell_trace(6_000_000, 64)
    n = x.size
    ref = oracle.parse(x, prior_segments_per_second=10.)
    ranges = shard_ranges(n, 8, 80000)
    pieces = []
    for lo, hi in ranges:
        b, f = oracle.parse_flags(x[lo:hi], prior_segments_per_second=10.)
        pieces.append((lo, hi, b, f))
    short = []

    def repair_capped(r, lo, hi):                # reaches at most 1.2e6 samples past the start of the stretch
        hi2 = min(hi, lo + 1_200_000)
        short.append(hi2 < hi)
        b, f = oracle.parse_flags(x[lo:hi2], prior_segments_per_second=10.)
        return b, f, hi2 - lo

    got = stitch_pieces(pieces, n, 10000, 100, repair=repair_capped, halo=80000)
    np.testing.assert_array_equal(got, ref)

    def repair_local_only(r, lo, hi):            # the old contract: only the rank's own shard + halo
        hi2 = min(hi, ranges[r][1])
        b, f = oracle.parse_flags(x[lo:hi2], prior_segments_per_second=10.)
        return b, f, hi2 - lo

    with pytest.raises(RuntimeError, match="cannot extend"):
        stitch_pieces(pieces, n, 10000, 100, repair=repair_local_only, halo=80000)


def _worker_trace(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import oracle
        from pypore_amd import dist as pdist
        from pypore_amd import synth
        n = 700_000
        x = synth.counts_to_pa(synth.random_dwell_counts(n, 63), np.float64)

        def seg(lo, hi):
            return oracle.parse_flags(x[lo:hi], prior_segments_per_second=10.)

        got = pdist.segment_trace_sharded(n, seg, 10000, 100, halo=80000)
        ok = bool(np.array_equal(got, oracle.parse(x, prior_segments_per_second=10.)))
        # dwells far longer than the halo: the seam is repaired by the upstream rank, one more gather of that stretch
        n2 = 3_000_000
        x2 = synth.counts_to_pa(synth.random_dwell_counts(n2, 64, 100000, 1000000), np.float64)
        calls = []

        def seg2(lo, hi):
            calls.append((lo, hi))
            return oracle.parse_flags(x2[lo:hi], prior_segments_per_second=10.)

        got2 = pdist.segment_trace_sharded(n2, seg2, 10000, 100, halo=80000)
        ok = ok and bool(np.array_equal(got2, oracle.parse(x2, prior_segments_per_second=10.)))
        ok = ok and (len(calls) >= 2 if rank == 0 else len(calls) == 1)      # rank 0 is the upstream side of the one seam
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


def test_sharded_trace_world2_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_trace, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(ok for _, ok in res)
