"""Currents that are written out on first access (grid.Deferred behind Segment.current): what File(filename) and the
device filter hand to the classes.  CPU only: the device routes are covered by tests/test_experiment.py on the GPU."""
import os
import numpy as np
import pytest

from pypore_amd import abf, core, grid
from pypore_amd.core import Segment, raw_current
from pypore_amd.grid import Deferred, GridArray, grid_of


def _counts(n=5000, seed=3):
    rng = np.random.default_rng(seed)
    return rng.integers(-3000, 3000, n).astype(np.int16)


def test_deferred_from_counts_is_the_reference_readers_array():
    k = _counts()
    q, o = 10.0 / 0.0005 / 20 / 32768, -1.25
    d = Deferred.from_counts(k, q, o)
    assert len(d) == k.size and d.shape == (k.size,) and d.dtype == np.float64 and not d.built
    assert grid_of(d) == (k, q, o) or (grid_of(d)[0] is k and grid_of(d)[1:] == (q, o))
    assert not d.built                                   # asking for the grid does not write the array out
    want = GridArray.from_counts(k, q, o)
    got = np.asarray(d)
    assert d.built and isinstance(d.value(), GridArray)
    assert np.array_equal(got, want) and got.dtype == np.float64
    assert d.value() is d.value()                        # built once


def test_deferred_slices_stay_unbuilt_and_share_the_parent():
    k = _counts()
    d = Deferred.from_counts(k, 0.03125, 0.0)
    s = d[100:900]
    t = s[10:20]
    assert isinstance(s, Deferred) and isinstance(t, Deferred) and len(s) == 800 and len(t) == 10
    assert np.array_equal(s.counts, k[100:900]) and np.array_equal(t.counts, k[110:120])
    assert not d.built and not s.built
    assert np.array_equal(np.asarray(t), k[110:120] * 0.03125)
    assert d.built                                       # the stretch came out of the parent's array ...
    assert np.shares_memory(np.asarray(t), d.value())    # ... as a view of it
    assert isinstance(d[5:9], GridArray)                 # once built, slices are the array's own
    assert np.array_equal(d[5:9].counts, k[5:9])
    assert d[7] == k[7] * 0.03125 and np.array_equal(d[::2], d.value()[::2])
    assert len(d[4000:9000]) == 1000 and len(d[10:5]) == 0


def test_segment_current_builds_on_first_access_only():
    k = _counts()
    seg = Segment(current=Deferred.from_counts(k, 0.5, 1.0), start=0, duration=k.size)
    assert seg.n == k.size and len(seg) == k.size
    assert isinstance(raw_current(seg), Deferred) and not raw_current(seg).built
    cur = seg.current
    assert isinstance(cur, np.ndarray) and np.array_equal(cur, k * 0.5 + 1.0)
    assert raw_current(seg) is cur                       # the built array replaced the stand-in
    assert seg.mean == pytest.approx(np.mean(k * 0.5 + 1.0))
    seg.current = np.arange(4.0)
    assert seg.n == 4 and seg.max == 3.0
    del seg.current
    with pytest.raises(AttributeError):
        seg.current
    plain = Segment(current=np.arange(6.0), start=0, duration=6)
    assert plain.current is raw_current(plain) and plain.n == 6


def test_meta_and_json_of_a_deferred_segment():
    k = _counts(200)
    seg = Segment(current=Deferred.from_counts(k, 0.25, 0.0), start=10, duration=200)
    d = seg.to_dict()
    assert d['mean'] == pytest.approx(float(np.mean(k * 0.25))) and d['start'] == 10
    seg.to_meta()
    assert not hasattr(seg, 'current') and seg.mean == pytest.approx(float(np.mean(k * 0.25)))
    m = core.MetaSegment(current=Deferred.from_counts(k, 0.25, 0.0), start=0, duration=200)
    assert m.n == 200 and m.min == k.min() * 0.25


def test_file_current_is_written_out_when_read(tmp_path):
    from pypore_amd.DataTypes import File
    k = _counts(20000)
    path = abf.write_abf(os.path.join(tmp_path, "d.abf"), k, adc_range=10.0, adc_resolution=32768,
                         instrument_scale=0.0005, signal_gain=20.0, instrument_offset=1.5)
    dt, want = abf.read_abf(path)
    f = File(path)
    assert isinstance(raw_current(f), Deferred) and not raw_current(f).built
    assert f.second == pytest.approx(1000.0 / dt)
    g = grid_of(raw_current(f))
    assert np.array_equal(g[0], k) and g[1] == want.quantum and g[2] == want.offset
    cur = f.current                                      # what the reference-style reader returns, bit for bit
    assert isinstance(cur, GridArray) and np.array_equal(cur, want) and np.array_equal(cur.counts, k)
    assert not cur.flags.writeable


def test_deferred_device_budget_counts_parked_bytes():
    class FakeTensor:                                    # (the CPU suite has no GPU: the accounting only)
        def __init__(self, a): self.a = a
        def numel(self): return self.a.size
        def element_size(self): return 8
        def cpu(self): return self
        def numpy(self): return self.a
    before = Deferred.live_device_bytes
    a = np.linspace(0.0, 1.0, 1000)
    d = Deferred.from_tensor(FakeTensor(a), 2.0)
    assert Deferred.live_device_bytes == before + 8000 and d.tensor is not None
    s = d[10:20]
    assert isinstance(s, Deferred) and s.counts is None
    assert np.array_equal(np.asarray(s), a[10:20] + 2.0)
    assert d.built and d.tensor is None and Deferred.live_device_bytes == before
    old = Deferred.DEVICE_BYTES_MAX
    try:
        Deferred.DEVICE_BYTES_MAX = before + 100
        e = Deferred.from_tensor(FakeTensor(a), 0.0)     # over the budget: copied right away
        assert e.built and Deferred.live_device_bytes == before
    finally:
        Deferred.DEVICE_BYTES_MAX = old
    f = Deferred.from_tensor(FakeTensor(a), 0.0)
    assert Deferred.live_device_bytes == before + 8000
    del f
    assert Deferred.live_device_bytes == before


def test_segments_from_edges_equals_the_constructor():
    from pypore_amd.core import segments_from_edges
    k = _counts(3000)
    edges = [0, 7, 250, 251, 1800, 3000]
    stats = np.arange(20.0).reshape(5, 4)
    for make in (lambda: Deferred.from_counts(k, 0.5, 2.0), lambda: Deferred.from_counts(k, 0.5, 2.0)[0:3000],
                 lambda: k * 0.5 + 2.0, lambda: GridArray.from_counts(k, 0.5, 2.0)):
        cur = make()
        segs = segments_from_edges(cur, edges, stats)
        ref = [Segment(current=(k * 0.5 + 2.0)[a:z], start=a, duration=z - a, end=z) for a, z in zip(edges, edges[1:])]
        for i, (s, r) in enumerate(zip(segs, ref)):
            assert (s.start, s.end, s.duration, s.n) == (r.start, r.end, r.duration, r.n)
            assert s.mean == stats[i][0] and s.max == stats[i][3]          # device rows win, as with the constructor
            if isinstance(cur, Deferred):
                assert isinstance(raw_current(s), Deferred) and np.array_equal(raw_current(s).counts, k[s.start:s.end])
            assert np.array_equal(s.current, r.current)
    d = Deferred.from_counts(k, 0.5, 2.0)
    d.value()
    assert all(isinstance(raw_current(s), np.ndarray) for s in segments_from_edges(d, edges))
    assert segments_from_edges(d, [0, 3000])[0].std == pytest.approx(np.std(k * 0.5 + 2.0))


def test_stretches_reach_the_device_as_views_of_the_familys_counts():
    """Round 5: the counts of a file go up once, its events are stretches of that tensor wherever they start (no alignment
    clone), and device_stretch names the family's tensor and the range -- what ps_filter_requantise_batch is handed."""
    class FakeTensor(object):
        def __init__(self, a, device="dev0"): self.a, self.device = a, device
        def numel(self): return self.a.size
        def element_size(self): return 2
        def __getitem__(self, sl): return FakeTensor(self.a[sl], self.device)
    uploads = []

    def upload(counts, dev):
        uploads.append(len(counts))
        return FakeTensor(np.asarray(counts, dtype=np.int16), dev)
    k = _counts(5000)
    root = Deferred.from_counts(k, 0.25, 1.5)
    a, b = root[1001:2501], root[3003:3503]                             # (odd starts: off every 16-byte boundary)
    ta = a.device_counts("dev0", upload)
    assert uploads == [5000] and np.array_equal(ta.a, k[1001:2501])
    wa, wb = a.device_stretch("dev0", upload), b.device_stretch("dev0", upload)
    assert uploads == [5000]                                            # one upload for the family
    assert wa[0] is wb[0] and wa[1:] == (1001, 1500) and wb[1:] == (3003, 500)
    assert root.device_stretch("dev0", upload)[1:] == (0, 5000)
    # a current without counts (a float64 device tensor, a plain array) has no stretch
    assert Deferred(10, lambda: np.zeros(10)).device_stretch("dev0", upload) is None
    root._release()
