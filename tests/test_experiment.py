"""Experiment / Sample (DataTypes.py:938-1049): the reference's default workflow -- files -> events -> filter ->
segments.  The host logic runs everywhere (parsers that work on numpy); the device route is a GPU test."""
import os

import numpy as np
import pytest

import oracle
from pypore_amd import abf, synth
from pypore_amd.core import Segment
from pypore_amd.DataTypes import Event, Experiment, File, MetaEvent, Sample
from pypore_amd.parsers import SpeedyStatSplit, lambda_event_parser


class _Halves(object):
    """A segmenter that needs no device: two halves."""

    def parse(self, current):
        n = len(current)
        return [Segment(current=current[:n // 2], start=0, duration=n // 2, end=n // 2),
                Segment(current=current[n // 2:], start=n // 2, duration=n - n // 2, end=n)]


def _host_detector():
    # custom rules are evaluated on the host over numpy pieces (parsers.lambda_event_parser.parse)
    return lambda_event_parser(threshold=90, rules=[lambda ev: ev.duration > 1000, lambda ev: ev.max < 90])


def _trace(seed):
    x = np.full(60000, 110.0)
    x[5000:20000] = 50.0 + (seed % 3)
    x[30000:52000] = 42.0
    x[55000:55500] = 40.0                                  # too short for the rule above
    return x


def test_experiment_walks_files_events_segments(capsys):
    files = [File(current=_trace(s), timestep=0.01) for s in (1, 2)]
    exp = Experiment(files, name="pair")
    exp.parse(event_detector=_host_detector(), segmenter=_Halves(), filter_params=None)
    out = capsys.readouterr().out.splitlines()
    assert out[0].startswith("Opening") and out[1] == "\tDetected 2 Events" and out[2] == "\t\tEvent 1 has 2 segments"
    assert exp.n == 2 and exp.name == "pair" and [f.n for f in exp.files] == [2, 2]
    assert len(exp.events) == 4 and len(exp.segments) == 8
    ev = exp.files[1].events[0]
    assert ev.start == pytest.approx(0.05) and ev.duration == pytest.approx(0.15) and not ev.filtered
    seg = ev.segments[1]                                     # segments are rescaled to seconds and know their event
    assert seg.event is ev and seg.start == pytest.approx(0.075) and seg.mean == pytest.approx(52.0)
    assert isinstance(ev.state_parser, _Halves) and exp.files[0].event_parser.threshold == 90


def test_experiment_without_segmenter_and_as_metadata():
    exp = Experiment([File(current=_trace(0), timestep=0.01)])
    exp.parse(event_detector=_host_detector(), segmenter=None, filter_params=None, verbose=False)
    assert [e.n for e in exp.events] == [0, 0] and exp.segments == []
    exp2 = Experiment([File(current=_trace(0), timestep=0.01)])
    exp2.parse(event_detector=_host_detector(), segmenter=_Halves(), filter_params=None, verbose=False, meta=True)
    assert all(isinstance(e, MetaEvent) for e in exp2.events) and not hasattr(exp2.files[0], "current")
    assert exp2.events[0].mean == pytest.approx(50.0)
    exp2.delete()
    with pytest.raises(NotImplementedError):
        Experiment([]).apply_hmm(None)


def test_sample_holds_events_and_files():
    f = File(current=_trace(0), timestep=0.01)
    f.parse(_host_detector())
    s = Sample(events=f.events, files=[f], label="substrate A")
    assert s.label == "substrate A" and len(s.events) == 2
    s.delete()
    assert not hasattr(s, "events") and not hasattr(s, "files")
    assert Sample().events == [] and Sample().label is None


@pytest.mark.gpu
def test_default_workflow_on_two_abf_files(tmp_path):
    """Experiment.parse with the reference's defaults (detector at 90 pA, first-order 2 kHz Bessel filter, SpeedyStatSplit
    with prior_segments_per_second=10 and cutoff_freq=2000) on two synthetic .abf files.  Per file the events are
    filtered and then segmented together (File.parse_events); the result equals the reference's loop -- event.filter();
    event.parse(segmenter) one event at a time -- and the oracle's parse of the same rounded filtered current."""
    paths = []
    for f in range(2):
        counts, _ = synth.file_trace_counts(1_600_000, 40 + f)
        path = os.path.join(str(tmp_path), "run%d.abf" % f)
        abf.write_abf(path, counts.astype(np.int16))
        paths.append(path)
    exp = Experiment(paths)
    exp.parse(verbose=False)
    assert exp.n == 2 and len(exp.events) >= 3
    seg_kw = dict(prior_segments_per_second=10, cutoff_freq=2000.)
    for file in exp.files:
        assert os.path.basename(file.filename).startswith("run") and file.event_parser.threshold == 90
        starts, lens = oracle.lambda_events(np.asarray(file.current), threshold=90.0)
        assert [int(round(e.start * file.second)) for e in file.events] == list(starts)
        for ev, a, n in zip(file.events, starts, lens):
            assert ev.filtered and ev.filter_order == 1 and ev.filter_cutoff == 2000 and ev.n > 1
            one = Event(current=np.array(file.current[a:a + n]), start=ev.start, end=ev.end, duration=ev.duration,
                        second=file.second, file=file)
            one.filter(1, 2000)
            np.testing.assert_array_equal(one.current, ev.current)
            one.parse(SpeedyStatSplit(**seg_kw))
            got = [int(round(s.start * file.second)) for s in ev.segments]
            assert got == [int(round(s.start * file.second)) for s in one.segments]
            rounded, step, _ = one._on_fine_grid()
            ref = oracle.parse(rounded, **seg_kw)
            np.testing.assert_array_equal(got[1:], ref)
            seg = ev.segments[len(ev.segments) // 2]
            i, j = int(round(seg.start * file.second)), int(round(seg.end * file.second))
            assert seg.event is ev and seg.mean == pytest.approx(float(np.mean(ev.current[i:j])), rel=1e-12)
    assert len(exp.segments) == sum(e.n for e in exp.events)


@pytest.mark.gpu
def test_filtered_batch_on_a_real_header_scale_with_offset(tmp_path):
    """The device-resident chain (parse_filtered_batch) on a file whose header gives a scale that is not a power of two
    and a non-zero offset: the counts are filtered, the offset is put back, and currents, segment boundaries and segment
    statistics equal those of event.filter(); event.parse() one event at a time.  An empty list of events is fine too."""
    counts, _ = synth.file_trace_counts(900_000, 77)
    path = os.path.join(str(tmp_path), "offset.abf")
    # 10 V / 0.0005 V/pA / 20 / 32768 = 0.0305... pA per count; the synthetic counts assume 2**-5, so the levels shift a
    # little -- what matters here is the arithmetic, not the physiology
    abf.write_abf(path, counts.astype(np.int16), adc_range=10.0, adc_resolution=32768, instrument_scale=0.0005,
                  signal_gain=20.0, instrument_offset=1.25, signal_offset=0.5)
    exp = Experiment([path])
    exp.parse(verbose=False)
    file = exp.files[0]
    assert file.n >= 1
    from pypore_amd.grid import grid_of
    assert grid_of(file.current)[2] != 0.0                                  # the reader reports an offset
    seg = SpeedyStatSplit(prior_segments_per_second=10, cutoff_freq=2000.)
    for ev in file.events:
        a = int(round(ev.start * file.second)); n = len(ev.current)
        one = Event(current=file.current[a:a + n], start=ev.start, end=ev.end, duration=ev.duration, second=file.second, file=file)
        one.filter(1, 2000)
        np.testing.assert_allclose(ev.current, one.current, rtol=0, atol=1e-9)
        one.parse(seg)
        assert [int(round(s.start * file.second)) for s in ev.segments] == [int(round(s.start * file.second)) for s in one.segments]
        assert ev.segments[0].mean == pytest.approx(one.segments[0].mean, rel=1e-12)
    assert seg.parse_filtered_batch([]) == []


@pytest.mark.gpu
def test_filter_again_on_a_parked_current_keeps_the_files_offset(tmp_path):
    """parse_events(filter_params) leaves the filtered current of every event parked on the device WITHOUT the
    file's offset (grid.Deferred.from_tensor(y, off)).  event.filter() right after it -- before anybody read
    event.current -- must filter that tensor and put the offset back, like the same two filters on the host values."""
    counts, _ = synth.file_trace_counts(900_000, 78)
    path = os.path.join(str(tmp_path), "offset2.abf")
    abf.write_abf(path, counts.astype(np.int16), adc_range=10.0, adc_resolution=32768, instrument_scale=0.0005,
                  signal_gain=20.0, instrument_offset=1.25, signal_offset=0.5)
    exp = Experiment([path])
    exp.parse(verbose=False)
    file = exp.files[0]
    from pypore_amd.grid import Deferred, grid_of
    assert file.n >= 1 and grid_of(file.current)[2] != 0.0
    was_parked = []
    for ev in file.events:
        cur = ev.__dict__.get("current")
        a = int(round(ev.start * file.second)); n = int(round(ev.duration * file.second))
        parked = isinstance(cur, Deferred) and cur.tensor is not None
        was_parked.append(parked)
        ev.filter(1, 2000)                                           # second pass, straight from the parked tensor
        ref = Event(current=file.current[a:a + n], start=ev.start, end=ev.end, duration=ev.duration, second=file.second, file=file)
        ref.filter(1, 2000)
        _ = ref.current                                              # (read: written out, no longer parked)
        ref.filter(1, 2000)
        np.testing.assert_allclose(ev.current, ref.current, rtol=0, atol=1e-9)
        assert abs(float(np.mean(ev.current)) - float(np.mean(file.current[a:a + n]))) < 1.0, parked
    assert any(was_parked)                                           # the path this test is about was taken


@pytest.mark.gpu
def test_experiment_currents_are_written_out_only_when_read(tmp_path):
    """Experiment.parse never builds the file's float64 array nor copies a filtered current back (grid.Deferred behind the
    `current` attribute); reading them afterwards gives what the eager route gives, and the file's counts went up once."""
    from pypore_amd.core import raw_current
    from pypore_amd.grid import Deferred, GridArray
    counts, _ = synth.file_trace_counts(1_200_000, 91)
    path = os.path.join(str(tmp_path), "lazy.abf")
    abf.write_abf(path, counts.astype(np.int16))
    before = Deferred.live_device_bytes
    exp = Experiment([path])
    exp.parse(verbose=False)
    file = exp.files[0]
    root = raw_current(file)
    assert isinstance(root, Deferred) and not root.built and root.__dict__.get('_dev_counts') is not None
    assert file.n >= 1
    for ev in file.events:
        cur = raw_current(ev)
        assert isinstance(cur, Deferred) and not cur.built and cur.tensor is not None and cur.tensor.is_cuda
        assert all(isinstance(raw_current(s), Deferred) for s in ev.segments)
        assert sum(s.n for s in ev.segments) == len(cur)
    assert Deferred.live_device_bytes > before                              # filtered currents and the file's counts
    ev = file.events[0]
    a, n = int(round(ev.start * file.second)), len(raw_current(ev))
    got = ev.current                                                        # one copy of the event's float64 current
    assert isinstance(got, np.ndarray) and raw_current(ev) is got and got.shape == (n,)
    seg = ev.segments[1]
    i, j = int(round(seg.start * file.second)), int(round(seg.end * file.second))
    assert np.shares_memory(seg.current, got) and np.array_equal(seg.current, got[i:j])
    assert not root.built                                                   # still nobody asked for the file's array
    one = Event(current=np.array(file.current[a:a + n]), start=ev.start, end=ev.end, duration=ev.duration,
                second=file.second, file=file)
    assert root is not raw_current(file) and isinstance(file.current, GridArray)
    one.filter(1, 2000)
    np.testing.assert_array_equal(one.current, got)
    exp.delete()


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["exact", "exact_on_near_tie"])
def test_default_workflow_with_the_exact_route_equals_the_oracle_event_by_event(tmp_path, mode):
    """Experiment.parse (detector at 90 pA, first-order 2 kHz Bessel filter) with a segmenter built for the reference's own
    arithmetic on filtered events, SpeedyStatSplit(off_grid="exact" / "exact_on_near_tie"): File.parse_events filters on the
    device and segments the filtered float64 currents through ps_segment_exact_f64 (one call per group of events; with
    "exact_on_near_tie" only the groups whose fast run counted a near tie).  Every event's boundaries equal the oracle's on the
    filtered current the user sees (ev.current, the file's offset included) -- with and without cutoff_freq in the segmenter
    (without it a smooth current is cut every ~130 samples and near ties are common: the exact route is what decides them
    like the reference) -- and equal what event.filter(); event.parse() gives one event at a time."""
    counts, _ = synth.file_trace_counts(1_200_000, 52)
    path = os.path.join(str(tmp_path), "exact.abf")
    abf.write_abf(path, counts.astype(np.int16), adc_range=10.0, adc_resolution=32768, instrument_scale=0.0005,
                  signal_gain=20.0, instrument_offset=1.25, signal_offset=0.5)
    for seg_kw in (dict(prior_segments_per_second=10, cutoff_freq=2000.), dict(prior_segments_per_second=10)):
        exp = Experiment([path])
        exp.parse(segmenter=SpeedyStatSplit(off_grid=mode, **seg_kw), verbose=False)
        file = exp.files[0]
        assert file.n >= 1
        for ev in file.events:
            assert ev.filtered and ev.n >= 1
            got = [int(round(s.start * file.second)) for s in ev.segments[1:]]
            ref = oracle.parse(np.asarray(ev.current, dtype=np.float64), **seg_kw)
            np.testing.assert_array_equal(got, ref, err_msg="%s %s" % (mode, seg_kw))
            a, n = int(round(ev.start * file.second)), len(ev.current)
            one = Event(current=file.current[a:a + n], start=ev.start, end=ev.end, duration=ev.duration, second=file.second, file=file)
            one.filter(1, 2000)
            one.parse(SpeedyStatSplit(off_grid=mode, **seg_kw))
            assert [int(round(s.start * file.second)) for s in one.segments[1:]] == got
        exp.delete()
