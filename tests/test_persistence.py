"""JSON persistence of the results: Event / File to_dict, to_json, from_json, to_meta, from_segments
(DataTypes.py:480-545, :683-796).  The first tests are host-side only (the segmentation results are put in by hand); the
last one (-m gpu) sends what the DEVICE produced through to_json -> from_json -> MemoryParse."""
import json
import os

import numpy as np
import pytest

from pypore_amd import abf, synth
from pypore_amd.core import MetaSegment, Segment
from pypore_amd.DataTypes import Event, File, MetaEvent
from pypore_amd.parsers import SpeedyStatSplit, lambda_event_parser


def _file_with_results(tmp_path):
    counts = synth.file_trace_counts(400000, 5)[0] if isinstance(synth.file_trace_counts(1000, 5), tuple) else synth.file_trace_counts(400000, 5)
    path = os.path.join(str(tmp_path), "run1.abf")
    abf.write_abf(path, np.asarray(counts, dtype=np.int16))
    f = File(filename=path[:-4] + ".abf")
    sec = f.second
    f.event_parser = lambda_event_parser(threshold=90)
    spans = [(60000, 210000), (250000, 390000)]
    cuts = [[20000, 75000], [40000]]
    for (s, e), cs in zip(spans, cuts):
        ev = Event(current=f.current[s:e], start=s / sec, end=e / sec, duration=(e - s) / sec, second=sec, file=f)
        edges = [0] + cs + [e - s]
        ev.segments = [Segment(current=ev.current[a:b], start=a / sec, end=b / sec, duration=(b - a) / sec,
                               second=sec, event=ev) for a, b in zip(edges, edges[1:])]
        ev.state_parser = SpeedyStatSplit(prior_segments_per_second=10)
        f.events.append(ev)
    return f, path


def test_file_json_round_trip_with_the_abf_at_hand(tmp_path):
    f, path = _file_with_results(tmp_path)
    js = f.to_json(os.path.join(str(tmp_path), "run1.json"))
    d = json.loads(js)
    assert d["name"] == "File" and d["n"] == 2 and d["event_parser"]["name"] == "lambda_event_parser"
    assert [len(ev["segments"]) for ev in d["events"]] == [3, 2]
    assert d["events"][0]["state_parser"]["prior_segments_per_second"] == 10
    seg0 = f.events[0].segments[0]
    assert d["events"][0]["segments"][0]["mean"] == pytest.approx(float(np.mean(seg0.current)))
    d["filename"] = path[:-4]                         # from_json appends ".abf" to the stored name (DataTypes.py:757)
    g = File.from_json(json.dumps(d))
    assert type(g) is File and g.n == 2 and g.event_parser.threshold == 90
    for ev_f, ev_g in zip(f.events, g.events):
        assert type(ev_g) is Event and not ev_g.filtered
        np.testing.assert_array_equal(ev_g.current, ev_f.current)
        assert ev_g.state_parser.prior_segments_per_second == 10
        for a, b in zip(ev_f.segments, ev_g.segments):
            assert (a.start, a.end, a.duration) == (b.start, b.end, b.duration)
            np.testing.assert_array_equal(a.current, b.current)
    # reading the file written to disk gives the same thing as the string
    on_disk = json.load(open(os.path.join(str(tmp_path), "run1.json")))
    assert on_disk == json.loads(js)
    on_disk["filename"] = path[:-4]
    json.dump(on_disk, open(os.path.join(str(tmp_path), "run1b.json"), "w"))
    h = File.from_json(os.path.join(str(tmp_path), "run1b.json"))           # a *.json path is read from disk (:745-747)
    assert h.filename == f.filename and h.n == 2 and [ev.n for ev in h.events] == [3, 2]


def test_file_json_round_trip_without_the_abf_gives_meta_objects(tmp_path):
    f, _ = _file_with_results(tmp_path)
    means = [[float(s.mean) for s in ev.segments] for ev in f.events]
    d = json.loads(f.to_json())
    d["filename"] = "/nonexistent/run1"               # no .abf to re-read: everything comes back as Meta* (DataTypes.py:759-761)
    g = File.from_json(json.dumps(d))
    assert all(isinstance(ev, MetaEvent) for ev in g.events)
    for ev, ms in zip(g.events, means):
        assert [s.mean for s in ev.segments] == pytest.approx(ms)
        assert all(isinstance(s, MetaSegment) for s in ev.segments)
    with pytest.raises(TypeError):
        File.from_json(json.dumps(dict(name="Event")))
    f.to_meta()                                       # :683-693, :480-491: currents dropped, statistics frozen
    assert not hasattr(f, "current") and all(type(ev).__name__ == "MetaEvent" for ev in f.events)
    assert all(type(s).__name__ == "MetaSegment" for ev in f.events for s in ev.segments)
    assert [[s.mean for s in ev.segments] for ev in f.events] == [pytest.approx(m) for m in means]


def test_event_json_and_from_segments(tmp_path):
    f, _ = _file_with_results(tmp_path)
    ev = f.events[0]
    d = json.loads(ev.to_json())
    assert d["name"] == "Event" and d["n"] == 3 and d["filtered"] is False and len(d["segments"]) == 3
    assert d["state_parser"]["name"] == "SpeedyStatSplit"
    m = Event.from_json(ev.to_json())
    assert isinstance(m, MetaEvent) and m.n == 3 and m.mean == pytest.approx(float(ev.mean))
    whole = Event.from_segments(ev.segments)                     # segments with current: concatenated
    np.testing.assert_array_equal(whole.current, ev.current)
    metas = [MetaSegment(mean=s.mean, std=s.std, duration=s.duration, start=s.start) for s in ev.segments]
    # metadata only (:538-545): duration-weighted mean and pooled variance of the segments (the reference's own result is
    # an Event with an empty current, its formulas never reach the object: DataTypes.from_segments documents the choice)
    me = Event.from_segments(metas)
    dur = sum(s.duration for s in metas)
    assert type(me).__name__ == "MetaEvent" and me.n == 3 and me.duration == pytest.approx(dur)
    assert me.mean == pytest.approx(sum(s.mean * s.duration for s in metas) / dur)
    assert me.segments[1].mean == pytest.approx(float(ev.segments[1].mean))


def test_reference_readme_example_loads_and_round_trips():
    """The one stored analysis the reference itself shows (README.md:346-391, a File.to_json of its own; the snippet
    is cut after the second segment of the first event, tests/golden/readme_file.json closes the brackets and changes
    nothing else): it loads as Meta* objects (no .abf at hand) and every key and value comes back out of to_json.
    Only the file's `n` is left out: the snippet says 16 events and shows one."""
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "readme_file.json")
    ref = json.load(open(here))
    f = File.from_json(here)
    assert f.filename == "13823006-s06" and f.event_parser.threshold == 50.0 and type(f.event_parser) is lambda_event_parser
    ev = f.events[0]
    assert isinstance(ev, MetaEvent) and ev.end == 31.26803 and ev.std == 1.9335278997265508
    sp = ev.state_parser
    assert type(sp) is SpeedyStatSplit and (sp.min_gain_per_sample, sp.min_width, sp.window_width, sp.max_width) == (0.5, 1000, 10000, 1000000)
    assert [type(s) for s in ev.segments] == [MetaSegment, MetaSegment] and ev.segments[1].mean == 24.084380592526145
    f.duration = ref["duration"]                     # (file-level numbers are attributes the caller owns: :708-736 writes what is there)
    out = json.loads(f.to_json())

    def contained(a, b, path="file"):
        if isinstance(a, dict):
            for k, v in a.items():
                if path == "file" and k == "n":
                    continue
                assert k in b, (path, k)
                contained(v, b[k], path + "." + k)
        elif isinstance(a, list):
            assert len(a) == len(b), path
            for i, (x, y) in enumerate(zip(a, b)):
                contained(x, y, "%s[%d]" % (path, i))
        elif path.endswith('.name'):
            assert b in (a, 'Meta' + a), (path, a, b)    # without the .abf the objects are the Meta* variants
        else:
            assert a == b, (path, a, b)
    contained(ref, out)


@pytest.mark.gpu
def test_device_results_round_trip_through_json_and_replay_with_memoryparse(tmp_path):
    """SURVEY 8 f-4 on what the device produced (VERDICT r5 next #3a): File.parse + File.parse_events on a synthetic .abf --
    detection and segmentation on the GPU, filtered currents still PARKED on the device (grid.Deferred) when to_json asks for
    the statistics -- -> to_json -> File.from_json with the .abf at hand (views of the current again, filtered events filtered
    again: DataTypes.py:753-790) and without it (Meta objects from the stored numbers) -> replay of the stored split points
    with MemoryParse (parsers.py:110-122, as File.from_database does: DataTypes.py:839-846): identical events, boundaries and
    means at every stage."""
    from pypore_amd.core import raw_current
    from pypore_amd.grid import Deferred
    from pypore_amd.parsers import MemoryParse
    counts, _ = synth.file_trace_counts(1_600_000, 40)
    path = os.path.join(str(tmp_path), "run0.abf")
    abf.write_abf(path, counts.astype(np.int16))
    seg_kw = dict(prior_segments_per_second=10, cutoff_freq=2000.)

    def analysed(filter_params):
        f = File(filename=path)
        f.parse(lambda_event_parser(threshold=90))
        f.parse_events(SpeedyStatSplit(**seg_kw), filter_params=filter_params)
        return f

    for filter_params in ((1, 2000), None):
        f = analysed(filter_params)
        rate = f.second
        assert f.n >= 2 and all(ev.n > 1 for ev in f.events)
        if filter_params is not None:
            # the case the review names: currents that have not left the device when the JSON is written
            parked = [isinstance(raw_current(ev), Deferred) and raw_current(ev).tensor is not None and raw_current(ev).tensor.is_cuda
                      for ev in f.events]
            assert all(parked), parked
        js = f.to_json(os.path.join(str(tmp_path), "run0.json"))
        d = json.loads(js)
        assert d["n"] == f.n and [len(e["segments"]) for e in d["events"]] == [ev.n for ev in f.events]
        assert all(e["filtered"] is (filter_params is not None) for e in d["events"])
        assert d["events"][0]["state_parser"]["name"] == "SpeedyStatSplit" and d["events"][0]["state_parser"]["cutoff_freq"] == 2000.0
        # the stored statistics are those of the (filtered) float64 current the user sees
        for ev, ej in zip(f.events, d["events"]):
            cur = np.asarray(ev.current)
            for sg, sj in zip(ev.segments, ej["segments"]):
                i, j = int(round(sg.start * rate)), int(round(sg.end * rate))
                assert sj["mean"] == pytest.approx(float(np.mean(cur[i:j])), rel=1e-9, abs=1e-9)
                assert sj["std"] == pytest.approx(float(np.std(cur[i:j])), rel=1e-6, abs=1e-9)
                assert (sj["start"], sj["end"], sj["duration"]) == (sg.start, sg.end, sg.duration)
        # ---- with the .abf at hand: Event / Segment objects on views of the file's current
        d_abf = dict(d, filename=path[:-4])
        g = File.from_json(json.dumps(d_abf))
        assert type(g) is File and g.n == f.n and g.event_parser.threshold == 90
        for ev_f, ev_g in zip(f.events, g.events):
            assert type(ev_g) is Event and bool(ev_g.filtered) is (filter_params is not None)
            a, b = int(ev_f.start * rate), int(ev_f.end * rate)               # (the reference's own truncation, :765)
            assert (ev_g.start, ev_g.end) == (a / rate, b / rate) and len(ev_g.current) == b - a
            if filter_params is None:
                np.testing.assert_array_equal(np.asarray(ev_g.current), np.asarray(f.current[a:b]))
            else:
                assert (ev_g.filter_order, ev_g.filter_cutoff) == (1, 2000)
                if (a, b) == (int(round(ev_f.start * rate)), int(round(ev_f.end * rate))):
                    np.testing.assert_allclose(np.asarray(ev_g.current), np.asarray(ev_f.current), rtol=0, atol=1e-9)   # filtered again: same filter, same samples
            assert ev_g.n == ev_f.n and type(ev_g.state_parser) is SpeedyStatSplit and ev_g.state_parser.cutoff_freq == 2000.0
            for sf, sg in zip(ev_f.segments, ev_g.segments):
                assert type(sg) is Segment and sg.event is ev_g
                assert (sg.start, sg.end, sg.duration) == (sf.start, sf.end, sf.duration)
                i, j = int(sf.start * rate), int(sf.end * rate)
                np.testing.assert_array_equal(np.asarray(sg.current), np.asarray(ev_g.current[i:j]))
                if (i, j) == (int(round(sf.start * rate)), int(round(sf.end * rate))) and len(ev_g.current) == len(ev_f.current):
                    assert sg.mean == pytest.approx(float(sf.mean), rel=1e-9, abs=1e-9)
        # ---- without it: Meta objects carrying the stored numbers
        h = File.from_json(json.dumps(dict(d, filename="/nonexistent/run0")))
        assert all(isinstance(ev, MetaEvent) for ev in h.events) and h.n == f.n
        for ev_f, ev_h in zip(f.events, h.events):
            assert [s.mean for s in ev_h.segments] == pytest.approx([float(s.mean) for s in ev_f.segments], rel=1e-12)
            assert [s.start for s in ev_h.segments] == [s.start for s in ev_f.segments]
            assert ev_h.mean == pytest.approx(float(ev_f.mean), rel=1e-12) and ev_h.n == ev_f.n
        # ---- MemoryParse: the stored split points replayed on a fresh File (File.from_database's route)
        r = File(filename=path)
        ev_starts = [int(round(e["start"] * rate)) for e in d["events"]]
        ev_ends = [int(round(e["end"] * rate)) for e in d["events"]]
        r.parse(MemoryParse(ev_starts, ev_ends))
        assert [(int(round(ev.start * rate)), int(round((ev.start + ev.duration) * rate))) for ev in r.events] == list(zip(ev_starts, ev_ends))
        for ev_r, ev_f, ej in zip(r.events, f.events, d["events"]):
            np.testing.assert_array_equal(np.asarray(ev_r.current), np.asarray(f.current[int(round(ev_f.start * rate)):int(round(ev_f.end * rate))]))
            if filter_params is not None:
                ev_r.filter(*filter_params)
                np.testing.assert_allclose(np.asarray(ev_r.current), np.asarray(ev_f.current), rtol=0, atol=1e-9)
            ev_r.filtered = False                       # (the replay cuts the current it is given: no re-quantisation, no device)
            ev_r.parse(MemoryParse([int(round(sj["start"] * rate)) for sj in ej["segments"]],
                                   [int(round(sj["end"] * rate)) for sj in ej["segments"]]))
            assert [s.start for s in ev_r.segments] == pytest.approx([s.start for s in ev_f.segments], abs=1e-12)
            assert [s.duration for s in ev_r.segments] == pytest.approx([s.duration for s in ev_f.segments], abs=1e-12)
            assert [s.n for s in ev_r.segments] == [s.n for s in ev_f.segments]
            assert [float(s.mean) for s in ev_r.segments] == pytest.approx([float(s.mean) for s in ev_f.segments], rel=1e-9, abs=1e-9)
            assert [float(s.std) for s in ev_r.segments] == pytest.approx([float(s.std) for s in ev_f.segments], rel=1e-6, abs=1e-9)
        # the device run itself is the oracle's (unfiltered route: the counts are exact)
        if filter_params is None:
            import oracle
            x = np.asarray(f.current, dtype=np.float64)
            for ev in f.events:
                a, b = int(round(ev.start * rate)), int(round(ev.end * rate))
                ref = oracle.parse(x[a:b], **seg_kw)
                np.testing.assert_array_equal([int(round(s.start * rate)) for s in ev.segments[1:]], ref)
