"""File / Event containers: the call contract of DataTypes.py that sits on either side of the
segmenter (File.parse :589-602, Event.parse :276-289,:333, Event.filter :258-274) and the JSON
persistence of the results (Event :480-545, File :683-796).  Plotting, HMM
merging, MySQL and Experiment are out of scope.
"""
import json

import numpy as np

from .core import MetaSegment, Segment, _jsonable, ignored
from .parsers import SpeedyStatSplit, lambda_event_parser, parser as _parser_base


def _json_dict(d):
    """numpy scalars -> Python numbers (the reference relied on Python-2 json accepting them)."""
    return {k: _jsonable(v) for k, v in d.items()}


class MetaEvent(MetaSegment):
    """DataTypes.py:49-80: an event without its current."""

    def __init__(self, **kwargs):
        MetaSegment.__init__(self, **kwargs)

    def delete(self):
        with ignored(AttributeError):
            del self.state_parser
        for segment in getattr(self, "segments", []):
            segment.delete()
        with ignored(AttributeError):
            del self.segments
        del self


class Event(Segment):
    """DataTypes.py:239-256."""

    def __init__(self, current, segments=[], **kwargs):
        if len(segments) > 0:
            try:
                current = np.concatenate([seg.current for seg in segments])
            except Exception:
                current = []
        Segment.__init__(self, current, filtered=False, segments=segments, **kwargs)

    def filter(self, order=1, cutoff=2000., quantum=None):
        """DataTypes.py:258-274: Bessel low-pass of the event's current, cutoff normalised by the Nyquist frequency
        of the file's sampling rate, applied forward and backward (scipy.signal.filtfilt semantics) on the device.
        Only the reference's default order (1) is implemented; other orders raise ValueError.  The result replaces
        `current` as float64, as in the reference.  It no longer lies on the ADC grid: see Event.parse for how a
        filtered event is segmented."""
        if type(self) != Event:
            raise TypeError("Cannot filter a metaevent. Must have the current.")
        from . import engine
        dev, q = engine.to_device_samples(self.current, quantum)
        out = engine.context(dev.device.index).filter_bessel(dev, q, cutoff=cutoff, sampling_freq=float(self.second), order=order)
        self.current = out.cpu().numpy()
        self.filtered = True
        self.filter_order = order
        self.filter_cutoff = cutoff

    def parse(self, parser=None, hmm=None):
        """DataTypes.py:276-289,:333: segments = parser.parse(current); each gets .event and is
        rescaled from samples to seconds with the file's sampling rate."""
        if parser is None:
            parser = SpeedyStatSplit(prior_segments_per_second=10)
        if hmm is not None:
            raise NotImplementedError("HMM-guided merging needs yahmm (out of scope)")
        if getattr(self, "filtered", False):
            # A filtered current is float64 off the ADC grid; the device segmenter works on exact integer sums.  The
            # current is centred on its mean (the gains are shift invariant) and rounded to the finest power-of-two
            # grid that keeps every count below 2**22 (2**-18 pA for a 100 pA range): on the golden vectors recorded
            # from the reference that reproduces every boundary the reference finds on the unrounded float64 current
            # (tests/test_filter.py); coarser grids do not -- a heavily smoothed current has almost no variance left.
            # Segments keep views of the unrounded current and take their statistics from those views (numpy).
            cur = np.asarray(self.current, dtype=np.float64)
            c0 = float(np.mean(cur)) if cur.size else 0.0
            span = float(np.max(np.abs(cur - c0))) if cur.size else 0.0
            fq = 2.0 ** (int(np.ceil(np.log2(span * 1.01))) - 22) if span > 0 else 1.0
            centre = np.rint(c0 / fq) * fq
            self.segments = parser.parse(np.rint((cur - centre) / fq) * fq)
            for segment in self.segments:
                segment.current = self.current[int(segment.start):int(segment.end)]
                segment.__dict__.pop('_gpu_stats', None)
        else:
            self.segments = parser.parse(self.current)
        for segment in self.segments:
            segment.event = self
            segment.scale(float(self.file.second))
        self.state_parser = parser

    @property
    def n(self):
        try:
            return len(self.segments)
        except Exception:
            return 0

    # ---- persistence (DataTypes.py:335-347, :480-545) ------------------------------------------------
    def delete(self):
        with ignored(AttributeError):
            del self.current
        with ignored(AttributeError):
            del self.state_parser
        for segment in getattr(self, "segments", []):
            segment.delete()
        with ignored(AttributeError):
            del self.segments
        del self

    def to_meta(self):
        """:480-491: freeze the statistics, drop the current (also of the segments), become a MetaEvent."""
        for prop in ['mean', 'std', 'duration', 'start', 'min', 'max', 'end', 'start']:
            with ignored(AttributeError, KeyError, ValueError):
                self.__dict__[prop] = getattr(self, prop)
        with ignored(AttributeError):
            del self.current
        self.__dict__.pop('_gpu_stats', None)
        for segment in self.segments:
            segment.to_meta()
        self.__class__ = type("MetaEvent", (MetaEvent,), self.__dict__)

    def to_dict(self):
        keys = ['mean', 'std', 'min', 'max', 'start', 'end', 'duration', 'filtered',
                'filter_order', 'filter_cutoff', 'n', 'state_parser', 'segments']
        d = {}
        for i in keys:
            with ignored(AttributeError, ValueError):
                d[i] = getattr(self, i)
        d['name'] = self.__class__.__name__
        return d

    def to_json(self, filename=None):
        d = self.to_dict()
        with ignored(KeyError, AttributeError):
            d['segments'] = [seg.to_dict() for seg in d['segments']]
        with ignored(KeyError, AttributeError):
            d['state_parser'] = d['state_parser'].to_dict()
        _json = json.dumps(_json_dict(d), indent=4, separators=(',', ' : '))
        if filename:
            with open(filename, 'w') as out:
                out.write(_json)
        return _json

    @classmethod
    def from_json(cls, _json):
        """:516-529: a JSON without `current` gives a MetaEvent carrying the stored attributes."""
        if _json.endswith(".json"):
            with open(_json, 'r') as infile:
                _json = ''.join(line for line in infile)
        d = json.loads(_json)
        event = MetaSegment()
        if 'current' not in d.keys():
            event.__class__ = type("MetaEvent", (MetaEvent,), d)
        else:
            event = cls(d['current'], start=d['start'])
        return event

    @classmethod
    def from_segments(cls, segments):
        """:532-545."""
        try:
            current = np.concatenate([seg.current for seg in segments])
            return cls(current=current, start=0, segments=segments)
        except AttributeError:
            dur = sum(seg.duration for seg in segments)
            mean = np.mean([seg.mean * seg.duration for seg in segments]) / dur
            std = np.sqrt(sum(seg.std ** 2 * seg.duration for seg in segments) / dur)
            self = cls(current=np.array([seg.mean for seg in segments]), start=0, segments=segments, mean=mean, std=std)
            self.__class__ = type("MetaEvent", (Event,), self.__dict__)
            return self


class File(Segment):
    """DataTypes.py:567-602."""

    def __init__(self, filename=None, current=None, timestep=None, **kwargs):
        if current is not None and timestep is not None:
            filename = ""
        elif filename and current is None and timestep is None:
            from .abf import read_abf
            timestep, current = read_abf(filename)
            filename = filename.split("\\")[-1].split(".abf")[0]
        else:
            raise SyntaxError("Must provide current and timestep, or filename corresponding to a valid abf file.")
        Segment.__init__(self, current=current, filename=filename, second=1000. / timestep, events=[], sample=None)

    def __getitem__(self, index):
        return self.events[index]

    def parse(self, parser=None):
        """DataTypes.py:589-602."""
        if parser is None:
            parser = lambda_event_parser(threshold=90)
        self.events = [Event(current=seg.current,
                             start=seg.start / self.second,
                             end=(seg.start + seg.duration) / self.second,
                             duration=seg.duration / self.second,
                             second=self.second,
                             file=self) for seg in parser.parse(self.current)]
        self.event_parser = parser

    def parse_events(self, parser=None):
        """Segments every event of the file in ONE device call (Experiment.parse inner loop,
        DataTypes.py:978-984, without the optional filter)."""
        if parser is None:
            parser = SpeedyStatSplit(prior_segments_per_second=10)
        if hasattr(parser, "parse_batch"):
            all_segs = parser.parse_batch([ev.current for ev in self.events])
        else:
            all_segs = [parser.parse(ev.current) for ev in self.events]
        for ev, segs in zip(self.events, all_segs):
            ev.segments = segs
            for segment in segs:
                segment.event = ev
                segment.scale(float(self.second))
            ev.state_parser = parser

    @property
    def n(self):
        return len(self.events)

    # ---- persistence (DataTypes.py:611-626, :683-796) ------------------------------------------------
    def delete(self):
        with ignored(AttributeError):
            del self.current
        with ignored(AttributeError):
            del self.event_parser
        for event in self.events:
            event.delete()
        del self

    def to_meta(self):
        with ignored(AttributeError):
            del self.current
        for event in self.events:
            event.to_meta()

    def to_dict(self):
        keys = ['filename', 'n', 'event_parser', 'mean', 'std', 'duration', 'start', 'end', 'events']
        if not hasattr(self, 'end') and (hasattr(self, 'start') and hasattr(self, 'duration')):
            setattr(self, 'end', self.start + self.duration)
        d = {}
        for i in keys:
            with ignored(AttributeError, ValueError):
                d[i] = getattr(self, i)
        d['name'] = self.__class__.__name__
        return d

    def to_json(self, filename=None):
        """:708-736: the file, its event parser, every event with its segments and state parser."""
        d = self.to_dict()
        devents = []
        for event in d['events']:
            devent = event.to_dict()
            try:
                devent['segments'] = [_json_dict(state.to_dict()) for state in devent['segments']]
                devent['state_parser'] = devent['state_parser'].to_dict()
            except Exception:
                with ignored(KeyError, AttributeError):
                    del devent['segments']
                    del devent['state_parser']
            devents.append(_json_dict(devent))
        d['events'] = devents
        d['event_parser'] = d['event_parser'].to_dict()
        _json = json.dumps(_json_dict(d), indent=4, separators=(',', ' : '))
        if filename:
            with open(filename, 'w') as outfile:
                outfile.write(_json)
        return _json

    @classmethod
    def from_json(cls, _json):
        """:739-796: rebuilds the file and its events; with the .abf at hand the events and segments get views of the
        current again (filtered events are re-filtered), otherwise everything comes back as Meta* objects."""
        if _json.endswith(".json"):
            with open(_json, 'r') as infile:
                _json = ''.join(line for line in infile)
        d = json.loads(_json)
        if d['name'] != "File":
            raise TypeError("JSON does not encode a file")
        try:
            file = File(filename=d['filename'] + ".abf")
            meta = False
        except Exception:
            file = File(current=[], timestep=1)
            meta = True
        file.event_parser = _parser_base.from_json(json.dumps(d['event_parser']))
        file.events = []
        for ej in d['events']:
            s, e = int(ej['start'] * file.second), int(ej['end'] * file.second)
            if meta:
                event = MetaEvent(**ej)
            else:
                event = Event(current=file.current[s:e], start=s / file.second, end=e / file.second,
                              duration=(e - s) / file.second, second=file.second, file=file)
            if ej['filtered']:
                if not meta:
                    event.filter(order=ej['filter_order'], cutoff=ej['filter_cutoff'])
            if meta:
                event.segments = [MetaSegment(**sj) for sj in ej['segments']]
            else:
                event.segments = [Segment(current=event.current[int(sj['start'] * file.second):int(sj['end'] * file.second)],
                                          second=file.second, event=event, **sj)
                                  for sj in ej['segments']]
            event.state_parser = _parser_base.from_json(json.dumps(ej['state_parser']))
            event.filtered = ej['filtered']
            file.events.append(event)
        return file
