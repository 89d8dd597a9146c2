"""File / Event containers: the call contract of DataTypes.py that sits on either side of the
segmenter (File.parse :589-602, Event.parse :276-289,:333).  Plotting, HMM merging, JSON/MySQL
persistence and Experiment are out of scope.
"""
import numpy as np

from .core import Segment
from .parsers import SpeedyStatSplit, lambda_event_parser


class Event(Segment):
    """DataTypes.py:239-256."""

    def __init__(self, current, segments=[], **kwargs):
        if len(segments) > 0:
            try:
                current = np.concatenate([seg.current for seg in segments])
            except Exception:
                current = []
        Segment.__init__(self, current, filtered=False, segments=segments, **kwargs)

    def parse(self, parser=None, hmm=None):
        """DataTypes.py:276-289,:333: segments = parser.parse(current); each gets .event and is
        rescaled from samples to seconds with the file's sampling rate."""
        if parser is None:
            parser = SpeedyStatSplit(prior_segments_per_second=10)
        if hmm is not None:
            raise NotImplementedError("HMM-guided merging needs yahmm (out of scope)")
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
