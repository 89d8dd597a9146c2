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
