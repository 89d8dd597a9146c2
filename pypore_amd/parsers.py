"""Parser plug-ins: the objects File.parse / Event.parse accept (`parse(current) -> list[Segment]`).

Surface kept from the reference (PyPore/parsers.py; SURVEY.md 8 a7, a11, b): `SpeedyStatSplit` with its eight
constructor keywords stored as attributes of the same names (:505-534) -- they are also its JSON schema --,
`lambda_event_parser(threshold, rules)` (:124-155), `MemoryParse(starts, ends)` (:110-122) and the base `parser`
with `to_dict` / `to_json` / `from_json` / `parse` (:34-107).  The Qt GUI hooks are UI and out of scope.

How it is built here: every subclass of `parser` registers itself by name (that is what `from_json` resolves),
a parser's JSON is its public scalar attributes, and the two parsers that compute -- SpeedyStatSplit and the
event detector with its default rules -- hand the samples to the HIP kernels through pypore_amd.engine.
"""
import numbers

import numpy as np

from .core import Segment, dump_json, load_json, plain
from .cparsers import FastStatSplit

_REGISTRY = {}


class parser(object):
    """Base of the plug-ins.  Its own parse() returns the whole array as one segment."""

    def __init_subclass__(cls, **kwargs):
        super().__init_subclass__(**kwargs)
        _REGISTRY[cls.__name__] = cls

    def __repr__(self):
        return self.to_json()

    def to_dict(self):
        """Public attributes that are plain scalars (numbers, strings, None), plus the class name: what the reference
        writes for the parsers in scope (rule lambdas and GUI widgets never reach its JSON either)."""
        d = {k: plain(v) for k, v in vars(self).items()
             if not k.startswith('_') and k != 'param_dict'
             and (v is None or isinstance(plain(v), (numbers.Number, str)))}
        d['name'] = type(self).__name__
        return d

    def to_json(self, filename=False):
        return dump_json(self.to_dict(), filename or None)

    def parse(self, current):
        return [Segment(current=current, start=0, duration=current.shape[0] / 100000)]

    @classmethod
    def from_json(cls, _json):
        """Parser of the class named in the JSON (text or *.json path), built from the remaining keys."""
        d = dict(load_json(_json))
        kind = _REGISTRY.get(d.pop('name'))
        if kind is None:
            raise AttributeError("no parser of that name in pypore_amd.parsers")
        return kind(**d)


_REGISTRY['parser'] = parser


class MemoryParse(object):
    """Replays stored split points: one Segment (copy of the samples) per (start, end) pair, in samples."""

    def __init__(self, starts, ends):
        self.starts = starts
        self.ends = ends

    def parse(self, current):
        spans = ((int(s), int(e), s, e) for s, e in zip(self.starts, self.ends))
        return [Segment(current=np.array(current[i:j], copy=True), start=s, duration=e - s) for i, j, s, e in spans]


class lambda_event_parser(parser):
    """Threshold event detector.  The trace is cut wherever it crosses `threshold`; a piece is an event when every
    rule accepts it -- by default: longer than 100 000 samples, minimum above -0.5 pA, maximum below the threshold
    (the blockade side of the cut).  Start and duration of the returned Segments are in samples."""
    MIN_DURATION = 100000
    MIN_CURRENT = -0.5

    def __init__(self, threshold=90, rules=None):
        self.threshold = threshold
        self._builtin = rules is None
        self.rules = rules or [lambda event: event.duration > self.MIN_DURATION,
                               lambda event: event.min > self.MIN_CURRENT,
                               lambda event: event.max < self.threshold]

    def parse(self, current, quantum=None, device=None, offset=None):
        """Built-in rules: one streaming pass on the GPU (ps_detect_events: crossings, then min / max of the long
        pieces).  Custom rules are Python callables on Segment objects and run on the host over numpy pieces."""
        if self._builtin:
            return self._parse_device(current, quantum, device, offset)
        x = np.asarray(current)
        below = x < self.threshold
        cuts = np.flatnonzero(below[1:] != below[:-1]) + 1
        edges = np.concatenate(([0], cuts, [x.shape[0]]))
        pieces = (Segment(current=np.array(x[a:b]), start=a, duration=b - a) for a, b in zip(edges[:-1], edges[1:]))
        return [piece for piece in pieces if all(rule(piece) for rule in self.rules)]

    def _parse_device(self, current, quantum, device, offset):
        from . import engine
        s = engine.to_device(current, quantum, offset, device)
        # the kernel compares count * quantum with its arguments: take the offset to the other side
        starts, lens = engine.context(device).detect_events(
            s.tensor, s.quantum, threshold=float(self.threshold) - s.offset, min_duration=self.MIN_DURATION,
            min_current=self.MIN_CURRENT - s.offset)
        if hasattr(current, "is_cuda"):             # device tensor in: the events' values come back as pA
            host = s.tensor.cpu().numpy().astype(np.float64)
            if not s.tensor.dtype.is_floating_point:
                host = host * s.quantum
            host = host + s.offset
        else:
            from .grid import Deferred
            host = current if isinstance(current, (np.ndarray, Deferred)) else np.asarray(current)
        # (slices, not copies: a GridArray -- or a current that has not been written out -- keeps its counts that way
        #  and Event.parse stays on the int16 route)
        return [Segment(current=host[a:a + n], start=a, duration=n) for a, n in zip(starts.tolist(), lens.tolist())]


class SpeedyStatSplit(parser):
    """The drop-in segmenter: stores the reference's eight parameters under the reference's names and runs
    FastStatSplit -- here the HIP kernels -- on parse().  Extra keywords (kept out of the JSON): `quantum` / `offset`
    describe the ADC grid of float input (pA per count, pA at count 0; found automatically when omitted), `device`
    picks the GPU, `off_grid` says what happens to float input that lies on NO grid (the reference takes any float64
    buffer, cparsers.pyx:53): "raise" (default) ValueError -- nothing is rounded silently --, "requantise" rounds it
    on the device to the finest power-of-two grid that keeps the counts below 2**22, "exact" segments it with
    the reference's own arithmetic -- its sequential fp64 cumsums and var_c expressions, on the device: the reference's
    boundaries on the same input by construction, tens of milliseconds per 1e6 samples --, "exact_on_near_tie" takes the fast
    route and the exact one only where that counted a near tie.  The last two also apply to FILTERED events (Event.parse,
    File.parse_events), whose current lies on no grid either."""

    def __init__(self, min_width=100, max_width=1000000, window_width=10000,
                 min_gain_per_sample=None, false_positive_rate=None,
                 prior_segments_per_second=None, sampling_freq=1.e5, cutoff_freq=None,
                 quantum=None, device=None, offset=None, off_grid="raise"):
        self.min_width = min_width
        self.max_width = max_width
        self.min_gain_per_sample = min_gain_per_sample
        self.window_width = window_width
        self.prior_segments_per_second = prior_segments_per_second
        self.false_positive_rate = false_positive_rate
        self.sampling_freq = sampling_freq
        self.cutoff_freq = cutoff_freq
        from .cparsers import OFF_GRID_MODES
        if off_grid not in OFF_GRID_MODES:
            raise ValueError("off_grid must be one of %s" % ", ".join(repr(m) for m in OFF_GRID_MODES))
        self._grid = dict(quantum=quantum, device=device, offset=offset, off_grid=off_grid)

    def _fast(self, cutoff=True):
        return FastStatSplit(self.min_width, self.max_width, self.window_width, self.min_gain_per_sample,
                             self.false_positive_rate, self.prior_segments_per_second, self.sampling_freq,
                             self.cutoff_freq if cutoff else None, **self._grid)

    def parse(self, current):
        return self._fast().parse(current)

    def parse_batch(self, currents, levels=None):
        """All events of a file in one device call (extension; same result as [parse(c) for c in currents]).  levels: the
        level in pA that was subtracted from each event upstream (Event.parse of a filtered event), or None."""
        return self._fast().parse_batch(currents, levels)

    def parse_exact(self, current):
        """The exact route for one float64 current (extension; cparsers.FastStatSplit.parse_exact_batch)."""
        return self._fast().parse_exact_batch([current])[0]

    @property
    def off_grid(self):
        return self._grid["off_grid"]

    def parse_filtered_batch(self, currents, order=1, cutoff=2000., sampling_freq=None):
        """event.filter(order, cutoff); event.parse(self) for many events, on the device from end to end (extension)."""
        return self._fast().parse_filtered_batch(currents, order, cutoff, self.sampling_freq if sampling_freq is None else sampling_freq)

    def best_single_split(self, current):
        """(gain, index) of the best single split; like the reference wrapper, without cutoff_freq (:530-534)."""
        return self._fast(cutoff=False).best_single_split(current)
