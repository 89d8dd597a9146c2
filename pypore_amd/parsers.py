"""Parser plug-ins with the class surface of PyPore/parsers.py (Python 3).

In scope: the base `parser` protocol (parsers.py:34-107), `SpeedyStatSplit`
(parsers.py:505-565) -- the drop-in whose parse() runs on the MI355X --, `lambda_event_parser`
(parsers.py:124-155) and `MemoryParse` (parsers.py:110-122).  The Qt GUI hooks of the
reference are UI and out of scope.
"""
import json
import sys

import numpy as np

from .core import Segment
from .cparsers import FastStatSplit


class parser(object):
    """parsers.py:34-107 -- duck-typed protocol: parse(current) -> list[Segment]."""

    def __init__(self):
        pass

    def __repr__(self):
        return self.to_json()

    def to_dict(self):
        d = {key: val for key, val in self.__dict__.items()
             if key != 'param_dict' and not key.startswith('_')
             if type(val) in (int, float) or ('Qt' not in repr(val)) and 'lambda' not in repr(val)}
        d['name'] = self.__class__.__name__
        return d

    def to_json(self, filename=False):
        _json = json.dumps(self.to_dict(), indent=4, separators=(',', ' : '))
        if filename:
            with open(filename, 'w') as out:
                out.write(_json)
        return _json

    def parse(self, current):
        """parsers.py:57-59: the whole array as one segment."""
        return [Segment(current=current, start=0, duration=current.shape[0] / 100000)]

    @classmethod
    def from_json(cls, _json):
        """parsers.py:97-107: the class is looked up by its `name` entry in this module."""
        if _json.endswith(".json"):
            with open(_json, 'r') as infile:
                _json = ''.join(line for line in infile)
        d = json.loads(_json)
        name = d['name']
        del d['name']
        return getattr(sys.modules[__name__], name)(**d)


class MemoryParse(object):
    """parsers.py:110-122: replay stored split points."""

    def __init__(self, starts, ends):
        self.starts = starts
        self.ends = ends

    def parse(self, current):
        return [Segment(current=np.array(current[int(s):int(e)], copy=True), start=s, duration=(e - s))
                for s, e in zip(self.starts, self.ends)]


class lambda_event_parser(parser):
    """parsers.py:124-155: threshold event detector with rule filter.

    Events are maximal runs on one side of `threshold`; a run is kept when every rule holds
    (defaults: duration > 100000 samples, min > -0.5 pA, max < threshold)."""

    def __init__(self, threshold=90, rules=None):
        self.threshold = threshold
        self._rules0 = [lambda event: event.duration > 100000,
                        lambda event: event.min > -0.5,
                        lambda event: event.max < self.threshold]
        self.rules = rules or self._rules0

    def _lambda_select(self, events):
        return [event for event in events if np.all([rule(event) for rule in self.rules])]

    def _default_rules(self):
        return self.rules is self._rules0

    def parse(self, current, quantum=None, device=None):
        """With the default rules the detection runs on the GPU (ps_detect_events: mask, edges, pieces,
        per-piece min/max); custom rule lambdas need the pieces on the host and take the numpy route of
        the reference.  Either way the result is the reference's: one Segment per kept piece, `current` a
        copy, start/duration in samples."""
        if self._default_rules():
            from . import engine
            host = np.asarray(current) if not hasattr(current, "is_cuda") else None
            t, q = engine.to_device_samples(current, quantum, device)
            st, ln = engine.context(device).detect_events(t, q, threshold=float(self.threshold))
            if host is None:                        # device tensor in: fetch the kept pieces' values once
                host = t.cpu().numpy().astype(np.float64)
                if not t.dtype.is_floating_point:
                    host = host * q
            return [Segment(current=np.array(host[s:s + n]), copy=True, start=s, duration=n)
                    for s, n in zip(st.tolist(), ln.tolist())]
        current = np.asarray(current)
        mask = np.where(current < self.threshold, 1, 0)
        mask = np.abs(np.diff(mask))
        tics = np.concatenate(([0], np.where(mask == 1)[0] + 1, [current.shape[0]]))
        del mask
        events = [Segment(current=np.array(piece), copy=True, start=tics[i], duration=piece.shape[0])
                  for i, piece in enumerate(np.split(current, tics[1:-1]))]
        return [event for event in self._lambda_select(events)]


class SpeedyStatSplit(parser):
    """parsers.py:505-565: holds the eight constructor parameters as attributes of the same
    names (they ARE the JSON schema, parsers.py:42-48) and delegates to FastStatSplit, which
    here runs on the GPU.  Extra keyword `quantum` (pA per ADC count, power of two) skips the
    host-side grid detection; `device` picks the GPU."""

    def __init__(self, min_width=100, max_width=1000000, window_width=10000,
                 min_gain_per_sample=None, false_positive_rate=None,
                 prior_segments_per_second=None, sampling_freq=1.e5, cutoff_freq=None,
                 quantum=None, device=None):
        self.min_width = min_width
        self.max_width = max_width
        self.min_gain_per_sample = min_gain_per_sample
        self.window_width = window_width
        self.prior_segments_per_second = prior_segments_per_second
        self.false_positive_rate = false_positive_rate
        self.sampling_freq = sampling_freq
        self.cutoff_freq = cutoff_freq
        self._quantum = quantum
        self._device = device

    def _fast(self, with_cutoff=True):
        return FastStatSplit(self.min_width, self.max_width, self.window_width, self.min_gain_per_sample,
                             self.false_positive_rate, self.prior_segments_per_second, self.sampling_freq,
                             self.cutoff_freq if with_cutoff else None,
                             quantum=self._quantum, device=self._device)

    def parse(self, current):
        """parsers.py:524-528."""
        return self._fast().parse(current)

    def parse_batch(self, currents):
        """All events of a file in one device call (extension; same result as [parse(c) for c])."""
        return self._fast().parse_batch(currents)

    def best_single_split(self, current):
        """parsers.py:530-534 (the reference drops cutoff_freq here)."""
        return self._fast(with_cutoff=False).best_single_split(current)
