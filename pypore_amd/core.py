"""Segment / MetaSegment value types -- the attribute surface of PyPore/core.py:14-249
(current, start, end, duration, mean, std, min, max, n, scale(), to_dict/to_json, to_meta)
in Python 3, so that DataTypes-style callers (`seg.event = ...; seg.scale(second)`,
DataTypes.py:287-289, :595-600) keep working.  The statistics are lazy like the reference's
properties; when the GPU already produced them (segstat kernel) they are served from there.
"""
import json
from contextlib import contextmanager

import numpy as np


@contextmanager
def ignored(*exceptions):
    """core.py:251-261."""
    try:
        yield
    except exceptions:
        pass


def _as_numpy(current):
    if hasattr(current, "detach"):          # torch tensor (possibly on the GPU)
        return current.detach().cpu().numpy()
    return np.asarray(current)


class MetaSegment(object):
    """Metadata of a segment without the current array (core.py:14-113)."""

    def __init__(self, **kwargs):
        for key, value in kwargs.items():
            with ignored(AttributeError):
                setattr(self, key, value)
        if hasattr(self, "current"):
            cur = _as_numpy(self.current)
            self.n = len(cur)
            self.mean = np.mean(cur)
            self.std = np.std(cur)
            self.min = np.min(cur)
            self.max = np.max(cur)
            del self.current
        if hasattr(self, "start") and hasattr(self, "end") and not hasattr(self, "duration"):
            self.duration = self.end - self.start
        elif hasattr(self, "start") and hasattr(self, "duration") and not hasattr(self, "end"):
            self.end = self.start + self.duration
        elif hasattr(self, "end") and hasattr(self, "duration") and not hasattr(self, "start"):
            self.start = self.end - self.duration

    def __repr__(self):
        return self.to_json()

    def __len__(self):
        return self.n

    def delete(self):
        del self

    def to_meta(self):
        pass

    def to_dict(self):
        keys = ['mean', 'std', 'min', 'max', 'start', 'end', 'duration']
        d = {i: _jsonable(getattr(self, i)) for i in keys if hasattr(self, i)}
        d['name'] = self.__class__.__name__
        return d

    def to_json(self, filename=None):
        _json = json.dumps(self.to_dict(), indent=4, separators=(',', ' : '))
        if filename:
            with open(filename, 'w') as outfile:
                outfile.write(_json)
        return _json

    @classmethod
    def from_json(cls, filename=None, json=None):
        assert filename or json and not (filename and json)
        import json as _json
        if filename:
            with open(filename, 'r') as infile:
                json = ''.join(line for line in infile)
        d = _json.loads(json)
        d.pop('name', None)
        return MetaSegment(**d)


def _jsonable(v):
    if isinstance(v, (np.floating,)):
        return float(v)
    if isinstance(v, (np.integer,)):
        return int(v)
    return v


class Segment(object):
    """A stretch of ionic current plus lazily computed statistics (core.py:115-249)."""

    def __init__(self, current, **kwargs):
        self.current = current
        for key, value in kwargs.items():
            if hasattr(self, key):          # cannot override the statistics (core.py:131-132)
                continue
            with ignored(AttributeError):
                setattr(self, key, value)

    def __repr__(self):
        return self.to_json()

    def __len__(self):
        return self.n

    def to_dict(self):
        keys = ['mean', 'std', 'min', 'max', 'start', 'end', 'duration']
        d = {i: _jsonable(getattr(self, i)) for i in keys if hasattr(self, i)}
        d['name'] = self.__class__.__name__
        return d

    def to_json(self, filename=None):
        _json = json.dumps(self.to_dict(), indent=4, separators=(',', ' : '))
        if filename:
            with open(filename, 'w') as outfile:
                outfile.write(_json)
        return _json

    def to_meta(self):
        """core.py:175-186: freeze the statistics, drop the array, become a MetaSegment."""
        for key in ['mean', 'std', 'min', 'max', 'end', 'start', 'duration']:
            with ignored(KeyError, AttributeError):
                self.__dict__[key] = getattr(self, key)
        del self.current
        self.__dict__.pop('_gpu_stats', None)
        self.__class__ = type("MetaSegment", (MetaSegment,), self.__dict__)

    def delete(self):
        with ignored(AttributeError):
            del self.current
        del self

    def scale(self, sampling_freq):
        """Samples -> seconds (core.py:199-207)."""
        with ignored(AttributeError):
            self.start /= sampling_freq
            self.end /= sampling_freq
            self.duration /= sampling_freq

    def _stat(self, i, fn):
        st = self.__dict__.get('_gpu_stats')
        if st is not None:
            return st[i]
        return fn(_as_numpy(self.current))

    @property
    def mean(self):
        return self._stat(0, np.mean)

    @property
    def std(self):
        return self._stat(1, np.std)

    @property
    def min(self):
        return self._stat(2, np.min)

    @property
    def max(self):
        return self._stat(3, np.max)

    @property
    def n(self):
        return len(self.current)

    @classmethod
    def from_json(cls, filename=None, json=None):
        assert filename or json and not (filename and json)
        import json as _json
        if filename:
            with open(filename, 'r') as infile:
                json = ''.join(line for line in infile)
        d = _json.loads(json)
        d.pop('name', None)
        if 'current' not in d:
            return MetaSegment(**d)
        current = np.array(d.pop('current'), dtype=np.float64)
        return Segment(current, **d)
