"""Segment / MetaSegment: the result types consumers of the parser plug-in API read.

What callers rely on (PyPore/core.py:14-249 is the reference surface): a Segment carries
`current` (a view of the caller's array) and `start`, `end`, `duration`; `mean`, `std` (population), `min`,
`max`, `n` are derived on first use; `scale(fs)` turns sample units into seconds (DataTypes.py:287-289);
`to_dict` / `to_json` emit the keys mean, std, min, max, start, end, duration, name (the JSON schema of
README.md:346-391); `to_meta()` drops the array and keeps the numbers (a MetaSegment).

Design here: the derived statistics are ONE non-data descriptor (`_Derived`) per name, so a frozen value in
the instance dict (to_meta) or a row of device-computed statistics (the K2 kernel's output, `_gpu_stats`)
shadows the numpy evaluation without any per-property code; serialisation is driven by a field tuple.
"""
import json
from contextlib import contextmanager

import numpy as np

from .grid import Deferred

JSON_STYLE = dict(indent=4, separators=(',', ' : '))      # the layout of README.md:346-391
STAT_COLUMNS = ('mean', 'std', 'min', 'max')              # column order of ps_segstat (include/poreseg.h)
SEGMENT_FIELDS = STAT_COLUMNS + ('start', 'end', 'duration')


@contextmanager
def ignored(*exceptions):
    """Suppress the given exception classes inside the block (core.py:251-261 offers the same helper)."""
    try:
        yield
    except exceptions:
        pass


def host_array(current):
    """numpy view of `current`, fetching it from the GPU if it is a torch tensor."""
    return current.detach().cpu().numpy() if hasattr(current, "detach") else np.asarray(current)


def plain(value):
    """numpy scalars -> Python numbers (json refuses numpy types under Python 3)."""
    if isinstance(value, np.generic):
        return value.item()
    return value


def dump_json(record, filename=None):
    text = json.dumps({k: plain(v) for k, v in record.items()}, **JSON_STYLE)
    if filename:
        with open(filename, 'w') as out:
            out.write(text)
    return text


def load_json(source):
    """dict from JSON text, or from the file when `source` names a *.json path."""
    if isinstance(source, str) and source.endswith(".json"):
        with open(source, 'r') as infile:
            source = infile.read()
    return json.loads(source)


def fields_of(obj, names):
    """{name: value} for the attributes that exist (lazy ones are evaluated), plus the class name."""
    out = {}
    for name in names:
        try:
            out[name] = plain(getattr(obj, name))
        except (AttributeError, ValueError, KeyError):
            continue
    out['name'] = type(obj).__name__
    return out


class _Derived(object):
    """Statistic of `obj.current`, unless the instance dict holds a value or a device row already."""

    def __init__(self, column, fn):
        self.column, self.fn = column, fn

    def __get__(self, obj, owner=None):
        if obj is None:
            return self
        row = obj.__dict__.get('_gpu_stats')
        if row is not None:
            return row[self.column]
        return self.fn(host_array(obj.current))


class _Current(object):
    """The `current` attribute of a Segment: what was stored, except that a current that has not been written out yet
    (grid.Deferred: a file's float64 array, a filtered current still on the GPU) is built on first access and kept.
    Code that only needs the length or the source of the samples reads `raw_current(obj)` instead."""

    def __get__(self, obj, owner=None):
        if obj is None:
            return self
        try:
            value = obj.__dict__['current']
        except KeyError:
            raise AttributeError('current')
        if isinstance(value, Deferred):
            value = obj.__dict__['current'] = value.value()
        return value

    def __set__(self, obj, value):
        obj.__dict__['current'] = value

    def __delete__(self, obj):
        try:
            del obj.__dict__['current']
        except KeyError:
            raise AttributeError('current')


def raw_current(obj):
    """What `obj.current` holds without building it (a grid.Deferred stays one)."""
    return obj.__dict__.get('current')


class _Record(object):
    """Shared serialisation of the value types."""
    json_fields = SEGMENT_FIELDS

    def to_dict(self):
        return fields_of(self, self.json_fields)

    def to_json(self, filename=None):
        return dump_json(self.to_dict(), filename)

    def __repr__(self):
        return self.to_json()

    def __len__(self):
        return self.n

    @classmethod
    def from_json(cls, filename=None, json=None):
        if bool(filename) == bool(json):
            raise AssertionError("give a filename or a JSON string")
        if filename:
            with open(filename, 'r') as infile:
                json = infile.read()
        d = load_json(json)
        d.pop('name', None)
        if 'current' in d:
            return Segment(np.array(d.pop('current'), dtype=np.float64), **d)
        return MetaSegment(**d)


_gc_lock = __import__("threading").Lock()
_gc_state = [0, True]                                  # [nesting count over all threads, was the collector enabled]


@contextmanager
def gc_paused():
    """No cyclic collections while a result list is built: tens of thousands of small objects trigger several full
    collections, each of which walks everything torch and numpy created at import (3 of 6 us per Segment)."""
    import gc
    # (counted: several host threads build result lists at once -- Experiment.parse with workers -- and the collector
    #  comes back on when the last of them is done)
    with _gc_lock:
        if _gc_state[0] == 0:
            _gc_state[1] = gc.isenabled()
            gc.disable()
        _gc_state[0] += 1
    try:
        yield
    finally:
        with _gc_lock:
            _gc_state[0] -= 1
            if _gc_state[0] == 0 and _gc_state[1]:
                gc.enable()


def segments_from_edges(current, edges, stats=None):
    """[Segment(current[a:z], start=a, duration=z - a, end=z)] for consecutive edges, `_gpu_stats` rows attached when
    given -- what the constructor does, without its keyword loop (a file's events hold 10^4 .. 10^5 segments)."""
    from .grid import stretches
    out = []
    new = Segment.__new__
    with gc_paused():
        edges = [int(e) for e in edges]
        parts = stretches(current, edges)
        for k in range(len(edges) - 1):
            a, z = edges[k], edges[k + 1]
            seg = new(Segment)
            d = seg.__dict__
            d['current'] = parts[k]
            d['start'], d['duration'], d['end'] = a, z - a, z
            if stats is not None:
                d['_gpu_stats'] = stats[k]
            out.append(seg)
    return out


class MetaSegment(_Record):
    """The numbers of a segment without its current.  Any two of start / end / duration give the third; a
    `current` keyword is reduced to n, mean, std, min, max on the spot."""

    def __init__(self, **kwargs):
        current = kwargs.pop('current', None)
        for key, value in kwargs.items():
            with ignored(AttributeError):
                setattr(self, key, value)
        if current is not None:
            cur = host_array(current)
            self.n = len(cur)
            for name, fn in (('mean', np.mean), ('std', np.std), ('min', np.min), ('max', np.max)):
                setattr(self, name, fn(cur))
        have = [hasattr(self, k) for k in ('start', 'end', 'duration')]
        if have == [True, True, False]:
            self.duration = self.end - self.start
        elif have == [True, False, True]:
            self.end = self.start + self.duration
        elif have == [False, True, True]:
            self.start = self.end - self.duration

    def to_meta(self):
        pass

    def delete(self):
        self.__dict__.clear()


class Segment(_Record):
    """A stretch of ionic current.  Keywords become attributes, except the names of the derived statistics."""
    mean = _Derived(0, np.mean)
    std = _Derived(1, np.std)
    min = _Derived(2, np.min)
    max = _Derived(3, np.max)
    derived = STAT_COLUMNS + ('n',)
    current = _Current()

    def __init__(self, current, **kwargs):
        self.current = current
        for key, value in kwargs.items():
            if key in Segment.derived:
                continue
            try:
                setattr(self, key, value)
            except AttributeError:                       # (a read-only property of a subclass: the reference skips those)
                pass

    @property
    def n(self):
        return len(raw_current(self))

    def scale(self, sampling_freq):
        """Sample units -> seconds."""
        for name in ('start', 'end', 'duration'):
            if name in self.__dict__:
                self.__dict__[name] = self.__dict__[name] / sampling_freq

    def freeze(self, names=SEGMENT_FIELDS):
        """Evaluates the lazy attributes into the instance dict and lets go of the array."""
        for name in names:
            with ignored(AttributeError, KeyError, ValueError):
                self.__dict__[name] = getattr(self, name)
        self.__dict__.pop('current', None)
        self.__dict__.pop('_gpu_stats', None)

    def to_meta(self):
        self.freeze()
        self.__class__ = MetaSegment

    def delete(self):
        self.__dict__.clear()
