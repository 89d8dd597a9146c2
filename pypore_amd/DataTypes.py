"""File / Event: the containers on either side of the segmenter.

Call contract kept from the reference (PyPore/DataTypes.py):
  File(filename) | File(current=, timestep=)   .second = 1000 / timestep, .events, .parse(parser)        (:567-602)
  Event(current, start, end, duration, second, file)   .filter(order, cutoff), .parse(parser), .segments   (:239-333)
  to_dict / to_json / from_json / to_meta on both, MetaEvent, Event.from_segments                         (:480-545, :683-796)
  Experiment(filenames).parse(event_detector, segmenter, filter_params), .files / .events / .segments; Sample   (:938-1049)
and the JSON schema of README.md:346-391 (tests/golden/readme_file.json).  Plotting, HMM merging and MySQL are out of
scope.

Built here around three pieces: `encode()` turns any tree of records into JSON-able dicts (one recursive walk
instead of per-class loops), `_rebuild_event()` is the inverse for one event, and a File read from an .abf keeps
the int16 counts next to the float64 current (pypore_amd.grid), so that parsing stays on the 2 B/sample device route.
"""
import numpy as np

from .core import MetaSegment, Segment, SEGMENT_FIELDS, dump_json, fields_of, gc_paused, ignored, load_json, plain, raw_current
from .parsers import SpeedyStatSplit, lambda_event_parser, parser as _parser_base

EVENT_FIELDS = SEGMENT_FIELDS + ('filtered', 'filter_order', 'filter_cutoff', 'n', 'state_parser', 'segments')
FILE_FIELDS = ('filename', 'n', 'event_parser', 'mean', 'std', 'duration', 'start', 'end', 'events')


def encode(value):
    """Records (anything with to_dict) -> dicts, sequences -> lists, numpy scalars -> numbers, recursively."""
    if hasattr(value, 'to_dict'):
        value = value.to_dict()
    if isinstance(value, dict):
        return {k: encode(v) for k, v in value.items()}
    if isinstance(value, (list, tuple)):
        return [encode(v) for v in value]
    return plain(value)


def _parser_from(d):
    return _parser_base.from_json(dump_json(d)) if d is not None else None


class MetaEvent(MetaSegment):
    """An event reduced to its numbers (and its MetaSegments)."""
    json_fields = EVENT_FIELDS

    def delete(self):
        for segment in self.__dict__.get('segments', []):
            segment.delete()
        self.__dict__.clear()


class Event(Segment):
    """A blockade event: its current, and after parse() its segments."""
    json_fields = EVENT_FIELDS

    def __init__(self, current, segments=(), **kwargs):
        segments = list(segments)
        if segments:                                     # an event assembled from segments owns their concatenation
            try:
                current = np.concatenate([seg.current for seg in segments])
            except (AttributeError, ValueError):
                current = []
        Segment.__init__(self, current, filtered=False, segments=segments, **kwargs)

    @property
    def n(self):
        return len(self.__dict__.get('segments') or ())

    # ---- Event.filter (DataTypes.py:258-274) ----------------------------------------------------------------
    def filter(self, order=1, cutoff=2000., quantum=None):
        """Bessel low-pass, cutoff relative to the Nyquist frequency of the file's sampling rate, run forward and
        backward (scipy.signal.filtfilt semantics) on the device; `current` becomes the float64 result.  Orders 1..8
        run on the device (1, the reference's default, by scans; 2..8 by segments with halos); higher orders raise
        ValueError.  The result no longer lies on the ADC grid: see parse.  Like the reference, this filters whatever
        `current` holds: a current on an ADC grid (what a file reader returns) goes up as counts, anything else -- an
        event that was filtered before (Experiment.parse twice on the same files), float64 data on no grid -- as float64."""
        if type(self) is not Event:
            raise TypeError("Cannot filter a metaevent. Must have the current.")
        import torch
        from . import engine
        cur = raw_current(self)
        s = None
        if not self.__dict__.get("filtered") or quantum is not None:
            try:
                s = engine.to_device(cur, quantum)
            except ValueError:
                if quantum is not None:
                    raise                                # the caller named a grid the data does not lie on
        if s is not None:
            ctx = engine.context(s.tensor.device.index)
            out = ctx.filter_bessel(s.tensor, s.quantum, cutoff=cutoff, sampling_freq=float(self.second), order=order)
            # (a DC offset passes a unit-gain low-pass unchanged: filter the counts, put the offset back)
            self.current = out.cpu().numpy() + s.offset if s.offset else out.cpu().numpy()
        else:
            # no grid: the float64 values themselves (a current still parked on the device stays there)
            t = getattr(cur, "tensor", None)
            off = 0.0
            if t is None or not t.is_cuda:
                a = np.ascontiguousarray(np.asarray(self.current), dtype=np.float64)
                if a.ndim != 1:
                    raise ValueError("Buffer has wrong number of dimensions (expected 1, got %d)" % a.ndim)
                t = torch.from_numpy(a).cuda()
            else:
                # a parked tensor holds the current WITHOUT the file's offset (grid.Deferred.from_tensor(y, off), what
                # parse_events leaves behind): a unit-gain low-pass passes a constant unchanged, so it goes back on afterwards
                off = float(getattr(cur, "offset", 0.0) or 0.0)
            ctx = engine.context(t.device.index)
            out = ctx.filter_bessel(t.contiguous(), 1.0, cutoff=cutoff, sampling_freq=float(self.second), order=order).cpu().numpy()
            self.current = out + off if off else out
        self.filtered, self.filter_order, self.filter_cutoff = True, order, cutoff

    # ---- Event.parse (DataTypes.py:276-289, :333) -----------------------------------------------------------
    def parse(self, parser=None, hmm=None):
        """segments = parser.parse(current); every segment learns its event and is rescaled from samples to seconds
        with the file's sampling rate."""
        if parser is None:
            parser = SpeedyStatSplit(prior_segments_per_second=10)
        if hmm is not None:
            raise NotImplementedError("HMM-guided merging needs yahmm (out of scope)")
        if self.__dict__.get("filtered"):
            self.segments = self._parse_filtered(parser)
        else:
            # (our own segmenter takes the counts behind a current that has not been written out; any other parser gets
            #  the float64 array)
            self.segments = parser.parse(raw_current(self) if isinstance(parser, SpeedyStatSplit) else self.current)
        rate = float(self.file.second)
        for segment in self.segments:
            segment.event = self
            segment.scale(rate)
        self.state_parser = parser

    def _on_fine_grid(self):
        """A filtered current is float64 off every ADC grid, and the device segmenter works on exact integer sums.
        The current is centred on its mean (the gains are shift invariant) and rounded to the finest power-of-two grid
        that keeps every count below 2**22 (2**-18 pA for a 100 pA range): on the golden vectors recorded from the
        reference that reproduces every boundary the reference finds on the unrounded float64 current
        (tests/test_filter.py); coarser grids do not -- a heavily smoothed current has almost no variance left.
        Returns (rounded current, grid step, the level that was subtracted)."""
        cur = np.asarray(self.current, dtype=np.float64)
        centre = float(np.mean(cur)) if cur.size else 0.0
        span = float(np.max(np.abs(cur - centre))) if cur.size else 0.0
        step = 2.0 ** (int(np.ceil(np.log2(span * 1.01))) - 22) if span > 0 else 1.0
        centre = np.rint(centre / step) * step
        return np.rint((cur - centre) / step) * step, step, float(centre)

    def _adopt_filtered(self, segments):
        """Segments found on the rounded current keep views of the unrounded one and take their statistics from those."""
        for segment in segments:
            segment.current = self.current[int(segment.start):int(segment.end)]
            segment.__dict__.pop('_gpu_stats', None)
        return segments

    def _parse_filtered(self, parser):
        mode = parser.off_grid if isinstance(parser, SpeedyStatSplit) else None
        if mode == "exact":
            # the reference's own arithmetic on the float64 current itself (ps_segment_exact_f64): nothing is rounded
            return self._adopt_filtered(parser.parse_exact(np.asarray(self.current, dtype=np.float64)))
        rounded, _, centre = self._on_fine_grid()
        if mode == "exact_on_near_tie":
            import warnings
            from . import engine
            with warnings.catch_warnings(record=True) as seen:
                warnings.simplefilter("always", engine.NearTieWarning)
                segs = parser.parse_batch([rounded], [centre])[0]
            # (near_ties() < 0: the call ran where near ties are not counted -- option scan_bs 0, min_width < 8)
            if any(issubclass(w.category, engine.NearTieWarning) for w in seen) or engine.context(parser._grid["device"]).near_ties() < 0:
                segs = parser.parse_exact(np.asarray(self.current, dtype=np.float64))
            return self._adopt_filtered(segs)
        if isinstance(parser, SpeedyStatSplit):
            # (the level goes along: the device judges near ties against the noise of the reference's cumsums, which run on
            #  the uncentred current -- include/poreseg.h, ps_sample_format)
            return self._adopt_filtered(parser.parse_batch([rounded], [centre])[0])
        return self._adopt_filtered(parser.parse(rounded))

    # ---- persistence ----------------------------------------------------------------------------------------
    def to_dict(self):
        return fields_of(self, self.json_fields)

    def to_json(self, filename=None):
        return dump_json(encode(self), filename)

    def to_meta(self):
        for segment in self.segments:
            segment.to_meta()
        self.freeze(SEGMENT_FIELDS + ('n',))
        self.__class__ = MetaEvent

    def delete(self):
        for segment in self.__dict__.get('segments', []):
            segment.delete()
        self.__dict__.clear()

    @classmethod
    def from_json(cls, _json):
        """Event from JSON text or a *.json path.  Without a `current` entry (what to_json writes) the result is a
        MetaEvent holding the stored numbers, its segments as MetaSegments and its parser rebuilt."""
        d = dict(load_json(_json))
        d.pop('name', None)
        if 'current' in d:
            return cls(np.asarray(d['current'], dtype=np.float64), start=d.get('start', 0))
        d['segments'] = [MetaSegment(**{k: v for k, v in sj.items() if k != 'name'}) for sj in d.get('segments', [])]
        if isinstance(d.get('state_parser'), dict):
            d['state_parser'] = _parser_from(d['state_parser'])
        return MetaEvent(**d)

    @classmethod
    def from_segments(cls, segments):
        """Event spanning `segments`.  With their currents at hand: the concatenation.  From metadata alone: a
        MetaEvent whose duration is the sum, whose mean is the duration-weighted mean and whose std pools the
        segment variances about their own means (what the reference's formulas at :538-545 are after; its own result
        there is an Event with an empty current)."""
        segments = list(segments)
        if all(hasattr(seg, 'current') for seg in segments):
            return cls(current=None, start=0, segments=segments)
        dur = float(sum(seg.duration for seg in segments))
        mean = sum(seg.mean * seg.duration for seg in segments) / dur
        std = float(np.sqrt(sum(seg.std ** 2 * seg.duration for seg in segments) / dur))
        return MetaEvent(start=0, duration=dur, mean=mean, std=std, n=len(segments), segments=segments, filtered=False)


class File(Segment):
    """One recording: the current of an .abf file (or an array with its time step in ms) and the events found in it."""
    json_fields = FILE_FIELDS

    def __init__(self, filename=None, current=None, timestep=None, **kwargs):
        if current is not None and timestep is not None:
            filename = ""
        elif filename and current is None and timestep is None:
            from .abf import read_abf_counts
            from .grid import Deferred
            # The reference's reader returns float64 pA (read_abf.py:208-212).  Here the array is written out when
            # somebody reads `file.current` (a GridArray then, as from abf.read_abf); detection and segmentation take
            # the file's int16 counts and never ask for it: 0.10 s of an 0.21 s Experiment.parse on a 1e8-sample file.
            timestep, counts, scale, offset = read_abf_counts(filename)
            current = Deferred.from_counts(counts, scale, offset)
            filename = filename.split("\\")[-1]
            filename = filename[:-4] if filename.endswith(".abf") else filename
        else:
            raise SyntaxError("Must provide current and timestep, or filename corresponding to a valid abf file.")
        Segment.__init__(self, current=current, filename=filename, second=1000. / timestep, events=[], sample=None)

    def __getitem__(self, index):
        return self.events[index]

    @property
    def n(self):
        return len(self.events)

    def parse(self, parser=None):
        """Event detection: one Event per Segment the parser returns, its start / end / duration in seconds."""
        if parser is None:
            parser = lambda_event_parser(threshold=90)
        rate = self.second
        device_route = isinstance(parser, lambda_event_parser) and parser._builtin
        self.events = [Event(current=raw_current(seg), start=seg.start / rate, end=(seg.start + seg.duration) / rate,
                             duration=seg.duration / rate, second=rate, file=self)
                       for seg in parser.parse(raw_current(self) if device_route else self.current)]
        self.event_parser = parser

    def parse_events(self, parser=None, filter_params=None):
        """The inner loop of Experiment.parse (DataTypes.py:975-984) for this file: every event is filtered when
        `filter_params` = (order, cutoff) is given, then all events are segmented in as few device calls as their
        representations allow -- one for unfiltered events, one per grid step for filtered ones (events of one file span
        similar ranges and mostly share a step).  Same result as `event.filter(...); event.parse(parser)` per event.
        With a SpeedyStatSplit the filtered currents never leave the device between the two steps
        (parse_filtered_batch: filter, re-quantisation and segmentation on the device, one copy of the float64 result
        back for Event.current)."""
        with gc_paused():                                # (one pause for the file: re-enabling after every event costs a
            self._parse_events(parser, filter_params)    #  collection each time, 0.5 ms x the number of events)

    def _parse_events(self, parser, filter_params):
        if parser is None:
            parser = SpeedyStatSplit(prior_segments_per_second=10)
        rate = float(self.second)
        if filter_params is not None and hasattr(parser, "parse_filtered_batch") and all(type(ev) is Event for ev in self.events):
            # filter -> grid -> segments without leaving the device: only the filtered float64 currents come back
            order, cutoff = (tuple(filter_params) + (2000.,))[:2] if len(tuple(filter_params)) else (1, 2000.)
            done = parser.parse_filtered_batch([raw_current(ev) for ev in self.events], order, cutoff, rate)
            for ev, (cur, segs) in zip(self.events, done):
                ev.current = cur
                ev.filtered, ev.filter_order, ev.filter_cutoff = True, order, cutoff
                ev.segments = segs
                for segment in segs:
                    segment.event = ev
                    segment.scale(rate)
                ev.state_parser = parser
            return
        if filter_params is not None:
            for ev in self.events:
                ev.filter(*filter_params)
        batched = hasattr(parser, "parse_batch")
        results = [None] * len(self.events)
        plain_idx = [i for i, ev in enumerate(self.events) if not ev.__dict__.get("filtered")]
        currents = [raw_current(self.events[i]) if batched else self.events[i].current for i in plain_idx]
        for i, segs in zip(plain_idx, parser.parse_batch(currents) if batched else [parser.parse(c) for c in currents]):
            results[i] = segs
        by_step = {}
        exact_modes = isinstance(parser, SpeedyStatSplit) and parser.off_grid in ("exact", "exact_on_near_tie")
        for i, ev in enumerate(self.events):
            if ev.__dict__.get("filtered"):
                if exact_modes:                          # (the event decides for itself: Event._parse_filtered)
                    results[i] = ev._parse_filtered(parser)
                    continue
                rounded, step, centre = ev._on_fine_grid()
                by_step.setdefault(step, []).append((i, rounded, centre))
        for group in by_step.values():
            currents = [r for _, r, _ in group]
            if batched and isinstance(parser, SpeedyStatSplit):
                found = parser.parse_batch(currents, [c for _, _, c in group])       # (levels: this package's extension)
            elif batched:
                found = parser.parse_batch(currents)     # a user's parser with the one-argument parse_batch of earlier rounds
            else:
                found = [parser.parse(c) for c in currents]
            for (i, _, _), segs in zip(group, found):
                results[i] = self.events[i]._adopt_filtered(segs)
        for ev, segs in zip(self.events, results):
            ev.segments = segs
            for segment in segs:
                segment.event = ev
                segment.scale(rate)
            ev.state_parser = parser

    # ---- persistence ----------------------------------------------------------------------------------------
    def to_dict(self):
        if 'end' not in self.__dict__ and 'start' in self.__dict__ and 'duration' in self.__dict__:
            self.end = self.start + self.duration
        return fields_of(self, self.json_fields)

    def to_json(self, filename=None):
        """The file, its event parser, and every event with its segments and state parser (README.md:346-391)."""
        return dump_json(encode(self), filename)

    def to_meta(self):
        self.__dict__.pop('current', None)
        for event in self.events:
            event.to_meta()

    def delete(self):
        for event in self.__dict__.get('events', []):
            event.delete()
        self.__dict__.clear()

    @classmethod
    def from_json(cls, _json):
        """Rebuilds a stored analysis (JSON text or *.json path).  If `<filename>.abf` can be read, events and segments
        get views of its current again and filtered events are filtered again; otherwise everything comes back as
        MetaEvent / MetaSegment."""
        d = load_json(_json)
        if d.get('name') != "File":
            raise TypeError("JSON does not encode a file")
        try:
            file = cls(filename=d['filename'] + ".abf")
            meta = False
        except Exception:
            file = cls(current=[], timestep=1)
            file.filename = d.get('filename', "")
            del file.current                            # nothing to take statistics of: the stored numbers are all there is
            meta = True
        if 'event_parser' in d:
            file.event_parser = _parser_from(d['event_parser'])
        file.events = [_rebuild_event(file, ej, meta) for ej in d.get('events', [])]
        return file


def _rebuild_event(file, ej, meta):
    """One event of a stored analysis: MetaEvent from the numbers, or Event on views of the file's current."""
    segs = ej.get('segments', [])
    filtered = bool(ej.get('filtered', False))
    state_parser = _parser_from(ej.get('state_parser'))
    if meta:
        event = MetaEvent(**{k: v for k, v in ej.items() if k not in ('name', 'segments', 'state_parser')})
        event.segments = [MetaSegment(**{k: v for k, v in sj.items() if k != 'name'}) for sj in segs]
    else:
        rate = file.second
        a, b = int(ej['start'] * rate), int(ej['end'] * rate)
        event = Event(current=file.current[a:b], start=a / rate, end=b / rate, duration=(b - a) / rate,
                      second=rate, file=file)
        if filtered:
            event.filter(order=ej['filter_order'], cutoff=ej['filter_cutoff'])
        event.segments = [Segment(current=event.current[int(sj['start'] * rate):int(sj['end'] * rate)], second=rate,
                                  event=event, **{k: v for k, v in sj.items() if k != 'name'}) for sj in segs]
    event.filtered = filtered
    if state_parser is not None:
        event.state_parser = state_parser
    return event


class Experiment(object):
    """Several recordings analysed together (DataTypes.py:938-1037): `parse` opens one file after the other, detects its
    events, filters and segments them, and keeps the files; `.events` / `.segments` run over all of them.

    `filenames` may also hold File objects (e.g. built from arrays), which are taken as they are.  Per file the events
    go to the device together (File.parse_events) instead of one by one."""

    def __init__(self, filenames, name=None):
        self.filenames = filenames
        self.name = name or "Experiment"
        self.files = []

    _DEFAULT = object()

    def parse(self, event_detector=None, segmenter=_DEFAULT, filter_params=(1, 2000), verbose=True, meta=False, workers=None):
        """Defaults as in the reference (:956-960): lambda_event_parser(threshold=90), SpeedyStatSplit with
        prior_segments_per_second=10 and cutoff_freq=2000, a first-order 2 kHz Bessel filter.  segmenter=None: events
        are detected (and filtered) only; filter_params=None: no filter; meta=True: the currents are dropped afterwards.

        The reference takes one file after the other (:968-984).  Files are independent, so here up to `workers` of them
        are in flight (default: 2 when detector and segmenter are this package's own device classes, else 1): while file
        k is segmented, file k+1 is read, uploaded and searched for events on another host thread -- every thread has
        its own device context (engine.context), its kernels overlap with the others' like the contexts of a StreamPool.
        `files`, and the lines printed with verbose=True, keep the order of `filenames` whatever finishes first."""
        if event_detector is None:
            event_detector = lambda_event_parser(threshold=90)
        if segmenter is Experiment._DEFAULT:
            segmenter = SpeedyStatSplit(prior_segments_per_second=10, cutoff_freq=2000.)
        entries = list(self.filenames)
        if workers is None:
            ours = isinstance(event_detector, lambda_event_parser) and getattr(event_detector, "_builtin", False) and \
                (segmenter is None or isinstance(segmenter, SpeedyStatSplit))
            workers = 2 if ours else 1
        workers = max(1, min(int(workers), len(entries)))

        def one(entry, say=None):
            lines = []
            say = say or lines.append                    # (one file at a time: the lines appear as the reference prints them)
            file = entry if isinstance(entry, File) else File(entry)
            say("Opening {}".format(file.filename))
            file.parse(parser=event_detector)
            say("\tDetected {} Events".format(file.n))
            if segmenter is not None:
                file.parse_events(segmenter, filter_params)
                for i, event in enumerate(file.events):
                    say("\t\tEvent {} has {} segments".format(i + 1, event.n))
            elif filter_params is not None:
                for event in file.events:
                    event.filter(*filter_params)
            if meta:
                file.to_meta()
            return file, lines

        def take(done):
            file, lines = done
            if verbose:
                for line in lines:
                    print(line)
            self.files.append(file)

        if workers == 1:
            for entry in entries:
                take(one(entry, print if verbose else (lambda line: None)))
            return
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=workers, thread_name_prefix="pypore-file") as pool:
            for fut in [pool.submit(one, entry) for entry in entries]:
                take(fut.result())

    def apply_hmm(self, hmm, filter=None, indices=None):
        raise NotImplementedError("HMM decoding needs yahmm (out of scope)")

    def delete(self):
        for file in self.files:
            file.delete()
        self.__dict__.clear()

    @property
    def n(self):
        return len(self.files)

    @property
    def events(self):
        """All events of all files, in file order."""
        return [event for file in self.files for event in file.events]

    @property
    def segments(self):
        """All segments of all events."""
        return [segment for event in self.events for segment in event.segments]


class Sample(object):
    """Events (and the files they came from) attributed to one substrate (DataTypes.py:1039-1049)."""

    def __init__(self, events=None, files=None, label=None):
        self.events = list(events) if events is not None else []
        self.files = list(files) if files is not None else []
        self.label = label

    def delete(self):
        with ignored(AttributeError):
            for file in self.files:
                file.delete()
        for event in self.events:
            event.delete()
        del self.events
        del self.files
