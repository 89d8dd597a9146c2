"""FastStatSplit -- the class surface of PyPore/cparsers.pyx:45-275 backed by the HIP kernels.

Same constructor signature (cparsers.pyx:55-57), same public attribute `min_gain`, same
methods parse / best_single_split / score_samples, same assertion and ValueError behaviour.
Every compute call goes through the C ABI (include/poreseg.h); nothing here computes on the CPU.
"""
import numpy as np

from . import _lib, engine
from .core import Segment, segments_from_edges

# What happens to float input that lies on NO ADC grid (the reference takes any float64 buffer, cparsers.pyx:53):
#   "raise"              ValueError: nothing is rounded silently (default);
#   "requantise"         centred and rounded on the device to the finest power-of-two grid that keeps the counts below 2**22, segmented
#                        from exact integer sums (fast; equal to the reference except in windows decided within the rounding noise
#                        of the reference's own cumsums -- engine.NearTieWarning says when: DESIGN.md 2);
#   "exact"              the reference's own arithmetic: its sequential fp64 cumsums and var_c expressions on the device
#                        (ps_segment_exact_f64) -- the reference's boundaries on the same input by construction, tens of ms per 1e6 samples;
#   "exact_on_near_tie"  the fast route, and the exact one only for input on which the fast route counted a near tie.
OFF_GRID_MODES = ("raise", "requantise", "exact", "exact_on_near_tie")
FILTER_BATCH = True          # False: a file's events are filtered and re-quantised one library call each (A/B: tools/profile_filtered_file.py)


def _dc_counts(level, step):
    """A level in pA as whole counts of `step`, clipped to int32 (ps_sample_format.offset_counts of fp32 samples)."""
    return int(max(-2147483647, min(2147483647, round(float(level) / float(step)))))


class FastStatSplit(object):
    def __init__(self, min_width=100, max_width=1000000, window_width=10000,
                 min_gain_per_sample=None, false_positive_rate=None,
                 prior_segments_per_second=None, sampling_freq=1.e5, cutoff_freq=None,
                 quantum=None, device=None, offset=None, off_grid="raise"):
        if off_grid not in OFF_GRID_MODES:
            raise ValueError("off_grid must be one of %s" % ", ".join(repr(m) for m in OFF_GRID_MODES))
        self.off_grid = off_grid
        self.min_width = int(min_width)
        self.max_width = int(max_width)
        self.window_width = int(window_width)
        self.sampling_freq = int(sampling_freq)              # cdef int (cparsers.pyx:51,61)
        self._params = _lib.split_params(min_width, max_width, window_width, min_gain_per_sample,
                                         false_positive_rate, prior_segments_per_second, sampling_freq,
                                         cutoff_freq)
        # validates with the reference's assertions and computes min_gain (cparsers.pyx:69-101)
        self.min_gain = _lib.min_gain(min_width=min_width, max_width=max_width, window_width=window_width,
                                      min_gain_per_sample=min_gain_per_sample,
                                      false_positive_rate=false_positive_rate,
                                      prior_segments_per_second=prior_segments_per_second,
                                      sampling_freq=sampling_freq, cutoff_freq=cutoff_freq)
        self.quantum = quantum                   # pA per ADC count and pA at count 0 of float input (default: found, see
        self.offset = offset                     # engine.to_device); ignored for GridArrays, which know their own
        self.device = device

    # ---- cparsers.pyx:103-118 -------------------------------------------------------------------
    def parse(self, current):
        """Segments of `current` (list of core.Segment whose `.current` are views, start/end/
        duration in samples), exactly like the reference."""
        return self.parse_batch([current])[0]

    def parse_batch(self, currents, levels=None):
        """One device call for many independent events (one reference parse() per event).  Events that reach the
        device in different representations (float32 pA / int16 counts, or int16 on different scales) go in one call
        per representation.  levels: per event, the level in pA that was subtracted from it upstream (a filtered event
        that Event.parse centred and rounded): passed on as offset_counts, see include/poreseg.h."""
        ctx = engine.context(self.device)
        import torch
        out = [None] * len(currents)

        def upload(idx, full_detect):
            parts = [engine.to_device(currents[i], self.quantum, self.offset, self.device, full_detect) for i in idx]
            q = self.quantum
            if q is None:                           # float32 events on power-of-two grids: the finest one serves all
                q = min(p.quantum for p in parts)
            return parts, q

        def requantised(i):
            """off_grid="requantise": float input that lies on no ADC grid (the reference takes any float64 buffer,
            cparsers.pyx:53,103-111 -- filtered or resampled on the host, np.random.normal test data) is centred and
            rounded on the device to the finest power-of-two grid that keeps its counts below 2**22 (ps_requantise, what
            Event.parse does for a filtered event) and segmented on the 64-bit digest.  The segments hold views of the
            ORIGINAL values and take their statistics from those."""
            cur = np.ascontiguousarray(currents[i], dtype=np.float64)
            if cur.ndim != 1:
                raise ValueError("Buffer has wrong number of dimensions (expected 1, got %d)" % cur.ndim)
            n = cur.size
            if n == 0:
                out[i] = [Segment(current=currents[i][0:0], start=0, duration=0, end=0)]
                return
            dev = torch.device("cuda", torch.cuda.current_device() if self.device is None else int(self.device))
            z, centre, step = ctx.requantise(torch.from_numpy(cur).to(dev))
            # (the level that was subtracted, as offset_counts: the device judges near ties against the noise of the reference's
            #  cumsums, which run on the uncentred values -- include/poreseg.h)
            bounds, boff, _ = ctx.segment_batch(z, np.array([0, n], dtype=np.int64), self._params, step, want_stats=False,
                                                offset_counts=_dc_counts(centre, step))
            edges = np.concatenate(([0], bounds.cpu().numpy(), [n])).tolist()
            src = currents[i]
            out[i] = [Segment(current=src[a:z_], start=a, duration=z_ - a, end=z_) for a, z_ in zip(edges, edges[1:])]

        def exact(i):
            """off_grid="exact": the reference's own prefix sums and expressions on the device (ps_segment_exact_f64)."""
            segs = self.parse_exact_batch([currents[i]])[0]
            out[i] = segs

        def off_grid_route(i):
            if self.off_grid == "exact":
                exact(i)
            elif self.off_grid == "exact_on_near_tie":
                import warnings
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore", engine.NearTieWarning)      # (acted upon right here)
                    requantised(i)
                if ctx.near_ties():
                    exact(i)
            else:
                requantised(i)

        def run(idx, full_detect=False):
            try:
                parts, q = upload(idx, full_detect)
            except ValueError:
                if self.off_grid == "raise" or self.quantum is not None:
                    raise
                if len(idx) > 1:                    # find the event(s) without a grid
                    for i in idx:
                        run([i], full_detect)
                else:
                    off_grid_route(idx[0])
                return
            kinds = {(p.tensor.dtype, p.quantum if p.tensor.dtype == torch.int16 else None) for p in parts}
            if len(kinds) > 1:                      # (float input that resolved to different grids: one call each)
                for i in idx:
                    run([i], full_detect)
                return
            if parts[0].tensor.dtype == torch.int16:
                q = parts[0].quantum
            lens = np.array([p.tensor.numel() for p in parts], dtype=np.int64)
            ev_off = np.concatenate(([0], np.cumsum(lens)))
            try:
                samples = parts[0].tensor if len(parts) == 1 else torch.cat([p.tensor for p in parts])
                dc = 0
                if levels is not None and samples.dtype == torch.float32:
                    dc = _dc_counts(max((levels[i] or 0.0 for i in idx), key=abs), q)
                bounds, boff, stats = ctx.segment_batch(samples, ev_off, self._params, q, offset_counts=dc)
            except ValueError:
                if self.quantum is not None or full_detect:
                    if self.off_grid != "raise" and self.quantum is None and len(idx) == 1:
                        off_grid_route(idx[0])
                        return
                    raise
                # the grid was detected on a subset of the samples and the device found a sample off it: search all
                # samples for the grid once (still ValueError if there is none) -- through run(), so that events which
                # then resolve to different representations are split again
                run(idx, True)
                return
            b = bounds.cpu().numpy()
            st = stats.cpu().numpy()
            for e, i in enumerate(idx):
                cur, n = currents[i], int(lens[e])
                rows = st[boff[e] + e: boff[e + 1] + e + 1]
                if parts[e].offset:                 # device statistics are those of count * quantum
                    rows = rows + np.array([parts[e].offset, 0.0, parts[e].offset, parts[e].offset])
                edges = np.concatenate(([0], b[boff[e]:boff[e + 1]], [n])).tolist()
                out[i] = segments_from_edges(cur, edges, rows)

        # group the events by the representation they will have on the device
        from .grid import grid_of
        groups = {}
        for i, cur in enumerate(currents):
            g = grid_of(cur) if self.quantum is None else None
            groups.setdefault(("counts", g[1]) if g is not None else ("values",), []).append(i)
        for idx in groups.values():
            run(idx)
        return out

    def parse_exact_batch(self, currents):
        """The exact route (ps_segment_exact_f64, include/poreseg.h) for a list of float64 currents (numpy arrays, or float64
        CUDA tensors): the reference's boundaries on the same values by construction.  Returns one list of Segments per
        current (views of the caller's arrays; of a tensor: stretches of its host copy), start / end in samples."""
        import torch
        ctx = engine.context(self.device)
        dev = torch.device("cuda", ctx.device)
        tensors, hosts = [], []
        for cur in currents:
            if isinstance(cur, torch.Tensor):
                t = cur.to(dev, torch.float64).contiguous()
                hosts.append(None)
            else:
                a = np.ascontiguousarray(np.asarray(cur), dtype=np.float64)
                if a.ndim != 1:
                    raise ValueError("Buffer has wrong number of dimensions (expected 1, got %d)" % a.ndim)
                t = torch.from_numpy(a).to(dev)
                hosts.append(cur)
            tensors.append(t)
        lens = np.array([t.numel() for t in tensors], dtype=np.int64)
        starts = np.concatenate(([0], np.cumsum(lens)))[:-1]
        out = []
        if len(tensors) == 0:
            return out
        allt = tensors[0] if len(tensors) == 1 else torch.cat(tensors)
        if allt.numel() == 0:
            return [[Segment(current=(h if h is not None else np.zeros(0))[0:0], start=0, duration=0, end=0)] for h in hosts]
        bounds, boff = ctx.segment_exact_f64(allt, starts, lens, self._params)
        b = bounds.cpu().numpy()
        for e, (t, h) in enumerate(zip(tensors, hosts)):
            n = int(lens[e])
            src = h if h is not None else t.cpu().numpy()
            edges = np.concatenate(([0], b[boff[e]:boff[e + 1]], [n])).tolist()
            out.append([Segment(current=src[a:z_], start=a, duration=z_ - a, end=z_) for a, z_ in zip(edges, edges[1:])])
        return out

    def parse_filtered_batch(self, currents, order=1, cutoff=2000., sampling_freq=1.e5):
        """Event.filter + Event.parse for many events without leaving the device in between (the inner loop of
        Experiment.parse, DataTypes.py:975-984): every current is filtered (ps_filter_bessel), re-quantised on the
        device (ps_requantise) and the events that share a grid step are segmented in one ps_segment_batch.  Returns
        [(filtered float64 current, [Segment...])] in input order, start / end in samples.  The filtered current --
        Event.current afterwards -- stays on the device until it is read (grid.Deferred: the copy of 8 B per sample
        was 0.08 s of an 0.21 s Experiment.parse); the segments hold stretches of it."""
        import torch
        ctx = engine.context(self.device)
        filtered, onto_grid, by_step = [None] * len(currents), [None] * len(currents), {}
        levels = [0.0] * len(currents)
        # Events that are stretches of ONE device tensor on one grid -- what File.parse leaves: the file's int16 counts went up
        # once -- are filtered and re-quantised in one library call per file (ps_filter_requantise_batch: two host
        # synchronisations for the batch instead of three per event; round 5: 314 calls of a 1e8-sample file's 157 events
        # were a third of Experiment.parse's wall clock).  Everything else takes the per-event route below.
        from .grid import grid_of
        batched = set()
        groups = {}
        for i, cur in enumerate(currents):
            t64 = getattr(cur, "tensor", None)
            if t64 is not None and getattr(t64, "is_cuda", False):
                continue                                     # (filtered before, parked on the device: per event)
            g = grid_of(cur) if self.quantum is None else None
            stretch = getattr(cur, "device_stretch", None)
            if g is None or stretch is None:
                continue
            dev = torch.device("cuda", torch.cuda.current_device() if self.device is None else int(self.device))
            where = stretch(dev, engine._int_counts_tensor)
            if where is None or where[2] <= 3 * (int(order) + 1):
                continue
            base, a0, n0 = where
            groups.setdefault((base.data_ptr(), float(g[1])), []).append((i, base, a0, n0, float(g[2])))
        for key, members in groups.items():
            if len(members) < 2 or not FILTER_BATCH:
                continue
            base = members[0][1]
            starts = [a0 for _, _, a0, _, _ in members]
            lens = [n0 for _, _, _, n0, _ in members]
            y_all, z_all, off, centres, steps = ctx.filter_requantise_batch(base, starts, lens, key[1], cutoff=cutoff,
                                                                            sampling_freq=sampling_freq, order=order)
            for k, (i, _, _, _, o) in enumerate(members):
                a, b = int(off[k]), int(off[k + 1])
                filtered[i], onto_grid[i] = (y_all[a:b], o), z_all[a:b]
                levels[i] = float(centres[k]) + o
                by_step.setdefault(float(steps[k]), []).append(i)
                batched.add(i)
        for i, cur in enumerate(currents):
            if i in batched:
                continue
            t64 = getattr(cur, "tensor", None)               # a current filtered before and still parked on the device
            if t64 is not None and t64.is_cuda and t64.dtype == torch.float64:
                y, off = ctx.filter_bessel(t64.contiguous(), 1.0, cutoff=cutoff, sampling_freq=sampling_freq, order=order), float(cur.offset)
            else:
                try:
                    s = engine.to_device(cur, self.quantum, self.offset, self.device)
                    y, off = ctx.filter_bessel(s.tensor, s.quantum, cutoff=cutoff, sampling_freq=sampling_freq, order=order), s.offset
                except ValueError:
                    if self.quantum is not None:
                        raise
                    # on no grid (filtered before, resampled on the host): the float64 values themselves, like the reference
                    a = np.ascontiguousarray(np.asarray(cur), dtype=np.float64)
                    if a.ndim != 1:
                        raise
                    y = ctx.filter_bessel(torch.from_numpy(a).to(torch.device("cuda", ctx.device)), 1.0, cutoff=cutoff,
                                          sampling_freq=sampling_freq, order=order)
                    off = 0.0
            z, centre, step = ctx.requantise(y)
            filtered[i], onto_grid[i] = (y, off), z
            levels[i] = centre + off
            by_step.setdefault(step, []).append(i)
        # (a DC offset passes a unit-gain low-pass unchanged: the counts were filtered, the offset is put back)
        from .grid import Deferred
        filtered = [Deferred.from_tensor(y, off) for y, off in filtered]
        out = [None] * len(currents)

        def exact_bounds(idx, lens):
            """the reference's own arithmetic on the filtered currents THE USER SEES (the file's offset put back, as
            Event.current has it): one ps_segment_exact_f64 for the group"""
            ys = [filtered[i].tensor + filtered[i].offset if filtered[i].offset else filtered[i].tensor for i in idx]
            allt = ys[0].contiguous() if len(ys) == 1 else torch.cat(ys)
            starts = np.concatenate(([0], np.cumsum(lens)))[:-1]
            bounds, boff = ctx.segment_exact_f64(allt, starts, lens, self._params)
            return bounds.cpu().numpy(), boff

        for step, idx in by_step.items():
            lens = np.array([onto_grid[i].numel() for i in idx], dtype=np.int64)
            ev_off = np.concatenate(([0], np.cumsum(lens)))
            if self.off_grid == "exact":
                b, boff = exact_bounds(idx, lens)
            else:
                import warnings
                samples = onto_grid[idx[0]] if len(idx) == 1 else torch.cat([onto_grid[i] for i in idx])
                with warnings.catch_warnings():
                    if self.off_grid == "exact_on_near_tie":
                        warnings.simplefilter("ignore", engine.NearTieWarning)
                    # (one level for the call: the largest of its events' -- the larger the level, the larger the reference's noise)
                    bounds, boff, _ = ctx.segment_batch(samples, ev_off, self._params, step, want_stats=False,
                                                        offset_counts=_dc_counts(max((levels[i] for i in idx), key=abs), step))
                b = bounds.cpu().numpy()
                if self.off_grid == "exact_on_near_tie" and ctx.near_ties():
                    b, boff = exact_bounds(idx, lens)          # (the count is per call: the whole group is redone)
            for e, i in enumerate(idx):
                cur = filtered[i]
                edges = np.concatenate(([0], b[boff[e]:boff[e + 1]], [int(lens[e])])).tolist()
                out[i] = (cur, segments_from_edges(cur, edges))
        return out

    # ---- cparsers.pyx:120-155 -------------------------------------------------------------------
    def best_single_split(self, current):
        ctx = engine.context(self.device)
        t, q, _ = engine.to_device(current, self.quantum, self.offset, self.device)
        return ctx.best_single_split(t, q)

    # ---- cparsers.pyx:205-275 -------------------------------------------------------------------
    def score_samples(self, current, no_split=False):
        """One dense gain array per window scan, in the reference's scan order; with
        no_split=True a single scan of the whole array returned as a list (cparsers.pyx:261-263)."""
        ctx = engine.context(self.device)
        t, q, _ = engine.to_device(current, self.quantum, self.offset, self.device)
        n = t.numel()
        mw, maxw, W = self.min_width, self.max_width, self.window_width

        def scan(start, end):
            if end - start <= 2 * mw:                                     # :229-230
                return -1, []
            split, sc = ctx.score_window(t[start:end], q, mw, self.min_gain)
            full = np.zeros(n)
            full[start:end] = sc.cpu().numpy()
            return (split + start if split >= 0 else -1), full

        if no_split:
            return list(scan(0, n)[1])

        def rec(start, end):
            scores, split_at = [], -1
            for ps in range(start, end - 2 * mw, W // 2):                 # :265
                if ps > start + maxw:                                     # :266-268
                    split_at = min(start + maxw, end - mw)
                    return scores + rec(split_at, end)
                pe = min(end, ps + W)
                split_at, sc = scan(ps, pe)
                scores.append(sc)
                if split_at >= 0:
                    break
            if split_at == -1:
                if end - start <= maxw:
                    return scores
                split_at = min(start + maxw, end - mw)
            return scores + rec(start, split_at) + rec(split_at, end)

        return rec(0, n)
