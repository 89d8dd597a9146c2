"""File-level workflow of Experiment.parse (DataTypes.py:956-988, filter off) kept on the GPU:
event detection (lambda_event_parser defaults) followed by SpeedyStatSplit on every event, the
events being sub-ranges of the one device-resident trace (no copies, no host round trip of samples).
BASELINE config 3 = this on a 10^8-sample .abf."""
import numpy as np

from . import _lib, engine


def segment_file_trace(samples, quantum, params=None, threshold=90.0, min_duration=100000, min_current=-0.5,
                       offset_counts=0, device=None, want_stats=False, offset=0.0, ctx=None, single_pass=True):
    """samples: 1-D CUDA tensor (float32 pA on the `quantum` grid, or int16 ADC counts); pA = (count + offset_counts)
    * quantum + offset (`offset`: the part of an .abf offset that is not a whole number of counts; the detector's
    thresholds move by it, the segmenter's gains do not depend on it).
    single_pass=False: the two calls ps_detect_events + ps_segment_events (two passes over the samples), as until round 5.
    Returns (ev_start, ev_len, bounds int32 CUDA tensor, bounds_off, stats or None)."""
    ctx = ctx or engine.context(device)
    if params is None:
        params = _lib.split_params(prior_segments_per_second=10.)
    if single_pass:
        # one library call, one pass over the samples: K0 over the whole trace serves the detector and every event (round 6)
        st, ln, bounds, boff, stats = ctx.detect_segment_trace(samples, quantum, params, threshold - offset, min_duration,
                                                               min_current - offset, offset_counts, want_stats)
    else:
        st, ln = ctx.detect_events(samples, quantum, threshold - offset, min_duration, min_current - offset, offset_counts)
        bounds, boff, stats = ctx.segment_events(samples, st, ln, params, quantum, offset_counts, want_stats)
    if stats is not None and offset:
        stats[:, 0] += offset; stats[:, 2] += offset; stats[:, 3] += offset
    return st, ln, bounds, boff, stats


def parse_abf(path, params=None, threshold=90.0, device=None):
    """read_abf -> File.parse(lambda_event_parser(threshold)) -> Event.parse(SpeedyStatSplit) with the raw
    int16 counts on the GPU (2 B/sample).  Returns (time_step_msec, ev_start, ev_len, list of boundary arrays)."""
    import torch
    from .abf import read_abf_counts
    dt, counts, scale, offset = read_abf_counts(path)
    dev = torch.device("cuda", torch.cuda.current_device() if device is None else int(device))
    t = torch.from_numpy(np.array(counts, dtype=np.int16)).to(dev)     # one host copy out of the memmap, then H2D
    st, ln, bounds, boff, _ = segment_file_trace(t, scale, params, threshold, device=device, offset=offset)
    b = bounds.cpu().numpy()
    return dt, st, ln, [b[boff[e]:boff[e + 1]] for e in range(len(st))]
