"""Two host threads in the segmenter at once.  The reference is safe under the GIL -- a new FastStatSplit
per parse(), nothing shared (parsers.py:524-528); here ctypes drops the GIL during a call, so the threads must not meet
in one ps_ctx: engine.context() hands every thread its own, and a Context that IS shared serialises its calls."""
import threading

import numpy as np
import pytest

import oracle
from pypore_amd import synth


def test_every_thread_gets_its_own_context_handle(monkeypatch):
    """engine.context(): the main thread keeps the process-wide context of a device, another thread gets its own."""
    from pypore_amd import engine

    class Fake(object):
        def __init__(self, device):
            self.device = device
    monkeypatch.setattr(engine, "Context", Fake)
    monkeypatch.setattr(engine, "_contexts", {})
    a = engine.context(0)
    assert engine.context(0) is a
    seen = []
    th = threading.Thread(target=lambda: seen.extend([engine.context(0), engine.context(0)]))
    th.start(); th.join()
    assert seen[0] is seen[1] and seen[0] is not a


@pytest.mark.gpu
def test_two_threads_parse_concurrently():
    from pypore_amd.parsers import SpeedyStatSplit
    traces = [np.asarray(synth.counts_to_pa(synth.random_dwell_counts(400_000 + 1000 * t, 40 + t)), dtype=np.float64) for t in range(4)]
    refs = [oracle.parse(x, prior_segments_per_second=10.) for x in traces]
    errors = []

    def work(t):
        try:
            p = SpeedyStatSplit(prior_segments_per_second=10., quantum=synth.QUANTUM)
            for rep in range(12):
                x = traces[(t + rep) % 4]
                got = np.array([s.start for s in p.parse(x)[1:]], dtype=np.int32)
                if not np.array_equal(got, refs[(t + rep) % 4]):
                    errors.append((t, rep, got.size))
        except Exception as e:                                        # noqa: BLE001 -- reported on the main thread
            errors.append((t, repr(e)))
    th = [threading.Thread(target=work, args=(t,)) for t in range(4)]
    for x in th: x.start()
    for x in th: x.join()
    assert not errors, errors[:3]


@pytest.mark.gpu
def test_one_context_shared_by_two_threads_serialises_its_calls():
    import torch
    from pypore_amd import _lib, engine
    ctx = engine.context(0)
    params = _lib.split_params(min_width=100, max_width=1000000, window_width=10000, prior_segments_per_second=10., sampling_freq=1e5)
    ks = [synth.random_dwell_counts(600_000, 70 + t) for t in range(2)]
    dev = [torch.from_numpy(synth.counts_to_pa(k, np.float32)).cuda() for k in ks]
    refs = [oracle.parse(synth.counts_to_pa(k, np.float64), prior_segments_per_second=10.) for k in ks]
    errors = []

    def work(t):
        for rep in range(10):
            b, _, _ = ctx.segment_batch(dev[t], np.array([0, dev[t].numel()], dtype=np.int64), params, synth.QUANTUM, want_stats=False)
            if not np.array_equal(b.cpu().numpy(), refs[t]):
                errors.append((t, rep))
    th = [threading.Thread(target=work, args=(t,)) for t in range(2)]
    for x in th: x.start()
    for x in th: x.join()
    assert not errors, errors[:3]


@pytest.mark.gpu
@pytest.mark.parametrize("options", [dict(k0_waves=1), dict(k0_waves=2), dict(k0_waves=1, k0_admit=2), dict(k0_shared=1, k0_waves=2)])
def test_pool_options_do_not_change_a_result(options):
    """What engine.StreamPool sets on its contexts while it runs (round 5: K0 persistent at n waves per SIMD, at most M K0s of
    the device in flight; the shared front stream, which nothing sets by default): same boundaries as the oracle, fp32 and
    int16, one long event and many short ones at odd offsets, from four threads with a context each."""
    import torch
    from pypore_amd import _lib, engine
    params = _lib.split_params(min_width=100, max_width=1000000, window_width=10000, prior_segments_per_second=10., sampling_freq=1e5)
    k_long = synth.random_dwell_counts(1_500_000, 91)
    shorts = [synth.random_dwell_counts(30_000 + 137 * e, 200 + e, 300, 4000) for e in range(12)]
    buf, starts, lens, pos = [], [], [], 0
    for e, k in enumerate(shorts):
        pad = np.zeros(e % 7 + 1, dtype=np.int64)
        buf += [pad, k]; pos += pad.size; starts.append(pos); lens.append(k.size); pos += k.size
    cat = np.concatenate(buf + [np.zeros(16, dtype=np.int64)])
    ref_long = oracle.parse(synth.counts_to_pa(k_long, np.float64), prior_segments_per_second=10.)
    ref_short = [oracle.parse(synth.counts_to_pa(k, np.float64), prior_segments_per_second=10.) for k in shorts]
    errors = []

    def work(t):
        try:
            cx = engine.Context(0)
            for name, value in options.items():
                cx.set_option(name, value)
            for dtype in (torch.float32, torch.int16):
                as_dev = (lambda k: torch.from_numpy(synth.counts_to_pa(k, np.float32)).cuda()) if dtype == torch.float32 else \
                    (lambda k: torch.from_numpy(k.astype(np.int16)).cuda())
                d_long, d_cat = as_dev(k_long), as_dev(cat)
                for rep in range(3):
                    b, _, _ = cx.segment_batch(d_long, np.array([0, d_long.numel()], dtype=np.int64), params, synth.QUANTUM, want_stats=False)
                    if not np.array_equal(b.cpu().numpy(), ref_long):
                        errors.append((t, str(dtype), "long"))
                    b, off, _ = cx.segment_events(d_cat, np.array(starts, dtype=np.int64), np.array(lens, dtype=np.int64), params,
                                                  synth.QUANTUM, want_stats=False)
                    b = b.cpu().numpy()
                    for e in range(len(shorts)):
                        if not np.array_equal(b[off[e]:off[e + 1]], ref_short[e]):
                            errors.append((t, str(dtype), "short", e))
            cx.close()
        except Exception as ex:                                       # noqa: BLE001 -- reported on the main thread
            errors.append((t, repr(ex)))
    th = [threading.Thread(target=work, args=(t,)) for t in range(4)]
    for x in th: x.start()
    for x in th: x.join()
    assert not errors, errors[:4]


def test_pool_sets_the_options_of_shared_contexts_and_gives_the_callers_context_back(monkeypatch):
    """engine.StreamPool.run: while several contexts are in flight they are configured for a shared device (ONE option,
    "shared_device" n: persistent K0, a limit on the K0s in flight, no look-ahead helpers -- include/poreseg.h); afterwards the
    caller's own context -- contexts[0], the one SpeedyStatSplit.parse uses -- is a lone context again, also when a job
    raised.  Options go through the context's lock.  (No GPU: recording contexts.)"""
    import threading
    from pypore_amd import engine

    class Rec(object):
        def __init__(self, device):
            self.device = device
            self.opts = {}
            self.log = []
            self.lock = threading.RLock()
            self.handle = 1

        def set_option(self, name, value):
            self.opts[name] = value
            self.log.append((name, value))
    monkeypatch.setattr(engine, "Context", Rec)
    monkeypatch.setattr(engine, "_contexts", {})
    monkeypatch.setattr(engine, "_live", {})
    monkeypatch.setattr(engine, "DEFAULT_OPTIONS", {})
    monkeypatch.setattr(engine, "POOL_OVERRIDES", {})
    pool = engine.StreamPool(0, 4)
    during = []
    out = pool.run(8, lambda cx, k, t: during.append((t, dict(cx.opts))) or k)
    assert out == list(range(8))
    for t, o in during:
        assert o.get("shared_device") == 4, (t, o)
    assert pool.contexts[0].opts["shared_device"] == 1                                            # the caller's context: lone again
    assert all(c.opts["shared_device"] == 4 for c in pool.contexts[1:])
    during.clear()
    pool.run(8, lambda cx, k, t: during.append((t, dict(cx.opts))) or k)                          # a second run sets context 0 again
    assert all(o.get("shared_device") == 4 for _, o in during)
    pool.run(1, lambda cx, k, t: during.append((t, dict(cx.opts))) or k)                          # one job: nothing shared
    assert during[-1][1]["shared_device"] == 1
    # a job that raises: the caller's context is still given back
    def boom(cx, k, t):
        raise ValueError("job %d" % k)
    with pytest.raises(ValueError):
        pool.run(8, boom)
    assert pool.contexts[0].opts["shared_device"] == 1
    # experiments name further options (engine.POOL_OVERRIDES / the constructor): applied on top, only while shared
    pool2 = engine.StreamPool(0, 4, overrides={"k0_admit": 2, "k0_waves": 2})
    seen = []
    pool2.run(8, lambda cx, k, t: seen.append(dict(cx.opts)) or k)
    assert all(o["k0_admit"] == 2 and o["k0_waves"] == 2 and o["shared_device"] == 4 for o in seen)
    pool.close()
    pool2.close()


def test_helpers_follow_the_number_of_live_contexts_not_the_thread(monkeypatch):
    """engine.context(): a program that does all its work on ONE worker thread keeps the look-ahead helpers;
    once a second context is alive on the device all of them run without.  (No GPU: recording contexts.)"""
    import threading
    from pypore_amd import engine

    class Rec(object):
        def __init__(self, device):
            self.device, self.opts, self.handle, self.lock = device, {}, 1, threading.RLock()

        def set_option(self, name, value):
            self.opts[name] = value
    monkeypatch.setattr(engine, "Context", Rec)
    monkeypatch.setattr(engine, "_contexts", {})
    monkeypatch.setattr(engine, "_live", {})
    monkeypatch.setattr(engine, "DEFAULT_OPTIONS", {})
    got = {}

    def work(name):
        got[name] = engine.context(0)
    th = threading.Thread(target=work, args=("a",))
    th.start(); th.join()
    assert "lat_help" not in got["a"].opts                                                        # alone on the device: library default (helpers on)
    th = threading.Thread(target=work, args=("b",))
    th.start(); th.join()
    assert got["a"].opts.get("lat_help") == 0 and got["b"].opts.get("lat_help") == 0
    main = engine.context(0)
    assert main.opts.get("lat_help") == 0


def test_env_variables_reach_a_context_only_through_apply_env_defaults(monkeypatch):
    """The package reads no PORESEG_* variable on its own (round 6): a variable in the environment changes nothing until a
    test / tool calls engine.apply_env_defaults()."""
    import threading
    from pypore_amd import engine

    class Rec(object):
        def __init__(self, device):
            self.device, self.opts, self.handle, self.lock, self.tiling = device, {}, 1, threading.RLock(), None

        def set_option(self, name, value):
            self.opts[name] = value
    monkeypatch.setenv("PORESEG_MODE", "2")
    monkeypatch.setenv("PORESEG_STITCH", "host")
    monkeypatch.setenv("PORESEG_TILE", "50000")
    monkeypatch.setattr(engine, "DEFAULT_OPTIONS", {})
    monkeypatch.setattr(engine, "DEFAULT_TILING", [0, 0])
    monkeypatch.setattr(engine, "POOL_OVERRIDES", {})
    opts, tiling, pool = engine.options_from_env()
    assert opts == {"mode": 2, "stitch_host": 1} and tiling == [50000, 0] and pool == {}
    assert engine.DEFAULT_OPTIONS == {} and engine.DEFAULT_TILING == [0, 0]
    engine.apply_env_defaults()
    assert engine.DEFAULT_OPTIONS == {"mode": 2, "stitch_host": 1} and engine.DEFAULT_TILING == [50000, 0]
    src = open(engine.__file__).read() + open(engine._lib.__file__).read()
    import re
    reads = set(re.findall(r"environ(?:\.get)?[\[(]\s*\"(PORESEG_[A-Z0-9_]+)\"", src))
    assert reads <= {"PORESEG_LIB"}, reads                                                       # (everything else goes through options_from_env's table)
