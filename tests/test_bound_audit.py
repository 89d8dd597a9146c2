"""Every pruning bound of the block-sum window scan against the gains it covers.

CPU: the numpy restatement of the group bound (tools/experiments/group_bound.py: convexity of the gain in the left part's
(k, S1, S2), amplitudes from block sums and min / max as K0 forms them) never lies below a gain inside the group, on the
windows the reference recursion (cparsers.pyx:180-203) scans.
GPU (ps_audit_bounds): the product's own scan code, compiled with its audit switch, sweeps every row of the given windows
and compares each bound it forms -- the corner bound of an 8-sample block, the two-boundary bound of the drain, the group
bound of the coarse pass -- with the screened gains of all candidates the bound covers, evaluated one by one from the raw
samples on the device.  A violation is a gain above its bound by more than 2 delta(n), the slack the pruning levels carry."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools", "experiments"))
import group_bound as gb  # noqa: E402

from pypore_amd import synth  # noqa: E402

PARAMS = dict(min_width=100, max_width=1000000, window_width=10000, prior_segments_per_second=10., sampling_freq=1e5)


def _traces():
    """(name, int counts) -- noise steps, the same on a large DC offset, a quiet trace with bursts, spikes, a ramp"""
    rng = np.random.default_rng(77)
    n = 400000
    base = synth.random_dwell_counts(n, 91).astype(np.int64)
    quiet = np.rint(rng.normal(0, 1.2, n)).astype(np.int64)
    for s in range(20000, n - 400, 37111):
        quiet[s:s + 200] += np.rint(rng.normal(0, 150, 200)).astype(np.int64)
    spikes = synth.random_dwell_counts(n, 92).astype(np.int64)
    idx = rng.integers(0, n, 300)
    spikes[idx] += np.rint(rng.normal(0, 2500, idx.size)).astype(np.int64)
    ramp = (synth.noise_counts(93, 0, n) + np.linspace(0, 6000, n).astype(np.int64))
    dense = synth.random_dwell_counts(n, 94, 150, 1500).astype(np.int64)
    return [("steps", base), ("offset", base + 9000), ("quiet", quiet), ("spikes", np.clip(spikes, -12000, 15000)),
            ("ramp", ramp), ("dense", dense)]


def _windows(n, rng, W=10000, mw=100):
    """the grid the recursion walks from a few anchors, plus windows of odd lengths and alignments"""
    w = []
    for a in (0, 1237, 7, 40003):
        ps = a
        while ps < n - 2 * mw - 1:
            w.append((ps, min(n, ps + W)))
            ps += W // 2
    for _ in range(300):
        ln = int(rng.choice([260, 700, 1100, 2049, 4096, 5555, 9999, 10000, 16000]))
        ps = int(rng.integers(0, n - ln))
        w.append((ps, ps + ln))
    return [(a, b) for a, b in w if b - a > 2 * mw]


@pytest.mark.parametrize("gs,mode", [(256, "group"), (256, "sym"), (256, "tight"), (256, "kernel"), (128, "kernel")])
def test_numpy_group_bound_never_below_an_interior_gain(gs, mode, capsys):
    sys.argv = ["group_bound.py"]
    y = synth.random_dwell_counts(300000, 2024).astype(np.float64)
    y -= y[0]
    c1, c2, wins, bounds = gb.rec_windows(y)
    assert len(wins) > 80 and len(bounds) > 20
    res = gb.run(y, c1, c2, wins, gs, mode)
    assert res["violations"] == 0 and res["groups"] > 1000
    assert res["rows_live"] < 0.4 * res["rows"]          # the point of it: most rows of the sweep are not needed


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["f32", "i16"])
def test_device_bounds_cover_every_interior_gain(dtype):
    import torch
    from pypore_amd import _lib, engine
    ctx = engine.context(0)
    params = _lib.split_params(**PARAMS)
    rng = np.random.default_rng(5)
    seen = {"corner": 0, "two_boundary": 0, "group": 0}
    for name, k in _traces():
        if dtype == "f32":
            x = torch.from_numpy(synth.counts_to_pa(k, np.float32)).cuda()
        else:
            x = torch.from_numpy(k.astype(np.int16)).cuda()
        r = ctx.audit_bounds(x, synth.QUANTUM, params, _windows(k.size, rng))
        for kind, cnt in (("corner", "blocks"), ("two_boundary", "blocks"), ("group", "groups")):
            assert r[kind]["violations"] == 0, (name, kind, r)
            seen[kind] += r[kind][cnt]
        assert r["windows_with_coarse_pass"] > 100, (name, r)
    # every kind of bound was exercised many times over
    assert seen["corner"] > 1e5 and seen["two_boundary"] > 1e5 and seen["group"] > 1e4, seen


@pytest.mark.gpu
def test_audit_refuses_what_the_scan_would_not_run():
    import torch
    from pypore_amd import _lib, engine
    ctx = engine.context(0)
    x = torch.from_numpy(synth.counts_to_pa(synth.random_dwell_counts(50000, 3), np.float32)).cuda()
    with pytest.raises(ValueError):
        ctx.audit_bounds(x, synth.QUANTUM, _lib.split_params(**PARAMS), [(0, 150)])           # not a window: <= 2 min_width
    with pytest.raises(ValueError):
        ctx.audit_bounds(x, synth.QUANTUM, _lib.split_params(**PARAMS), [(0, 60000)])         # beyond the trace
