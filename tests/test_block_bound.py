"""The two-boundary bound of an 8-sample block (seg_bs.hpp: bs_block_bound2, used by the drain to re-judge queued blocks):
the numpy restatement in tools/experiments/two_boundary_bound.py must never lie below the largest gain inside a block."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "experiments"))
import two_boundary_bound as tbb  # noqa: E402


@pytest.mark.parametrize("kind", ["noise", "offset", "step", "spikes", "ramp", "quiet"])
def test_bound_is_never_below_an_interior_gain(kind):
    rng = np.random.default_rng(["noise", "offset", "step", "spikes", "ramp", "quiet"].index(kind) + 100)
    for n in (800, 1500, 6000):
        b = tbb.bounds(tbb.window(kind, rng, n))
        b = b[np.isfinite(b[:, 1])]
        assert b.shape[0] > 10
        assert (b[:, 1] <= b[:, 2] + 1e-6).all()            # corner bound (what the sweep uses)
        assert (b[:, 1] <= b[:, 3] + 1e-6).all()            # two boundaries, exact block sums
        assert (b[:, 1] <= b[:, 5] + 1e-6).all()            # two boundaries, from what the kernel has (min with the corner bound)
        assert (b[:, 1] <= b[:, 6] + 0.02 + 8e-6 * n).all()  # the same in float32 with the kernel's margins (screen tolerance delta)


def test_bound_is_tight_on_noise():
    rng = np.random.default_rng(11)
    b = tbb.bounds(tbb.window("noise", rng, 8000))
    b = b[np.isfinite(b[:, 1])]
    loose_corner, loose_two = b[:, 2] - b[:, 4], b[:, 5] - b[:, 4]
    assert np.median(loose_corner) > 8.0 and np.median(loose_two) < 2.0
    thr = 18.4204807339517 * tbb.LOG2E - 0.4
    assert (b[:, 2] >= thr).mean() > 0.01 and (b[:, 5] >= thr).mean() < 0.002
