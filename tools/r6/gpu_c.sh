#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
echo "build: $(sha256sum pypore_amd/libporeseg.so | cut -c1-16)  $(date -u +%FT%TZ)"
timeout 1500 python -m pytest tests/test_parity_sample.py -q -m gpu -x -s 2>&1 | tail -15
timeout 900 python -m pytest tests -q -m gpu --deselect tests/test_parity_sample.py 2>&1 | tail -4
python - <<'PY'
import time, numpy as np, torch, sys
sys.path.insert(0, '.')
from pypore_amd import synth
from pypore_amd.parsers import SpeedyStatSplit
import warnings
warnings.simplefilter("ignore")
for n in (100000, 1000000):
    x = synth.offgrid_trace(n, 5, 1.0, 1000, 20000)
    for mode in ("requantise", "exact"):
        p = SpeedyStatSplit(off_grid=mode, prior_segments_per_second=10.)
        p.parse(x)
        t0 = time.perf_counter()
        for _ in range(3):
            s = p.parse(x)
        print("off-grid trace of %d samples, %s: %.2f ms per parse (incl. upload), %d segments" % (n, mode, (time.perf_counter() - t0) / 3 * 1e3, len(s)))
PY
