"""One-off robustness check (GPU box): thousands of short events in one call (striding workgroups, event search in K0,
per-event offsets) against the oracle, default and LDS-window scans."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import oracle
from pypore_amd import _lib, engine, synth
from pypore_amd import engine as _ps_engine
_ps_engine.apply_env_defaults()           # tools take their settings from PORESEG_* variables; the product reads none
ctx = engine.context(0)
rng = np.random.RandomState(4242)
n_ev = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
lens = rng.randint(150, 30000, n_ev)
lens[rng.randint(0, n_ev, 20)] = 0                      # some empty events
lens[rng.randint(0, n_ev, 20)] = rng.randint(1, 220, 20)  # some shorter than 2*min_width
starts, pos, evs = [], 0, []
for e, n in enumerate(lens):
    pos += int(rng.randint(0, 5)); starts.append(pos)
    evs.append(synth.random_dwell_counts(int(n), 50_000 + e, 300, 6000) if n else np.zeros(0, dtype=np.int64)); pos += int(n)
buf = np.zeros(pos + 16, dtype=np.int16)
for k, s in zip(evs, starts):
    buf[s:s + len(k)] = k
dev = torch.from_numpy(buf).cuda()
params = dict(min_width=100, max_width=1000000, window_width=10000, prior_segments_per_second=10.)
t0 = time.time()
refs = [oracle.parse(k.astype(np.float64) * synth.QUANTUM, **params) for k in evs]
print("oracle: %.1f s for %d events, %d samples" % (time.time() - t0, n_ev, int(lens.sum())))
bad = 0
for bs in (1, 0):
    ctx.set_option("scan_bs", bs)
    b, boff, st = ctx.segment_events(dev, np.array(starts), lens.astype(np.int64), _lib.split_params(**params), synth.QUANTUM, want_stats=True)
    b = b.cpu().numpy()
    for e, ref in enumerate(refs):
        if not np.array_equal(b[boff[e]:boff[e + 1]], ref):
            bad += 1
            if bad < 5: print("MISMATCH scan_bs", bs, "event", e, "len", lens[e])
    print("scan_bs", bs, "boundaries", len(b), ctx.timings())
ctx.set_option("scan_bs", 1)
print("problems:", bad)
sys.exit(1 if bad else 0)
