#!/usr/bin/env python3
"""Round 5: Experiment.parse over several files, one after the other (workers=1, the reference's loop) against files in flight
on worker threads (workers=2, 4): wall clock of alternating runs.  usage: bench_experiment_files.py [files] [samples per file]"""
import gc, os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pypore_amd import abf, synth
from pypore_amd import engine as _ps_engine
_ps_engine.apply_env_defaults()           # tools take their settings from PORESEG_* variables; the product reads none
from pypore_amd.DataTypes import Experiment
nf = int(sys.argv[1]) if len(sys.argv) > 1 else 8
m = int(float(sys.argv[2])) if len(sys.argv) > 2 else 50_000_000
d = tempfile.mkdtemp()
paths = []
for f in range(nf):
    c, _ = synth.file_trace_counts(m, 100 + f)
    p = os.path.join(d, "f%d.abf" % f)
    abf.write_abf(p, c.astype(np.int16))
    paths.append(p)
Experiment(paths).parse(verbose=False, workers=1)
Experiment(paths).parse(verbose=False, workers=4)
res = {}
for rep in range(7):
    for w in (1, 2, 4):
        e = None
        gc.collect()                                     # (the previous result -- 10^5 objects in cycles -- is collected outside the clock)
        e = Experiment(paths)
        t0 = time.perf_counter()
        e.parse(verbose=False, workers=w)
        res.setdefault(w, []).append(time.perf_counter() - t0)
for w, v in res.items():
    v = sorted(v)
    print("%d files of %.1e samples, workers=%d: median %.3f s (min %.3f, max %.3f)  %s" % (nf, m, w, v[len(v) // 2], v[0], v[-1], " ".join("%.3f" % x for x in v)))
