"""Where the wall clock of Experiment.parse goes (cProfile of the second pass over a synthetic 1e8-sample .abf)."""
import os, sys, time, tempfile, cProfile, pstats
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pypore_amd import abf, synth
from pypore_amd import engine as _ps_engine
_ps_engine.apply_env_defaults()           # tools take their settings from PORESEG_* variables; the product reads none
from pypore_amd.DataTypes import Experiment, File
from pypore_amd.parsers import SpeedyStatSplit, lambda_event_parser

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
counts, _ = synth.file_trace_counts(n, 7)
path = os.path.join(tempfile.mkdtemp(), "bench.abf")
abf.write_abf(path, counts.astype(np.int16))
Experiment([path]).parse(verbose=False)
for rep in range(2):
    t0 = time.perf_counter(); f = File(path); t1 = time.perf_counter()
    f.parse(lambda_event_parser(threshold=90)); t2 = time.perf_counter()
    f.parse_events(SpeedyStatSplit(prior_segments_per_second=10, cutoff_freq=2000.), (1, 2000)); t3 = time.perf_counter()
    print("File() %.3f s, detection %.3f s, filter + segmentation of %d events %.3f s" % (t1 - t0, t2 - t1, f.n, t3 - t2))
pr = cProfile.Profile()
pr.enable()
Experiment([path]).parse(verbose=False)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
pr = cProfile.Profile()
pr.enable()
Experiment([path]).parse(filter_params=None, segmenter=SpeedyStatSplit(prior_segments_per_second=10), verbose=False)
pr.disable()
print("---- filter_params=None")
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
import time as _t
for rep in range(3):
    t0 = _t.perf_counter(); e = Experiment([path]); e.parse(verbose=False); t1 = _t.perf_counter()
    del e
    t2 = _t.perf_counter()
    print("Experiment.parse %.3f s, dropping the result %.3f s" % (t1 - t0, t2 - t1))
