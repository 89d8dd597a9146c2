"""End to end: the reference's DEFAULT workflow, Experiment.parse (DataTypes.py:956-988), on one synthetic .abf file --
read, detect events at 90 pA, first-order 2 kHz Bessel filtfilt of every event, SpeedyStatSplit(prior_segments_per_second
=10, cutoff_freq=2000) on every filtered event -- wall clock on the host, everything included.  Beside it the same steps
on the CPU for the first events (scipy-equivalent filter and segmenter of oracle/, one core), scaled to the file.
Round 5: then EIGHT files of half that length through Experiment.parse with workers=1 (the reference's loop, one file
after the other) and with the default workers (files in flight on their own host threads and device contexts).
usage: bench_experiment.py [samples, default 1e8]"""
import os, sys, time, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle
from pypore_amd import abf, synth
from pypore_amd import engine as _ps_engine
_ps_engine.apply_env_defaults()           # tools take their settings from PORESEG_* variables; the product reads none
from pypore_amd.DataTypes import Experiment

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
counts, _ = synth.file_trace_counts(n, 7)
path = os.path.join(tempfile.mkdtemp(), "bench.abf")
abf.write_abf(path, counts.astype(np.int16))
times = []
for rep in range(5):                                   # (first pass: allocations, library load; then every pass is printed)
    exp = None                                         # (the previous result is dropped first, as a caller's loop would)
    exp = Experiment([path])
    t0 = time.perf_counter()
    exp.parse(verbose=False)
    times.append(time.perf_counter() - t0)
dt = sorted(times[1:])[len(times[1:]) // 2]
print("Experiment.parse, wall clock of passes 1..5: " + " ".join("%.3f" % t for t in times) + " s (pass 1 sizes the buffers; median of the rest below)")
times2 = []
for rep in range(4):                                   # the same without the filter (BASELINE config 3 through the classes)
    exp2 = None
    exp2 = Experiment([path])
    t0 = time.perf_counter()
    exp2.parse(filter_params=None, segmenter=__import__("pypore_amd.parsers", fromlist=["x"]).SpeedyStatSplit(prior_segments_per_second=10), verbose=False)
    times2.append(time.perf_counter() - t0)
dt2 = sorted(times2[1:])[len(times2[1:]) // 2]
print("Experiment.parse(filter_params=None): %d events, %d segments: %.3f s = %.1f Msamples/s of file"
      % (len(exp2.events), len(exp2.segments), dt2, n / dt2 / 1e6))
ev = exp.events
ns = sum(len(e.current) for e in ev)
print("Experiment.parse: %d samples, %d events (%d samples in events), %d segments: %.3f s = %.1f Msamples/s of file"
      % (n, len(ev), ns, len(exp.segments), dt, n / dt / 1e6))
# the same on one CPU core for events worth ~4e6 samples
x = np.asarray(exp.files[0].current)
done = 0; t0 = time.perf_counter(); k = 0
for e in ev:
    a = int(round(e.start * exp.files[0].second)); m = len(e.current)
    y = oracle.bessel_filtfilt(x[a:a + m], 2000., exp.files[0].second)
    b = oracle.parse(y, prior_segments_per_second=10, cutoff_freq=2000.)
    done += m; k += 1
    if done >= 4_000_000: break
cpu = time.perf_counter() - t0
t0 = time.perf_counter(); oracle.lambda_events(x[:20_000_000], threshold=90.0); det = (time.perf_counter() - t0) * n / 20_000_000
print("CPU (oracle, one core): filter + segmentation of %d events (%d samples) %.2f s -> %.1f s for the file's events, "
      "+ detection %.1f s: %.1f Msamples/s of file" % (k, done, cpu, cpu * ns / done, det, n / (cpu * ns / done + det) / 1e6))

# ---- round 5: eight files, one after the other vs. in flight ----------------------------------------------------------
m = n // 2
paths = []
for f in range(8):
    c8, _ = synth.file_trace_counts(m, 100 + f)
    p8 = os.path.join(os.path.dirname(path), "f%d.abf" % f)
    abf.write_abf(p8, c8.astype(np.int16))
    paths.append(p8)
res = {}
for w in (1, None, 1, None, 1, None):
    e8 = None
    __import__("gc").collect()                         # (the previous result is collected outside the clock)
    e8 = Experiment(paths)
    t0 = time.perf_counter()
    e8.parse(verbose=False, workers=w)
    res.setdefault(w, []).append(time.perf_counter() - t0)
    key = (len(e8.events), len(e8.segments), [f.filename for f in e8.files])
    assert res.setdefault("key", key) == key                       # same events, segments and file order either way
seq, par = sorted(res[1])[1], sorted(res[None])[1]
print("Experiment.parse over 8 files of %.1e samples (%d events, %d segments): workers=1 %s s, default workers %s s: %.2f x"
      % (m, key[0], key[1], " ".join("%.3f" % t for t in res[1]), " ".join("%.3f" % t for t in res[None]), par / seq))
