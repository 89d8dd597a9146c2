#!/usr/bin/env python3
"""Records golden vectors from the compiled, UNMODIFIED reference (PyPore/cparsers.pyx and
the reference's own parsers.py) -- SURVEY.md 8(c) sets G1..G8.

Run in the build container only (needs /root/reference and oracle/build_reference.sh):

    ./oracle/build_reference.sh && python tests/golden/make_golden.py

Outputs (committed): tests/golden/golden.npz + tests/golden/manifest.json.  The inputs are
not stored when pypore_amd.synth can regenerate them from an integer spec; small hand-made
edge-case inputs are stored as int32 ADC counts (pA = counts * 2**-5).
"""
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ref_shims          # noqa: E402
from pypore_amd import synth          # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
cparsers = ref_shims.load_cparsers()
ref_parsers = ref_shims.load_reference_parsers()

arrays = {}
manifest = {"reference_sha256_cparsers_pyx": hashlib.sha256(
    open("/root/reference/PyPore/cparsers.pyx", "rb").read()).hexdigest(), "cases": []}

DEF = dict(min_width=100, max_width=1000000, window_width=10000, prior_segments_per_second=10.)


def gen_input(gen):
    kind = gen["kind"]
    if kind == "step":
        return synth.step_counts(gen["n"], gen["dwell"], gen["seed"], gen.get("level_offset", 0))
    if kind == "random_dwell":
        return synth.random_dwell_counts(gen["n"], gen["seed"], gen.get("lo", 1000), gen.get("hi", 20000))
    if kind == "stored":
        return arrays[gen["key"]]
    raise ValueError(kind)


def run_parse(name, gen, params, store_stats=True, offset=0):
    counts = gen_input(gen)
    x = synth.counts_to_pa(counts[offset:], np.float64)
    p = cparsers.FastStatSplit(**params)
    segs = p.parse(x)
    bounds = np.array([s.start for s in segs[1:]], dtype=np.int32)
    assert [s.end for s in segs] == list(bounds) + [len(x)]
    case = dict(name=name, op="parse", gen=gen, params=params, min_gain=repr(float(p.min_gain)),
                n=int(len(x)), offset=offset, n_bounds=int(len(bounds)))
    arrays[name + "/bounds"] = bounds
    if store_stats:
        arrays[name + "/mean"] = np.array([s.mean for s in segs], dtype=np.float64)
        arrays[name + "/std"] = np.array([s.std for s in segs], dtype=np.float64)
        arrays[name + "/min"] = np.array([s.min for s in segs], dtype=np.float64)
        arrays[name + "/max"] = np.array([s.max for s in segs], dtype=np.float64)
    manifest["cases"].append(case)
    print("%-28s n=%-10d bounds=%d" % (name, len(x), len(bounds)))
    return bounds


# ---- G1: config-1 signal ---------------------------------------------------------------------
run_parse("G1_config1", dict(kind="step", n=10000, dwell=2000, seed=1), DEF)

# ---- G2: 8 events of config-2 shape (two with dwell 500 / 20000) -----------------------------
for ev in range(6):
    run_parse("G2_event%d" % ev, dict(kind="step", n=50000, dwell=10000, seed=ev), DEF)
run_parse("G2_dwell500", dict(kind="step", n=50000, dwell=500, seed=106), DEF)
run_parse("G2_dwell20000", dict(kind="step", n=50000, dwell=20000, seed=107), DEF)

# ---- G3: per-candidate gains of one 10k window (pins var_c/_best_split_stepwise to the ulp) --
c3 = synth.step_counts(10000, 2000, 1)
p = cparsers.FastStatSplit(**DEF)
sc = np.array(p.score_samples(synth.counts_to_pa(c3, np.float64), no_split=True), dtype=np.float64)
arrays["G3_scores/scores"] = sc
manifest["cases"].append(dict(name="G3_scores", op="score_window", gen=dict(kind="step", n=10000, dwell=2000, seed=1),
                              params=DEF, min_gain=repr(float(p.min_gain)), n=10000))
c3b = synth.random_dwell_counts(7777, 11, 300, 3000)
sc = np.array(p.score_samples(synth.counts_to_pa(c3b, np.float64), no_split=True), dtype=np.float64)
arrays["G3_scores_b/scores"] = sc
manifest["cases"].append(dict(name="G3_scores_b", op="score_window", gen=dict(kind="random_dwell", n=7777, seed=11, lo=300, hi=3000),
                              params=DEF, min_gain=repr(float(p.min_gain)), n=7777))

# ---- G4: edge cases ----------------------------------------------------------------------------
run_parse("G4_n_eq_2mw", dict(kind="step", n=200, dwell=100, seed=3), DEF)
run_parse("G4_n_2mw_plus1", dict(kind="step", n=201, dwell=100, seed=3), DEF)
run_parse("G4_n_small", dict(kind="step", n=37, dwell=10, seed=3), DEF)
run_parse("G4_n_one", dict(kind="step", n=1, dwell=10, seed=3), DEF)
run_parse("G4_n_260", dict(kind="step", n=260, dwell=130, seed=4), DEF)
# both forced-split branches: flat noise (no hits) with small max_width
run_parse("G4_forced_flat", dict(kind="step", n=60000, dwell=1000000, seed=5),
          dict(min_width=100, max_width=12000, window_width=10000, prior_segments_per_second=10.))
run_parse("G4_forced_flat_b", dict(kind="step", n=33333, dwell=1000000, seed=6),
          dict(min_width=50, max_width=7001, window_width=3000, prior_segments_per_second=10.))
run_parse("G4_forced_late", dict(kind="step", n=12050, dwell=1000000, seed=7),
          dict(min_width=100, max_width=12000, window_width=10000, prior_segments_per_second=10.))
run_parse("G4_forced_mixed", dict(kind="random_dwell", n=150000, seed=8, lo=20000, hi=60000),
          dict(min_width=100, max_width=15000, window_width=10000, prior_segments_per_second=10.))
run_parse("G4_defaults_mingain0", dict(kind="step", n=10000, dwell=2000, seed=1),
          dict(min_width=100, max_width=1000000, window_width=10000))
run_parse("G4_mgps", dict(kind="step", n=30000, dwell=3000, seed=9),
          dict(min_width=100, max_width=1000000, window_width=10000, min_gain_per_sample=0.5))
run_parse("G4_cutoff2000", dict(kind="step", n=50000, dwell=5000, seed=10),
          dict(min_width=100, max_width=1000000, window_width=10000, prior_segments_per_second=10., cutoff_freq=2000.))
run_parse("G4_fpr", dict(kind="step", n=50000, dwell=2500, seed=12),
          dict(min_width=100, max_width=1000000, window_width=10000, prior_segments_per_second=20., false_positive_rate=50.))
# noise-free step (log(0) = -inf, inf/NaN gains; first +inf candidate wins, NaN never wins)
nf = np.repeat(synth.LEVEL_COUNTS[[0, 1, 2, 3, 4]], 700).astype(np.int32)
arrays["G4_noisefree/input"] = nf
run_parse("G4_noisefree", dict(kind="stored", key="G4_noisefree/input"), DEF)
nf2 = nf.copy()
nf2[::97] += 1            # nearly noise-free: mostly zero-variance runs
arrays["G4_nearly_noisefree/input"] = nf2
run_parse("G4_nearly_noisefree", dict(kind="stored", key="G4_nearly_noisefree/input"), DEF)
const = np.full(5000, 1600, dtype=np.int32)
arrays["G4_constant/input"] = const
run_parse("G4_constant", dict(kind="stored", key="G4_constant/input"), DEF)
# different widths
run_parse("G4_small_windows", dict(kind="random_dwell", n=120000, seed=13, lo=200, hi=4000),
          dict(min_width=20, max_width=50000, window_width=1500, prior_segments_per_second=10.))
run_parse("G4_odd_window", dict(kind="random_dwell", n=99991, seed=14, lo=500, hi=9000),
          dict(min_width=33, max_width=40000, window_width=4097, prior_segments_per_second=10.))
run_parse("G4_minwidth2", dict(kind="random_dwell", n=20000, seed=15, lo=50, hi=900),
          dict(min_width=2, max_width=5000, window_width=400, prior_segments_per_second=10.))
run_parse("G4_big_window", dict(kind="random_dwell", n=300000, seed=16, lo=1000, hi=20000),
          dict(min_width=100, max_width=1000000, window_width=30000, prior_segments_per_second=10.))
run_parse("G4_window_eq_2mw", dict(kind="random_dwell", n=40000, seed=17, lo=500, hi=5000),
          dict(min_width=100, max_width=30000, window_width=200, prior_segments_per_second=10.))

# ---- G5: best_single_split ------------------------------------------------------------------------
for nm, gen in [("G5_bss_config1", dict(kind="step", n=10000, dwell=2000, seed=1)),
                ("G5_bss_two", dict(kind="step", n=6000, dwell=3500, seed=21)),
                ("G5_bss_flat", dict(kind="step", n=3000, dwell=1000000, seed=22))]:
    x = synth.counts_to_pa(gen_input(gen), np.float64)
    g, i = cparsers.FastStatSplit(**DEF).best_single_split(x)
    manifest["cases"].append(dict(name=nm, op="best_single_split", gen=gen, gain=repr(float(g)), index=int(i), n=len(x)))
    print("%-28s gain=%r idx=%d" % (nm, g, i))

# ---- G6: event detector (the reference's own parsers.py lambda_event_parser) ---------------------
# [110 pA x 50k | 45 pA-ish event 150k | 110 x 30k | short dip 50k (fails duration) | 110 x 20k |
#  event 130k that dips below -0.5 pA once (fails min rule) | 110 x 10k | event 120k | 110 x 5k]
def _ev(n, seed, off):
    return synth.step_counts(n, 7000, seed, off)
parts = [synth.OPEN_COUNTS + synth.noise_counts(31, 0, 50000), _ev(150000, 32, 0),
         synth.OPEN_COUNTS + synth.noise_counts(33, 0, 30000), _ev(50000, 34, 1),
         synth.OPEN_COUNTS + synth.noise_counts(35, 0, 20000), _ev(130000, 36, 2),
         synth.OPEN_COUNTS + synth.noise_counts(37, 0, 10000), _ev(120000, 38, 3),
         synth.OPEN_COUNTS + synth.noise_counts(39, 0, 5000)]
parts[5] = parts[5].copy()
parts[5][77777] = -40            # -1.25 pA: violates min > -0.5
g6 = np.concatenate(parts).astype(np.int32)
arrays["G6_events/input"] = g6
x6 = synth.counts_to_pa(g6, np.float64)
evs = ref_parsers.lambda_event_parser(threshold=90).parse(x6)
arrays["G6_events/starts"] = np.array([int(e.start) for e in evs], dtype=np.int64)
arrays["G6_events/lengths"] = np.array([int(e.duration) for e in evs], dtype=np.int64)
manifest["cases"].append(dict(name="G6_events", op="lambda_event_parser", threshold=90, n=int(len(g6)),
                              n_events=len(evs)))
print("G6_events", [(int(e.start), int(e.duration)) for e in evs])
# per-event segmentation through the reference wrapper (parsers.py:505-534)
spl = ref_parsers.SpeedyStatSplit(prior_segments_per_second=10.)
for k, e in enumerate(evs):
    segs = spl.parse(e.current)
    arrays["G6_events/ev%d_bounds" % k] = np.array([s.start for s in segs[1:]], dtype=np.int32)
    print("   event %d: %d segments" % (k, len(segs)))

# ---- G8: anchor-shift pairs (pins the tiling/stitch logic) --------------------------------------
gen8 = dict(kind="random_dwell", n=400000, seed=41)
b_full = run_parse("G8_full", gen8, DEF, store_stats=False)
for a in (1, 4999, 123457, 250000):
    run_parse("G8_shift%d" % a, gen8, DEF, store_stats=False, offset=a)

# ---- G9: medium traces with assorted dwell regimes ---------------------------------------------
run_parse("G9_rd_2M", dict(kind="random_dwell", n=2000000, seed=3), DEF, store_stats=False)
run_parse("G9_rd_short_dwell", dict(kind="random_dwell", n=1000000, seed=51, lo=150, hi=1500), DEF, store_stats=False)
run_parse("G9_rd_long_dwell", dict(kind="random_dwell", n=3000000, seed=52, lo=30000, hi=200000), DEF, store_stats=False)
run_parse("G9_cutoff", dict(kind="random_dwell", n=1000000, seed=53),
          dict(min_width=100, max_width=1000000, window_width=10000, prior_segments_per_second=10., cutoff_freq=2000.),
          store_stats=False)

# ---- G7: 10^8-sample trace: count + SHA-256 + first/last 32 -------------------------------------
if "--no-g7" not in sys.argv:
    t0 = time.time()
    gen7 = dict(kind="random_dwell", n=100000000, seed=2024)
    x7 = synth.counts_to_pa(gen_input(gen7), np.float64)
    t1 = time.time()
    p7 = cparsers.FastStatSplit(**DEF)
    segs = p7.parse(x7)
    t2 = time.time()
    b7 = np.array([s.start for s in segs[1:]], dtype=np.int32)
    del segs
    arrays["G7_1e8/first32"] = b7[:32]
    arrays["G7_1e8/last32"] = b7[-32:]
    manifest["cases"].append(dict(name="G7_1e8", op="parse_digest", gen=gen7, params=DEF, n=100000000,
                                  n_bounds=int(len(b7)), sha256=hashlib.sha256(b7.tobytes()).hexdigest(),
                                  ref_seconds=round(t2 - t1, 2), gen_seconds=round(t1 - t0, 2)))
    print("G7_1e8 bounds=%d  ref parse %.1fs" % (len(b7), t2 - t1))
else:
    old = json.load(open(os.path.join(HERE, "manifest.json")))
    oldz = np.load(os.path.join(HERE, "golden.npz"))
    manifest["cases"] += [c for c in old["cases"] if c["name"] == "G7_1e8"]
    for k in ("G7_1e8/first32", "G7_1e8/last32"):
        arrays[k] = oldz[k]

np.savez_compressed(os.path.join(HERE, "golden.npz"), **arrays)
with open(os.path.join(HERE, "manifest.json"), "w") as f:
    json.dump(manifest, f, indent=1)
print("wrote", len(arrays), "arrays,", len(manifest["cases"]), "cases")
